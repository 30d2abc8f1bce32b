"""Checkpoint <-> init reconciliation.

Contract (what a caller of big_vision/models/common.py:24-92 relies on): the result has exactly the leaves of
`inited`; a leaf comes from `loaded` unless one of the `dont_load` regexes full-matches its `/`-joined name; a name
that exists on one side only and is not excused by `dont_load` is an error whose text lists both trees and the two
difference sets (with the ` - ` / ` + ` markers tools grep for).  The error text is API; the code below is ours."""
from big_vision_amd import utils as u

_SECTIONS = (
    ("Params in checkpoint", "  "),
    ("Params in model (code)", "  "),
    ("Params in model (code) but not in checkpoint and not `dont_load`ed", " - "),
    ("Params in checkpoint but not in model (code) and not `dont_load`ed", " + "),
)


def _report(groups):
  """One block per non-empty name set: a title line, then the sorted names behind that set's marker."""
  blocks = []
  for (title, marker), names in zip(_SECTIONS, groups):
    blocks.append("\n".join([f"{title}:"] + [marker + n for n in sorted(names)]) if names else "")
  return "\n".join(blocks)


def merge_params(loaded, inited, dont_load=(), match_dtype=False):
  """`loaded` reshaped onto the structure of `inited` (see the module docstring).  `match_dtype` casts each
  taken leaf to the dtype of the init leaf it replaces."""
  if inited is None:
    return loaded
  excused = u.check_and_compile_patterns(dont_load)
  have = dict(u.tree_flatten_with_names(loaded)[0])
  want = dict(u.tree_flatten_with_names(inited)[0])
  pinned = {n for n in have.keys() | want.keys() if any(rx.fullmatch(n) for rx in excused)}

  only_model = want.keys() - have.keys() - pinned
  only_ckpt = have.keys() - want.keys() - pinned
  if only_model or only_ckpt:
    raise ValueError(_report((have.keys(), want.keys(), only_model, only_ckpt)))

  def pick(name, init_leaf):
    if name in pinned or name not in have:
      return init_leaf
    leaf = have[name]
    if match_dtype and hasattr(leaf, "dtype") and leaf.dtype != init_leaf.dtype:
      leaf = leaf.to(init_leaf.dtype) if hasattr(leaf, "to") else leaf.astype(init_leaf.dtype)
    return leaf

  names = list(want)
  return u.recover_tree(names, [pick(n, want[n]) for n in names])
