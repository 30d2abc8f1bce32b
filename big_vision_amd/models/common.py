"""Checkpoint <-> init reconciliation, mirroring big_vision/models/common.py:24-92."""
from big_vision_amd import utils as u


def merge_params(loaded, inited, dont_load=(), match_dtype=False):
  """Makes `loaded` match the structure of `inited`; `dont_load` regexes keep
  the init value.  Raises ValueError with a formatted diff on unexplained
  mismatches (same contract as the reference)."""
  del match_dtype
  if inited is None:
    return loaded
  dont_load = u.check_and_compile_patterns(dont_load)

  def should_merge(name):
    return not any(p.fullmatch(name) for p in dont_load)

  loaded_flat = dict(u.tree_flatten_with_names(loaded)[0])
  inited_flat = dict(u.tree_flatten_with_names(inited)[0])
  merged = {}
  for name, init_val in inited_flat.items():
    merged[name] = loaded_flat[name] if (name in loaded_flat and should_merge(name)) else init_val

  def pp(title, names, indent="  "):
    return (f"{title}:\n" + "\n".join(f"{indent}{k}" for k in sorted(names))) if names else ""

  not_in_loaded = {k for k in inited_flat.keys() - loaded_flat.keys() if should_merge(k)}
  not_in_inited = {k for k in loaded_flat.keys() - inited_flat.keys() if should_merge(k)}
  if not_in_loaded or not_in_inited:
    raise ValueError(
        pp("Params in checkpoint", loaded_flat.keys()) + "\n" +
        pp("Params in model (code)", inited_flat.keys()) + "\n" +
        pp("Params in model (code) but not in checkpoint and not `dont_load`ed", not_in_loaded, indent=" - ") + "\n" +
        pp("Params in checkpoint but not in model (code) and not `dont_load`ed", not_in_inited, indent=" + "))
  return u.recover_tree(merged.keys(), merged.values())
