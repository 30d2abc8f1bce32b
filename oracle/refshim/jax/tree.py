"""jax.tree.map / flatten / leaves on nested dicts, lists, tuples and NamedTuples (dict children in sorted-key order,
like jax; `None` and an empty NamedTuple are nodes WITHOUT leaves, like jax).  `flatten` returns a tree definition with
`unflatten` (big_vision/utils.py:642-668 threads a token tree through it); `map` follows the FIRST tree: where that one
has a leaf the other trees contribute the whole subtree they have there (optax.masked relies on it: a `MaskedNode()` sits
where the mask says False)."""


def _is_namedtuple(t):
  return isinstance(t, tuple) and hasattr(t, "_fields")


def _rebuild(t, children):
  if _is_namedtuple(t):
    return type(t)(*children)
  return type(t)(children)


def map(f, tree, *rest, is_leaf=None):  # pylint: disable=redefined-builtin
  if is_leaf is not None and is_leaf(tree):
    return f(tree, *rest)
  if isinstance(tree, dict):
    return {k: map(f, tree[k], *[r[k] for r in rest], is_leaf=is_leaf) for k in tree}
  if isinstance(tree, (list, tuple)):
    return _rebuild(tree, [map(f, v, *[r[i] for r in rest], is_leaf=is_leaf) for i, v in enumerate(tree)])
  if tree is None:
    return None
  return f(tree, *rest)


class TreeDef:
  """Structure of a flattened tree: `unflatten(leaves)` puts a sequence of leaves back in flatten's order."""

  def __init__(self, skeleton, n):
    self.skeleton, self.num_leaves = skeleton, n

  def unflatten(self, leaves):
    it = iter(leaves)

    def build(s):
      kind, t, children = s
      if kind == "leaf":
        return next(it)
      if kind == "none":
        return None
      if kind == "dict":
        return {k: build(c) for k, c in children}
      return _rebuild(t, [build(c) for c in children])

    out = build(self.skeleton)
    rest = list(it)
    assert not rest, f"{len(rest)} leaves too many for this tree definition"
    return out


def flatten(tree, is_leaf=None):
  leaves = []

  def walk(t):
    if is_leaf is not None and is_leaf(t):
      leaves.append(t)
      return ("leaf", None, None)
    if isinstance(t, dict):
      return ("dict", None, [(k, walk(t[k])) for k in sorted(t)])
    if isinstance(t, (list, tuple)):
      return ("seq", t, [walk(v) for v in t])
    if t is None:
      return ("none", None, None)
    leaves.append(t)
    return ("leaf", None, None)

  skeleton = walk(tree)
  return leaves, TreeDef(skeleton, len(leaves))


def unflatten(treedef, leaves):
  return treedef.unflatten(leaves)


def leaves(tree, is_leaf=None):
  return flatten(tree, is_leaf)[0]


def structure(tree):
  return flatten(tree)[1]
