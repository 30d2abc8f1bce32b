"""jax.tree.map / flatten on nested dicts, lists and tuples (dict children in sorted-key order, like jax)."""


def map(f, tree, *rest):  # pylint: disable=redefined-builtin
  if isinstance(tree, dict):
    return {k: map(f, tree[k], *[r[k] for r in rest]) for k in tree}
  if isinstance(tree, (list, tuple)):
    return type(tree)(map(f, v, *[r[i] for r in rest]) for i, v in enumerate(tree))
  if tree is None:
    return None
  return f(tree, *rest)


def flatten(tree):
  leaves = []

  def walk(t):
    if isinstance(t, dict):
      for k in sorted(t):
        walk(t[k])
    elif isinstance(t, (list, tuple)):
      for v in t:
        walk(v)
    elif t is not None:
      leaves.append(t)
  walk(tree)
  return leaves, None


def leaves(tree):
  return flatten(tree)[0]
