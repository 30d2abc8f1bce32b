"""jax.tree_util: the older spellings of jax.tree (same semantics, see tree.py)."""
from .tree import flatten as tree_flatten, leaves as tree_leaves, map as tree_map, structure as tree_structure  # noqa: F401


def tree_unflatten(treedef, leaves):
  return treedef.unflatten(leaves)
