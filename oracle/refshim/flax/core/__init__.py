"""flax.core: variables are plain nested dicts here, so freeze / unfreeze copy the structure."""


def unfreeze(tree):
  return {k: unfreeze(v) for k, v in tree.items()} if isinstance(tree, dict) else tree


freeze = unfreeze
FrozenDict = dict
