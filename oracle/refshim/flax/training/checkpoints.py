"""flax.training.checkpoints.convert_pre_linen: renames pre-Linen `Dense_0`-style trees; every tree that reaches it
here is already in Linen naming, for which the real function is the identity."""


def convert_pre_linen(params):
  return params
