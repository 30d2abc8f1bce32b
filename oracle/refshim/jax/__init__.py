"""jax stand-in (see ../README.md): enough of the top-level namespace for the reference's model files to import and
for their `__call__` bodies to run on numpy float64."""
from . import numpy  # noqa: F401  (jax.numpy)
from . import tree, tree_util, random, nn, lax, scipy, sharding, image  # noqa: F401

Array = object


class _Policies:
  """jax.checkpoint_policies: vit.Encoder looks its remat policy up by name; remat is the identity here."""
  nothing_saveable = "nothing_saveable"
  dots_with_no_batch_dims_saveable = "dots_with_no_batch_dims_saveable"


checkpoint_policies = _Policies()


def tree_map(f, tree_, *rest):
  return tree.map(f, tree_, *rest)


def local_devices():
  return []


def device_count():
  return lax.device_count()


class _Config:
  def parse_flags_with_absl(self):
    pass

  def update(self, *a, **k):
    pass


config = _Config()


def process_index():
  return 0


def vmap(fn, in_axes=0, out_axes=0):
  """jax.vmap over the leading axis of every argument (in_axes all 0): a Python loop, results stacked."""
  import numpy as _np
  assert out_axes == 0 and (in_axes == 0 or all(a == 0 for a in in_axes)), (in_axes, out_axes)

  def mapped(*args):
    n = len(args[0])
    return _np.stack([fn(*[a[i] for a in args]) for i in range(n)])

  return mapped
