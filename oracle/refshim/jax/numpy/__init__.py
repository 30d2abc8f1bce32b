"""jax.numpy stand-in = numpy, with ONE policy: every floating-point array is float64, whatever float dtype is asked
for (../README.md: dtype promotion is not pinned here; the comparison against the fp64 oracle needs full precision)."""
import numpy as _np
from numpy import (arange, roll, eye, diag, dot, not_equal, logical_not, logical_or, logical_and, pad, finfo, equal, cos, einsum, exp, log, mean, mgrid, prod, reshape, sin, sqrt, sum, tile,  # noqa: F401
                   linalg, maximum, minimum, where, stack, transpose, zeros_like, ones_like, tanh, abs, max, min,
                   argmin, argmax, take, expand_dims, squeeze, ndarray, pi, newaxis, inf, int32, int64, uint32, bool_, searchsorted,
                   square, delete, argsort)

float32 = _np.float32
float64 = _np.float64
bfloat16 = "bfloat16"
float16 = _np.float16


def _is_float(dtype):
  if dtype is None:
    return False
  if isinstance(dtype, str):
    return dtype in ("float32", "float64", "bfloat16", "float16")
  try:
    return _np.issubdtype(_np.dtype(dtype), _np.floating)
  except TypeError:
    return False


def _dt(dtype):
  return _np.float64 if _is_float(dtype) else dtype


def asarray(x, dtype=None):
  a = _np.asarray(x, _dt(dtype))
  return a.astype(_np.float64) if _np.issubdtype(a.dtype, _np.floating) else a


array = asarray


def ones(shape, dtype=float32):
  return _np.ones(shape, _dt(dtype))


def zeros(shape, dtype=float32):
  return _np.zeros(shape, _dt(dtype))


def concatenate(arrays, axis=0):
  """numpy's, plus jax's handling of ONE array argument (its leading axis is the sequence: axes 0 and 1 are merged,
  also when that leaves zero rows - the "no other devices" case of the reference's all_gather on one device)."""
  if isinstance(arrays, _np.ndarray):
    assert axis == 0 and arrays.ndim >= 2
    return arrays.reshape((arrays.shape[0] * arrays.shape[1],) + arrays.shape[2:])
  return _np.concatenate(arrays, axis)


def clip(x, a_min=None, a_max=None, *, min=None, max=None):  # pylint: disable=redefined-builtin
  """jnp.clip: either bound may be missing (utils.py:280 clips from below only)."""
  lo = a_min if a_min is not None else min
  hi = a_max if a_max is not None else max
  x = _np.asarray(x)
  if lo is not None:
    x = _np.maximum(x, lo)
  if hi is not None:
    x = _np.minimum(x, hi)
  return x
