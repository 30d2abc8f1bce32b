"""flax.linen.initializers stand-ins: the same DISTRIBUTIONS from a numpy Generator seeded by (seed, parameter path) -
not JAX's random bits (../README.md).  Signature as in jax: init(key, shape, dtype) -> array (float64 here)."""
import numpy as np


def _fans(shape, in_axis=-2, out_axis=-1):
  if len(shape) < 2:
    n = int(shape[0]) if len(shape) == 1 else 1
    return n, n
  rf = int(np.prod(shape)) // (shape[in_axis] * shape[out_axis])
  return shape[in_axis] * rf, shape[out_axis] * rf


def zeros(key, shape, dtype=None):
  return np.zeros(shape, np.float64)


def ones(key, shape, dtype=None):
  return np.ones(shape, np.float64)


zeros_init = lambda: zeros
ones_init = lambda: ones


def normal(stddev=1e-2):
  def init(key, shape, dtype=None):
    return key.generator().standard_normal(tuple(int(s) for s in shape)) * stddev
  return init


def variance_scaling(scale, mode, distribution, in_axis=-2, out_axis=-1):
  def init(key, shape, dtype=None):
    shape = tuple(int(s) for s in shape)
    fan_in, fan_out = _fans(shape, in_axis, out_axis)
    denom = {"fan_in": fan_in, "fan_out": fan_out, "fan_avg": (fan_in + fan_out) / 2}[mode]
    var = scale / max(1.0, denom)
    g = key.generator()
    if distribution == "uniform":
      lim = np.sqrt(3.0 * var)
      return g.uniform(-lim, lim, shape)
    if distribution == "truncated_normal":     # jax: normal truncated at +-2 sigma, rescaled to the requested variance
      x = g.standard_normal(shape)
      while True:
        bad = np.abs(x) > 2.0
        if not bad.any():
          break
        x[bad] = g.standard_normal(int(bad.sum()))
      return x * np.sqrt(var) / 0.87962566103423978
    return g.standard_normal(shape) * np.sqrt(var)
  return init


def xavier_uniform(in_axis=-2, out_axis=-1):
  return variance_scaling(1.0, "fan_avg", "uniform", in_axis, out_axis)


glorot_uniform = xavier_uniform


def lecun_normal(in_axis=-2, out_axis=-1):
  return variance_scaling(1.0, "fan_in", "truncated_normal", in_axis, out_axis)
