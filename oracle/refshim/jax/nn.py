"""jax.nn pieces the model files reach through flax.linen (gelu, softmax)."""
import numpy as np


def gelu(x, approximate=True):
  assert approximate, "the reference calls nn.gelu with its default (tanh form)"
  return 0.5 * x * (1.0 + np.tanh(np.sqrt(2.0 / np.pi) * (x + 0.044715 * x ** 3)))


def softmax(x, axis=-1):
  m = np.max(x, axis=axis, keepdims=True)
  e = np.exp(x - m)
  return e / np.sum(e, axis=axis, keepdims=True)


def log_sigmoid(x):
  """-softplus(-x), the stable form."""
  x = np.asarray(x, np.float64)
  return np.minimum(x, 0.0) - np.log1p(np.exp(-np.abs(x)))


def log_softmax(x, axis=-1):
  x = np.asarray(x, np.float64)
  m = np.max(x, axis=axis, keepdims=True)
  return x - m - np.log(np.sum(np.exp(x - m), axis=axis, keepdims=True))
