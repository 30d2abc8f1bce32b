import numpy as np


def logsumexp(a, axis=None, keepdims=False):
  m = np.max(a, axis=axis, keepdims=True)
  out = np.log(np.sum(np.exp(a - m), axis=axis, keepdims=True)) + m
  return out if keepdims else np.squeeze(out, axis=axis)
