"""jax.random stand-in: a key is (seed, path-salt); initialisers turn it into a numpy Generator."""
import zlib

import numpy as np


class Key:
  def __init__(self, seed, salt=""):
    self.seed, self.salt = int(seed), salt

  def fold(self, what):
    return Key(self.seed, f"{self.salt}|{what}")

  def generator(self):
    return np.random.default_rng([self.seed, zlib.crc32(self.salt.encode())])


def PRNGKey(seed):
  return Key(seed)


key = PRNGKey


def split(k, num=2):
  return [k.fold(f"split{i}") for i in range(num)]


def fold_in(k, data):
  return k.fold(f"fold{data}")


def beta(k, a, b, shape=None, dtype=None):
  """A Beta(a, b) draw from the stand-in generator of this key (utils.py:1149: the mixup coefficient).  The STREAM is not
  JAX's; what runs downstream of the draw is."""
  del dtype
  return k.generator().beta(a, b, size=shape)
