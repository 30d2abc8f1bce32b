"""ml_collections stand-in: imported by big_vision/utils.py at module level, unused on this path."""
class ConfigDict(dict):
  __getattr__ = dict.__getitem__
