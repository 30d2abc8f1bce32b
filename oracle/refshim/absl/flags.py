"""absl.flags stand-in (big_vision/utils.py imports it at module level; nothing on this path defines a flag)."""
class _Flags:
  def __getattr__(self, k):
    raise AttributeError(k)
FLAGS = _Flags()
