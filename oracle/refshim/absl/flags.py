"""absl.flags stand-in: the reference's trainer modules DEFINE their flags at import time; nothing reads them here."""
class _Flags:
  def __getattr__(self, k):
    raise AttributeError(k)
FLAGS = _Flags()
def DEFINE_string(*a, **k): pass
def DEFINE_boolean(*a, **k): pass
def DEFINE_integer(*a, **k): pass
DEFINE_bool = DEFINE_boolean
