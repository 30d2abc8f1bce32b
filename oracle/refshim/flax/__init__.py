"""flax stand-in (see ../README.md)."""
from . import core, serialization  # noqa: F401
from . import linen  # noqa: F401
