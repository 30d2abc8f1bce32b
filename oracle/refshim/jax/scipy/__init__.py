from . import special  # noqa: F401
