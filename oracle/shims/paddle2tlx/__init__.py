from . import pd2tlx  # noqa: F401
