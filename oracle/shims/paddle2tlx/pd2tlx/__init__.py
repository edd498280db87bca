from . import ops, utils  # noqa: F401
