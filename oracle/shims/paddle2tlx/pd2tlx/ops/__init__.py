from . import tlxops  # noqa: F401
