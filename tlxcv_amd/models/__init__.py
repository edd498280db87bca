from .classification import *  # noqa: F401,F403
