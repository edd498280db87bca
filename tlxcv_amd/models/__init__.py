from .classification import *  # noqa: F401,F403
from .detection import *  # noqa: F401,F403
