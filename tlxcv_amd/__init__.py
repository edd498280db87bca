"""tlxcv_amd — MI355X-native forward-pass engine behind the TensorLayerX / tlxcv API surface.

    import tlxcv_amd; tlxcv_amd.install()          # `import tensorlayerx`, `tlxcv.models`, `tlxcv.tasks`
    from tlxcv.models import resnet50              # now resolve to this package
    from tlxcv.tasks import ImageClassification

Everything computes in hand-written HIP kernels (tlxcv_amd/csrc -> libtlxmi.so, C-ABI in
include/tlxmi.h).  There is no CPU path: importing works anywhere, running needs a gfx950 device.
"""
import importlib
import sys

from . import _lib, engine  # noqa: F401
from .engine import set_precision, precision  # noqa: F401

__version__ = "0.1.0"


def install():
    """Alias `tensorlayerx` -> tlxcv_amd.tlx and `tlxcv.{models,tasks}` -> tlxcv_amd.{models,tasks}
    so the reference's demo scripts (demo/image_classification/predict.py:1-6) run unchanged."""
    from . import tlx
    sys.modules.setdefault("tensorlayerx", tlx)
    sys.modules.setdefault("tensorlayerx.nn", tlx.nn)
    sys.modules.setdefault("tensorlayerx.nn.initializers", tlx.nn.initializers)
    sys.modules.setdefault("tensorlayerx.ops", tlx.ops)
    sys.modules.setdefault("tensorlayerx.vision", tlx.vision)
    sys.modules.setdefault("tensorlayerx.vision.transforms", tlx.vision.transforms)
    sys.modules.setdefault("tensorlayerx.vision.transforms.utils", tlx.vision.transforms.utils)
    for sub in ("models", "tasks"):
        m = importlib.import_module(f"{__name__}.{sub}")
        sys.modules.setdefault(f"tlxcv.{sub}", m)
    import types
    pkg = sys.modules.setdefault("tlxcv", types.ModuleType("tlxcv"))
    pkg.models = sys.modules["tlxcv.models"]
    pkg.tasks = sys.modules["tlxcv.tasks"]
    if not hasattr(pkg, "__path__"):
        pkg.__path__ = []
    return tlx
