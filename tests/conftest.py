import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    """cuda:0 — gpu-marked tests FAIL (not skip) without a device: the product has no CPU path."""
    import torch
    assert torch.cuda.is_available(), "gpu-marked test needs an MI355X; none visible"
    import tlxcv_amd
    n = tlxcv_amd._lib.load().tlxmi_device_count()
    assert n >= 1, f"tlxmi_device_count() = {n}: no gfx950 device"
    return torch.device("cuda:0")


@pytest.fixture()
def fp32_mode():
    import tlxcv_amd
    tlxcv_amd.set_precision("fp32")
    yield
    tlxcv_amd.set_precision("fp16")


@pytest.fixture()
def fp16_mode():
    import tlxcv_amd
    tlxcv_amd.set_precision("fp16")
    yield
    tlxcv_amd.set_precision("fp16")


GOLDEN = os.path.join(REPO, "tests", "golden")
