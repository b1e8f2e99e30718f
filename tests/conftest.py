import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (str(ROOT), str(ROOT / "ebnerd-benchmark_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip():
    """The loaded C-ABI library on a box with a GPU (gpu-marked tests only)."""
    import torch

    assert torch.cuda.is_available(), "gpu-marked test without a visible GPU"
    from ebrec import _hip

    _hip.lib()
    torch.cuda.set_device(0)
    return _hip


@pytest.fixture(autouse=True)
def _release_device_temporaries(request):
    yield
    if request.node.get_closest_marker("gpu") is not None:
        from tests.hip_testutil import release_kept

        release_kept()
