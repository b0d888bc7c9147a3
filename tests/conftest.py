import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_ctypes as oc
    oc.lib()
    return oc


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def _torch_gpu_first(request):
    """Tests that share the process between torch and libexpann_hip: let torch initialise its HIP
    context before the library creates streams (late torch initialisation after heavy use of the
    library was observed to report 'No HIP GPUs are available' on the GPU box)."""
    if any(item.get_closest_marker("gpu") for item in request.session.items):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass
    yield
