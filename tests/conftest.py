import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    if os.environ.get("TAPER_FORCE_NO_GPU"):
        return False
    return os.path.exists("/dev/kfd") and any(
        p.name.startswith("renderD") for p in Path("/dev/dri").glob("renderD*")) if os.path.exists("/dev/dri") else False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _leave_no_tape_behind(request):
    """a GPU test that fails half-way must not leave its tape / mode switches to the tests behind it (one failure then read as dozens)"""
    yield
    if "gpu" in request.keywords and _has_gpu():
        try:
            import taper_amd as T
            T.Tape.reset()
            T.set_full_backward(False)
        except Exception:
            pass
