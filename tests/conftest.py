import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on a GPU box)")
    config.addinivalue_line("markers", "gpu_slow: the largest variants of a gpu test (A/B-only parametrisations, second full-size oracle forwards): "
                                       "skipped unless LS3D_GPU_SLOW=1, so that `pytest -m gpu` stays well inside the driver's time limit")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("LS3D_GPU_SLOW") == "1":
        return
    skip = pytest.mark.skip(reason="gpu_slow variant: LS3D_GPU_SLOW=1 python -m pytest tests -m gpu runs it")
    for item in items:
        if "gpu_slow" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _exact_f32_baseline():
    """every test starts from the exact-f32 arithmetic (the baseline the parity tolerances are written against) and says so when it
    wants one of the f32-grade plane modes; the library's own default is "bf16x6" (lidarseg3d_amd/ops.py)"""
    from lidarseg3d_amd import ops
    ops.set_precision("f32")
    yield
    ops.set_precision("f32")
