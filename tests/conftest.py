import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_count():
    try:
        from flexs_amd import _native

        return _native.lib().fx_device_count()
    except Exception:  # noqa: BLE001
        return 0


def pytest_collection_modifyitems(config, items):
    if _gpu_count() > 0:
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _build_native():
    """The CPU suite needs the shared libraries (ABI/export checks, packing hooks)."""
    import __graft_entry__ as g

    g.build(quiet=True)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
