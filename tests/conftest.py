import os
import sys

import pytest

# The suite drives route-forcing and fault-injection knobs (CNIIC_TEST_*, CNIIC_DBG_*, CNIIC_KM_SUP, ...): they exist only in the
# testing build of the library (cniic_amd/libcniic_hip_testing.so, -DCNIIC_TESTING).  Tests that must see the RELEASE library
# (tests/test_abi.py: no test hooks in it; __graft_entry__.smoke(), bench.py) load it by path or in a child process.
os.environ.setdefault("CNIIC_USE_TESTING_LIB", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
