import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs the reference checkout at /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isfile("/root/reference/lib/chips/cchips.cpp")
    skip_ref = pytest.mark.skip(reason="reference checkout not present")
    for item in items:
        if "ref" in item.keywords and not have_ref:
            item.add_marker(skip_ref)


@pytest.fixture(autouse=True)
def _fresh_symbol_names():
    """mx.sym auto-names (`blockgrad0`, ...) come from a per-thread counter like MXNet's NameManager; start every
    test from zero so that assertions on generated names do not depend on test order."""
    from sniper_amd.mx import symbol
    symbol._counter().clear()
    yield
