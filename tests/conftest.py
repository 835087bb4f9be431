import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "bx-python_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than ~20 s on CPU")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_trees():
    return load_golden("ivtree_random.json")["cases"]


@pytest.fixture(scope="session")
def golden_bitsets():
    return load_golden("binnedbitset_ops.json")


@pytest.fixture(scope="session")
def golden_cli():
    return load_golden("cli/expected.json")


@pytest.fixture(scope="session")
def golden_scale():
    return load_golden("scale.json")["points"]


@pytest.fixture(scope="session")
def golden_scale_doc():
    return load_golden("scale.json")


def has_gpu():
    """(through libbxmi, not torch: importing torch first would make its bundled HIP runtime the process's first one, and the tests
    drive libbxmi -- linked against /opt/rocm -- and, through it, the system's RCCL)"""
    try:
        from bxmi import _ffi

        return _ffi.device_count() > 0
    except Exception:
        return False
