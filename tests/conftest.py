import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)


# K2 renders the last blocks of a long narrow-bus launch with several short workgroups per block (zl_launch_render: windows of 2048 blocks and more).
# The test tier's batches are shorter: lower the threshold so that every narrow-bus batch scene of the tier walks both forms of the launch -- the
# blocks in front with one workgroup each, the tail split -- and tests/test_k2_tail.py holds the cases at the shipped threshold.
os.environ.setdefault("ZL_K2_TAIL_MIN_BLOCKS", "24")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build the in-tree native pieces the CPU tier needs (engine .so is cross-compiled, not run)."""
    from libzl_amd import build
    build.build_oracle()
    build.build_cpu_harness()
    build.build_engine()
    return True
