import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as ge  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return ge.load_package()


@pytest.fixture(scope="session")
def orc():
    mod = ge.load_oracle()
    mod.lib()
    # (bit-identical for any thread count: oracle.set_threads; bench.py's cpu_baseline leg keeps 1 thread)
    mod.set_threads(min(32, len(os.sched_getaffinity(0))))
    return mod


@pytest.fixture(scope="session")
def hiplib(pkg):
    """The loaded C-ABI library; building it is part of __graft_entry__.build()."""
    if not os.path.exists(pkg.lib_path()):
        ge.build()
    return pkg.load_library()
