import os
import sys

import pytest

# the sgk_debug_* test hooks of libsgk.so answer only in a process that asked for them before the library was loaded (include/sgk.h)
os.environ["SGK_ENABLE_TEST_HOOKS"] = "1"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "safe-grid-agents_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # (re)build the native pieces when sources are newer than the in-tree .so files (hipcc cross-compiles on CPU boxes)
    try:
        from safe_grid_agents_amd import _lib

        _lib.build()
    except Exception as exc:  # the tests that need the library will fail loudly on load
        print("conftest: libsgk build skipped/failed: %r" % (exc,))
    try:
        from oracle import oracle as O

        O.build()
    except Exception as exc:
        print("conftest: oracle build skipped/failed: %r" % (exc,))


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
