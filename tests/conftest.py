import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# A chip-resident solve whose hand-off times out is re-run on another algorithm and returns the same
# answer (lp_simplex_stats::fell_back says so).  In the tests that must never pass silently: with this set
# the library returns an error instead of falling back (tests that exercise the fallback unset it).
os.environ.setdefault("LP_RESIDENT_STRICT", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """One lp_context on cuda:0 for the whole GPU session (fails loudly without a GPU)."""
    from simplexmethod_amd import capi
    c = capi.Context(0)
    yield c
    c.close()
