import os
import sys

import pytest

# the builders' device workspace comes from a cache of reused blocks (sigax_index_build.hip: DevPool); in the tests every
# block is filled with 0xA5 before it is handed out, so that nothing can lean on hipMalloc's cleared memory unnoticed
os.environ.setdefault("SIGAX_POOL_POISON", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU test taking more than a few seconds")


@pytest.fixture(scope="session")
def tmp_session(tmp_path_factory):
    return tmp_path_factory.mktemp("siga")


@pytest.fixture(scope="session", autouse=True)
def _native_libs():
    """Build (if stale) the product library and the oracle before any test runs."""
    from oracle import pyoracle
    from siga_amd import build as sbuild
    sbuild.build_libsigax()
    pyoracle.build()
