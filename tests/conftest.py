import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# One tuning-table cache for the whole test session (ccvpe_amd/tuning.py): the first model of a shape measures its launches,
# later handles - in this process or in a child - run exactly the same launches.  Hermetic: never the user's ~/.cache.
if "CCVPE_TUNE_CACHE" not in os.environ:
    import tempfile
    os.environ["CCVPE_TUNE_CACHE"] = os.path.join(tempfile.mkdtemp(prefix="ccvpe_tune_"), "tuning.txt")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_library():
    """Build (or reuse) the in-tree HIP library; hipcc cross-compiles gfx950 without a GPU."""
    from ccvpe_amd import build
    return build.build(verbose=False)
