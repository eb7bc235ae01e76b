import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_library():
    """Build (or reuse) the in-tree HIP library; hipcc cross-compiles gfx950 without a GPU."""
    from ccvpe_amd import build
    return build.build(verbose=False)
