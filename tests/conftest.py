import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, 'tests', 'golden')


def _ensure_built():
    """The shared libraries are build products (git-ignored).  Build them when a checkout has none yet, so
    that the suite does not depend on __graft_entry__.build() having run first.  hipcc cross-compiles for
    gfx950 without a GPU."""
    import subprocess
    lib = os.path.join(REPO, 'meshflow_amd', 'libmeshflow_hip.so')
    if not os.path.exists(lib):
        subprocess.run(['make', '-j8', '-C', os.path.join(REPO, 'meshflow_amd', 'csrc')], check=True)
    if not os.path.exists(os.path.join(REPO, 'oracle', '_build', 'liboracle.so')):
        subprocess.run(['make', '-C', os.path.join(REPO, 'oracle')], check=True)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    _ensure_built()


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
