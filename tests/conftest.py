import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # a fresh checkout has no built artefacts (they are git-ignored): build them once
    lib = os.path.join(ROOT, 'pysubstringsearch_amd', 'libpss.so')
    if not os.path.exists(lib):
        import subprocess
        subprocess.run(['make', '-C', os.path.join(ROOT, 'pysubstringsearch_amd', 'csrc'), '-j4'], check=True,
                       capture_output=True)


@pytest.fixture(scope='session')
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope='session')
def pss():
    import pysubstringsearch_amd as P
    return P
