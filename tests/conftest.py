import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The suite's expectations (HBM budgets, residency, which context holds what) are written for ONE device; handles
# opened without a device argument use every visible GPU by default (pss_default_devices), so the session pins the
# default list to device 0.  The tests of the default list itself (test_default_device_list,
# test_unchanged_call_uses_the_default_device_list, tests/test_dist_gpu.py) set or unset the variable themselves.
os.environ.setdefault('PSS_DEVICES', '0')


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


@pytest.fixture
def search_env(monkeypatch):
    """Sets PSS_* switches of the search path for one test.  The library reads them once, not on
    every call (single-query latency), so a change must be followed by pss_reload_env()."""
    from pysubstringsearch_amd import _ffi

    def set_(**kv):
        for k, v in kv.items():
            if v is None:
                monkeypatch.delenv(k, raising=False)
            else:
                monkeypatch.setenv(k, str(v))
        _ffi.lib.pss_reload_env()

    yield set_
    monkeypatch.undo()
    _ffi.lib.pss_reload_env()
