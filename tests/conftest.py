import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session', autouse=True)
def _native_pieces_are_built():
    """The in-tree libgu.so / libgu_oracle.so normally travel with the snapshot (built by __graft_entry__.build());
    if a checkout arrives without them, build them once before any test needs them.  This is test plumbing: the
    product itself never builds or falls back -- it raises GuError when the library is missing."""
    from griduniverse_amd import _lib
    if _lib.is_stale():  # missing, or built from sources other than the ones on disk
        _lib.build()
    from oracle import c_oracle
    c_oracle.build()


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture
def gu_option():
    """set(name, value): process-wide default of a libgu launch-shape / search option for the rest of the test (None = the
    built-in default); whatever the test set is put back afterwards.  (Rounds 1 and 2 flipped these through environment
    variables; the library no longer reads any -- include/gu.h "options".)"""
    from griduniverse_amd import _lib
    touched = {}

    def set_option(name, value):
        if name not in touched:
            touched[name] = None  # tests start from the built-in defaults
        _lib.set_default_option(name, value)

    yield set_option
    for name in touched:
        _lib.set_default_option(name, None)
