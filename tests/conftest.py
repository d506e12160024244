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


def _session_options():
    """GU_TEST_OPTIONS="rollout_rows=1,traj_layout=1": process-wide defaults of libgu's launch-shape options for the whole
    session (tools/gpu_soak_switches.sh, gpu_fuzz.sh run the suite under several sets).  TEST PLUMBING: the product library reads
    no such switch from the environment (include/gu.h "options"); this translates the variable into gu_set_option(NULL, ...)."""
    out = {}
    for item in filter(None, (x.strip() for x in os.environ.get('GU_TEST_OPTIONS', '').split(','))):
        name, _, value = item.partition('=')
        out[name.strip()] = int(value)
    return out


@pytest.fixture(scope='session', autouse=True)
def _session_option_defaults(_native_pieces_are_built):
    wanted = _session_options()
    if wanted:
        from griduniverse_amd import _lib
        for name, value in wanted.items():
            _lib.set_default_option(name, value)
        in_force = {name: _lib.get_default_option(name) for name in wanted}
        assert in_force == wanted, (in_force, wanted)
        sys.stderr.write('[conftest] GU_TEST_OPTIONS in force (gu_get_option): %s\n' % ', '.join('%s=%d' % kv for kv in sorted(in_force.items())))
    yield


def pytest_report_header(config):
    wanted = _session_options()
    return 'GU_TEST_OPTIONS: ' + (', '.join('%s=%d' % kv for kv in sorted(wanted.items())) if wanted else '(none: the default dispatch)')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture
def gu_option():
    """set(name, value): process-wide default of a libgu launch-shape / search option for the rest of the test (None = the
    built-in default); whatever the test set is put back afterwards.  (Rounds 1 and 2 flipped these through environment
    variables; the library no longer reads any -- include/gu.h "options".)"""
    from griduniverse_amd import _lib
    session = _session_options()
    touched = set()

    def set_option(name, value):
        touched.add(name)
        _lib.set_default_option(name, value)

    yield set_option
    for name in touched:
        _lib.set_default_option(name, session.get(name))  # (the built-in default, or this session's GU_TEST_OPTIONS value)
