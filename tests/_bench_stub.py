"""bench.py's whole flow on the oracle-backed stub engine -- the entry point of the CPU tests that start the script the way the
driver does (plainly, with --gpus N, or with --single-process).  bench.py's own command line can only measure
griduniverse_amd.Engine; TEST CODE lives here, and is what a plain --gpus N start of THIS script re-launches as its ranks."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from tests._oracle_engine import OracleEngine  # noqa: E402

if __name__ == '__main__':
    if os.environ.get('GU_TEST_DIE_RANK') is not None and os.environ.get('GU_TEST_DIE_RANK') == os.environ.get('RANK'):
        sys.exit(3)  # (tests/test_multiprocess.py: a rank that dies before it joins the rendezvous)
    sys.exit(bench.main(sys.argv[1:], engine_cls=OracleEngine, script=os.path.abspath(__file__)))
