"""The pin, re-checked by a command: every committed fixture under tests/golden/ is regenerated from the REAL reference
(/root/reference under the stub gym, tests/golden/make_golden.py --check) into a temporary directory and compared with the
committed one by content -- .json by equality of the parsed data, .npz array by array, byte-wise.

Runs only where the reference is (the build container); skipped everywhere else, and not a `gpu` test: it never travels to or
runs on the GPU box, and nothing of the reference does.  ~3 minutes: 12 M reference `step()` calls among others.  The two
full-size digests (65 536 x 1000 on config 3, 262 144 x 250 on config 4: 131 M reference steps, ~7 min) are left to
`python tests/golden/make_golden.py --check big traj`."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('GU_REFERENCE', '/root/reference')


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'core', 'envs')), reason='the reference is not here (it never leaves the build container)')
def test_every_fixture_regenerates_from_the_reference_identically():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1', MPLBACKEND='Agg')
    run = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'golden', 'make_golden.py'), '--check'], cwd=ROOT, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    tail = '\n'.join(run.stdout.splitlines()[-25:])
    assert run.returncode == 0, tail
    assert 'all identical to the committed ones' in tail, tail
