"""The gathered view (csrc/gu_comm.hip) and the N > 1 forms of bench.py on the one GPU of the box: which RCCL is loaded, several ranks through a test double of RCCL, bench.py with two ranks and with --single-process."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from griduniverse_amd import Engine, GridSpec, _lib
from griduniverse_amd.engine import Engine
from griduniverse_amd.grid import GridSpec
import griduniverse_amd as gua

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(tmp_path, *args):
    """Returns (the ONE line of stdout, under 4 KB; the side file)."""
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    side = os.path.join(str(tmp_path), 'bench_detail.json')
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--detail', side] + list(args), cwd=ROOT, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=1200)
    assert proc.returncode == 0, proc.stderr.decode()[-3000:]
    lines = [ln for ln in proc.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4096, lines
    return json.loads(lines[0]), json.load(open(side))


SMALL = ['--envs', '16384', '--T', '200', '--steps', '3', '--warmup', '1', '--min-seconds', '0.05', '--c4-envs', '16384', '--no-cpu-baseline', '--no-live-traffic']


@pytest.mark.parametrize('order', ['gu_first', 'torch_first'])
def test_rccl_is_taken_from_the_rocm_stack_libgu_runs_on(order):
    """A process may hold two ROCm stacks (the system one and the copy a PyTorch wheel bundles).  Whichever order they are
    loaded in, the gathered view must come up: gu_comm.hip takes the librccl next to the libamdhip64 that serves its own HIP
    calls.  (A caller that imports torch brings that second stack; bench.py itself no longer does.)"""
    import subprocess
    import sys
    code = '''
import sys
sys.path.insert(0, %r)
order = %r
if order == 'torch_first':
    import torch, torch.distributed
import numpy as np
import griduniverse_amd as gua
eng = gua.Engine(4096, gua.GridSpec(8, 8, [0], [63], [], []), seed=1)
if order == 'gu_first':
    import torch, torch.distributed
eng.reset()
eng.rollout(50, 'uniform', True, False)
eng.comm_init(1, 0, gua.Engine.comm_unique_id())
view = eng.allgather_view()
own = eng.read_outputs()
assert all(np.array_equal(a, b) for a, b in zip(view, own))
eng.comm_destroy()
eng.close()
print('VIEW-OK')
''' % (__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))), order)
    out = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert out.returncode == 0 and b'VIEW-OK' in out.stdout, out.stdout.decode()[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize('nranks,N', [(2, 4096), (4, 1000), (8, 32768)])
def test_gathered_view_with_several_ranks_on_one_gpu_through_a_test_double_of_rccl(tmp_path, nranks, N):
    """RCCL refuses two ranks on one device and only one device is ever at hand, so gu_comm_init with nranks > 1 and the
    rank-major -> env-major unpack of gu_allgather_view had never run on hardware.  Here GU_RCCL_LIB points libgu at a test
    double (tests/c_abi/fake_rccl.hip: ranks = threads of one process, all-gather = rendezvous + device-to-device copies);
    every rank is an engine holding the shard [rank * N, (rank + 1) * N) of one batch.  Every rank's view must equal the
    single-engine batch of nranks * N envs (8 x 32 768 = config 4).  Own process: the library is chosen once per process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = tmp_path / 'libfake_rccl.so'
    build = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-O2', '-o', str(lib),
                            os.path.join(root, 'tests', 'c_abi', 'fake_rccl.hip')], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert build.returncode == 0, build.stdout.decode()[-2000:]
    code = '''
import sys, threading
sys.path.insert(0, %r)
import numpy as np
import griduniverse_amd as gua
nranks, N = %d, %d
lava = [16 + 32 * r for r in range(24)]
spec = gua.GridSpec(32, 32, [0], [1023], lava, [])
whole = gua.Engine(nranks * N, spec, seed=9)
whole.reset()
whole.rollout(150, 'uniform', True, False)
want = whole.read_outputs()
shards = [gua.Engine(N, spec, seed=9, env_id0=r * N) for r in range(nranks)]
for e in shards:
    e.reset()
    e.rollout(150, 'uniform', True, False)
uid = gua.Engine.comm_unique_id()
views, errors = [None] * nranks, []
def run(r):
    try:
        shards[r].comm_init(nranks, r, uid)
        views[r] = shards[r].allgather_view()
        views[r] = shards[r].allgather_view()  # (a second gather reuses the communicator)
        shards[r].comm_destroy()
    except Exception as exc:
        errors.append((r, repr(exc)))
threads = [threading.Thread(target=run, args=(r,)) for r in range(nranks)]
[t.start() for t in threads]
[t.join(120) for t in threads]
assert not errors, errors
for r in range(nranks):
    assert views[r] is not None, r
    for got, exp, name in zip(views[r], want, ('obs', 'reward', 'done')):
        assert got.shape == (nranks * N,) and np.array_equal(got, exp), (r, name)
assert want[2].sum() > 0
print('VIEW-OK')
''' % (root, nranks, N)
    env = dict(os.environ, GU_RCCL_LIB=str(lib))
    out = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, env=env)
    assert out.returncode == 0 and b'VIEW-OK' in out.stdout, out.stdout.decode()[-3000:]


def test_bench_started_plainly_with_two_ranks_on_the_one_gpu(tmp_path):
    """`python bench.py --gpus 2`, no launcher: two rank processes (sharing device 0 here), rank 0's one line; RCCL itself refuses
    two ranks on one device, which the line reports instead of dying."""
    line, detail = _bench(tmp_path, '--gpus', '2', *SMALL)
    assert line['n_gpus'] == 2 and len(line['per_rank']['value']) == 2 and line['engine'] == 'griduniverse_amd.engine.Engine'
    assert line['bit_exact_vs_oracle'] is True and line['final_state_vs_oracle']['equal'] is True
    assert line['rccl']['nranks'] == 2 and (line['rccl'].get('view_equals_shards') is True or 'error' in line['rccl'])
    assert line['strong_c4']['n_gpus'] == 2 and line['strong_c4']['shards_equal_oracle'] is True
    assert detail['device']['arch'].startswith('gfx950') and detail['roofline']['trajectory_placement'] is not None
    if 'error' in line['rccl']:  # (what the real RCCL says to two ranks on one device: kept for profiles/)
        print('RCCL with 2 ranks on one device: ' + line['rccl']['error'])


def test_bench_single_process_form_on_the_one_gpu(tmp_path):
    """--single-process --gpus 2: one process, two engines (both on device 0 here), launches enqueued engine after engine; the
    gu_comm_init_all view needs one device per engine and is reported as refused on this box."""
    line, detail = _bench(tmp_path, '--gpus', '2', '--single-process', *SMALL)
    assert line['n_gpus'] == 2 and line['mode'] == 'single-process' and line['config']['devices'] == [0, 0]
    assert len(line['per_rank']['value']) == 2 and line['bit_exact_vs_oracle'] is True and line['final_state_vs_oracle']['equal'] is True
    assert line['rccl']['nranks'] == 2 and ('error' in line['rccl'] or line['rccl']['view_equals_shards'] is True)
    assert line['strong_c4']['shards_equal_oracle'] is True and len(detail['roofline']['trajectory_placement']) == 2
    if 'error' in line['rccl']:
        print('ncclCommInitAll with the same device twice: ' + line['rccl']['error'])
    line, _ = _bench(tmp_path, '--gpus', '1', '--single-process', '--gather-view', *SMALL)
    assert line['n_gpus'] == 1 and line['rccl']['view_equals_shards'] is True and line['rccl']['nranks'] == 1
