"""N > 1 path on CPU: two processes launched the way the driver launches bench.py (python -m torch.distributed.run
--nproc-per-node 2 --master-addr 127.0.0.1 ...), bench.py started plainly with --gpus 2 (it spawns its ranks), and the
single-process form -- all on the oracle-backed stub engine; the ranks' host channel is griduniverse_amd/rendezvous.py."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from griduniverse_amd.parallel import shard_range, unpack_view

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range():
    assert [shard_range(262144, 8, g) for g in (0, 1, 7)] == [(0, 32768), (32768, 32768), (229376, 32768)]
    assert shard_range(10, 1, 0) == (0, 10)
    for bad in ((10, 3, 0), (0, 1, 0), (8, 2, 2), (8, 0, 0), (8, 2, -1)):
        with pytest.raises(ValueError):
            shard_range(*bad)


def test_unpack_view_layout():
    n, world = 5, 3
    blocks = np.arange(world * 3 * n, dtype=np.int32).reshape(world, 3 * n)
    obs, rew, don = unpack_view(blocks, n)
    for r in range(world):
        assert np.array_equal(obs[r * n:(r + 1) * n], blocks[r, :n])
        assert np.array_equal(rew[r * n:(r + 1) * n], blocks[r, n:2 * n])
        assert np.array_equal(don[r * n:(r + 1) * n], blocks[r, 2 * n:])


def test_two_rank_shards_and_gathered_view_under_torch_distributed_run(tmp_path):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.join(ROOT, 'tests', '_dist_worker.py'), str(tmp_path)]
    proc = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert proc.returncode == 0, proc.stdout.decode()[-3000:]
    results = [json.load(open(os.path.join(str(tmp_path), 'rank%d.json' % r))) for r in range(2)]
    assert [r['ids'] for r in results] == [[0, 2047], [2048, 4095]]
    for r in results:
        assert r['world'] == 2 and r['ok_shard'] and r['ok_view'] and r['ok_max'], r
    # the JSON line of bench.py's N = 2 flow (same script, oracle-backed stub engine, the host channel instead of RCCL)
    text = open(os.path.join(str(tmp_path), 'bench_line.json')).read()
    assert len(text) < 4096  # the driver parses ONE line and keeps only a few KB of stdout
    line = json.loads(text)
    assert line['n_gpus'] == 2 and line['scaling'] == 'weak' and line['engine'] == 'tests._oracle_engine.OracleEngine'
    assert line['config']['global_envs'] == 1024 and line['steps'] == 2 and line['timing']['blocks'] >= 3
    assert line['timing']['ms_per_step_min'] <= line['ms_per_step'] <= line['timing']['ms_per_step_max']
    assert abs(line['value'] - 1024 * 40 * 2 / (line['ms_per_step'] * 2 / 1e3)) < 1e-5 * line['value']
    assert line['rccl']['nranks'] == 2 and line['rccl']['view_equals_shards'] is True and line['rccl']['view_envs'] == 1024
    c4 = line['strong_c4']
    assert c4['scaling'] == 'strong' and c4['total_envs'] == 2048 and c4['envs_per_gpu'] == 1024 and c4['shards_equal_oracle'] is True
    assert len(line['per_rank']['value']) == 2 and line['roofline']['traffic_measured_by_child_runs'] is False
    assert line['roofline']['traffic'] is None  # (the committed PMC profile is of the default launch size only)
    assert line['bit_exact_vs_oracle'] is True and line['final_state_vs_oracle']['equal'] is True
    assert line['bit_exact_vs_reference_digest'] is None  # 512 envs x 40 steps is not the captured run
    other = line['other_modes']['stats_only']
    assert other['returns_vs_oracle'] is True and other['value'] > 0 and 'packed_rows' not in line['other_modes']  # (the stub has no packed rows)
    # everything that is not on the line: the side file, full precision
    detail = json.load(open(os.path.join(str(tmp_path), line['detail'])))
    assert abs(detail['value'] - line['value']) < 1e-5 * line['value'] and detail['metric'] == line['metric']
    assert 'device' in detail and 'trajectory_placement' in detail['roofline'] and 'topology' in detail
    assert detail['timing']['launches_per_block'] == 2 and len(detail['per_rank']['ms_per_step']) == 2


BENCH_ON_STUB = os.path.join(ROOT, 'tests', '_bench_stub.py')  # bench.main(..., engine_cls=OracleEngine)
STUB = ['--envs', '512', '--T', '40', '--steps', '2', '--warmup', '1',
        '--min-seconds', '0.02', '--c4-envs', '2048', '--detail', '']


def _one_json_line(proc):
    assert proc.returncode == 0, proc.stderr.decode()[-3000:]
    lines = [ln for ln in proc.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    assert len(lines[0]) < 4096, len(lines[0])
    return json.loads(lines[0])


def _scrubbed_env():
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    return env


def test_bench_started_plainly_with_gpus_2_spawns_its_ranks_and_prints_one_line():
    """`python bench.py --gpus 2` without torch.distributed.run (the way the driver starts --gpus 1): the script launches its two
    ranks itself and relays rank 0's line."""
    proc = subprocess.run([sys.executable, BENCH_ON_STUB, '--gpus', '2'] + STUB, env=_scrubbed_env(), cwd=ROOT,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    line = _one_json_line(proc)
    assert line['n_gpus'] == 2 and len(line['per_rank']['value']) == 2 and line['engine'] == 'tests._oracle_engine.OracleEngine'
    assert line['rccl']['nranks'] == 2 and line['rccl']['view_equals_shards'] is True and line['rccl']['view_envs'] == 1024
    assert line['strong_c4']['n_gpus'] == 2 and line['strong_c4']['envs_per_gpu'] == 1024 and line['strong_c4']['shards_equal_oracle'] is True
    assert line['config']['global_envs'] == 1024 and line['bit_exact_vs_oracle'] is True and line['final_state_vs_oracle']['equal'] is True
    assert abs(line['value'] - 1024 * 40 * 2 / (line['ms_per_step'] * 2 / 1e3)) < 1e-5 * line['value']


def test_bench_single_process_form_drives_two_engines_and_prints_one_line():
    """--single-process: one host process, one engine per device, launches enqueued device after device, the view through the
    comm_init_all form."""
    proc = subprocess.run([sys.executable, BENCH_ON_STUB, '--gpus', '2', '--single-process'] + STUB, env=_scrubbed_env(),
                          cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    line = _one_json_line(proc)
    assert line['n_gpus'] == 2 and line['mode'] == 'single-process' and len(line['per_rank']['value']) == 2
    assert line['config']['global_envs'] == 1024 and line['config']['devices'] == [0, 0]
    assert line['rccl']['nranks'] == 2 and line['rccl']['view_equals_shards'] is True and line['rccl']['view_envs'] == 1024
    assert line['strong_c4']['n_gpus'] == 2 and line['strong_c4']['total_envs'] == 2048 and line['strong_c4']['shards_equal_oracle'] is True
    assert line['bit_exact_vs_oracle'] is True and line['final_state_vs_oracle']['equal'] is True
    assert abs(line['value'] - 1024 * 40 * 2 / (line['ms_per_step'] * 2 / 1e3)) < 1e-5 * line['value']


def test_bench_and_the_sharded_product_path_import_no_torch():
    """north_star: host code has no PyTorch.  Importing bench.py, building a rendezvous and the sharded classes pulls in no torch."""
    code = ('import sys; sys.path.insert(0, %r); import bench; from griduniverse_amd import parallel, rendezvous; '
            'r = rendezvous.Rendezvous(0, 1); r.barrier(); assert r.reduce([1.0, 2.0], "MAX") == [1.0, 2.0]; '
            'assert "torch" not in sys.modules, "torch was imported"' % ROOT)
    proc = subprocess.run([sys.executable, '-c', code], env=_scrubbed_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert proc.returncode == 0, proc.stderr.decode()[-2000:]


def test_rendezvous_collectives_three_ranks_over_tcp_and_unix():
    """barrier / broadcast / gather / reduce of griduniverse_amd.rendezvous with three ranks, on both transports."""
    code = '''
import os, sys
sys.path.insert(0, %r)
from griduniverse_amd.rendezvous import Rendezvous
r = Rendezvous()
r.barrier()
assert r.broadcast_bytes(b"id-%%d" %% r.rank, 1) == b"id-1"
assert r.gather([float(r.rank), 7.0]) == [[0.0, 7.0], [1.0, 7.0], [2.0, 7.0]]
assert r.reduce([float(r.rank)], "MAX") == [2.0] and r.reduce([float(r.rank)], "MIN") == [0.0] and r.reduce([1.0], "SUM") == [3.0]
assert r.gather_bytes(bytes([r.rank]) * (r.rank + 1)) == [b"\\x00", b"\\x01\\x01", b"\\x02\\x02\\x02"]
r.close()
''' % ROOT
    for transport in ('auto', 'tcp'):
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        procs = [subprocess.Popen([sys.executable, '-c', code], stderr=subprocess.PIPE,
                                  env=dict(_scrubbed_env(), RANK=str(r), WORLD_SIZE='3', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                                           GU_RDZV=transport)) for r in (2, 0, 1)]
        for p in procs:
            err = p.communicate(timeout=120)[1]
            assert p.returncode == 0, err.decode()[-2000:]


def test_bench_finds_the_reference_digest_of_its_default_run():
    """bench.py compares its first launch with the sha256 the reference itself produced -- only when grid, seed, batch and
    length are those of the captured run."""
    import bench
    for name, N, T in (('c3', 65536, 1000), ('c4', 262144, 250), ('c2', 4096, 1000)):
        template, _ = bench.build_workload(name)
        seed = bench.WORKLOAD_SEED[name]
        digest = bench.reference_digest(name, template, seed, N, T, 0)
        assert isinstance(digest, str) and len(digest) == 64, name
        assert bench.reference_digest(name, template, seed, N, T, 64) is None
        assert bench.reference_digest(name, template, seed + 1, N, T, 0) is None
        assert bench.reference_digest(name, template, seed, N // 2, T, 0) is None


def test_native_chatter_inside_the_collective_check_never_reaches_stdout():
    """bench.py prints ONE JSON line on stdout.  RCCL writes a banner to the C-level stdout when a communicator comes up (and C
    stdio flushes it after the line when stdout is a pipe), and so may any other native library: whatever native code prints
    inside bench.native_stdout_to_stderr() must land on stderr, in order, and stdout must be usable again afterwards."""
    code = '''
import ctypes, sys
sys.path.insert(0, %r)
import bench
libc = ctypes.CDLL(None)
print("before")
with bench.native_stdout_to_stderr():
    libc.puts(b"native chatter")           # buffered by C stdio: would otherwise come out at exit, AFTER everything else
    print("python chatter inside")
    sys.stdout.flush()
print("after")
''' % ROOT
    proc = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert proc.returncode == 0, proc.stderr.decode()[-2000:]
    assert proc.stdout.decode().split() == ['before', 'after']
    err = proc.stderr.decode()
    assert 'native chatter' in err and 'python chatter inside' in err


def test_bench_cannot_be_pointed_at_another_engine_from_its_command_line():
    """The script that produces the contract number has no switch that swaps the thing measured: `--engine` is gone; the CPU
    tests reach the stub engine through tests/_bench_stub.py -> bench.main(argv, engine_cls=...)."""
    import bench
    with pytest.raises(SystemExit):
        bench.parse_args(['--engine', 'tests._oracle_engine:OracleEngine'])
    assert 'engine' not in vars(bench.parse_args([]))
    assert 'import_module' not in open(os.path.join(ROOT, 'bench.py')).read()


@pytest.mark.parametrize('form', ['ranks', 'single-process'])
def test_eight_gpu_preflight_on_the_stub_engine(form):
    """The driver's 8-GPU run, rehearsed without an 8-GPU node: `--gpus 8` started plainly (eight rank processes over the socket
    rendezvous) and with --single-process, config 4 at its full 262 144 envs = 32 768 per rank."""
    extra = ['--single-process'] if form == 'single-process' else []
    proc = subprocess.run([sys.executable, BENCH_ON_STUB, '--gpus', '8'] + extra + ['--envs', '256', '--T', '24', '--steps', '2', '--warmup', '1',
                                                                                    '--min-seconds', '0.02', '--c4-envs', '262144', '--detail', ''],
                          env=_scrubbed_env(), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    line = _one_json_line(proc)
    assert line['n_gpus'] == 8 and len(line['per_rank']['value']) == 8
    assert line['config']['global_envs'] == 8 * 256 and line['scaling'] == 'weak'
    assert line['strong_c4']['envs_per_gpu'] == 32768 and line['strong_c4']['total_envs'] == 262144 and line['strong_c4']['n_gpus'] == 8
    assert line['strong_c4']['shards_equal_oracle'] is True
    assert line['rccl']['nranks'] == 8 and line['rccl']['view_equals_shards'] is True and line['rccl']['view_envs'] == 8 * 256
    assert line['bit_exact_vs_oracle'] is True and line['final_state_vs_oracle']['equal'] is True


def test_a_rank_that_dies_takes_the_launch_down_at_once():
    """spawn_ranks watches EVERY child: rank 1 exiting with an error before it joins the rendezvous must end the launch within
    seconds with that exit code -- not leave rank 0 waiting in accept() for the rendezvous timeout with its GPU held."""
    import time
    env = dict(_scrubbed_env(), GU_TEST_DIE_RANK='1')
    t0 = time.time()
    proc = subprocess.run([sys.executable, BENCH_ON_STUB, '--gpus', '2'] + STUB, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          timeout=120)
    assert proc.returncode == 3 and time.time() - t0 < 60, (proc.returncode, time.time() - t0)
    assert proc.stdout.strip() == b'' and b'rank 1 exited with code 3' in proc.stderr


def test_a_collective_that_never_comes_back_costs_the_rccl_object_not_the_line():
    """RCCL with more than one rank has never run before the driver's own multi-GPU run.  The view check is the LAST thing bench.py
    does, behind a watchdog: if it hangs, rank 0 still prints its line (with the refusal in `rccl.error`) and every rank leaves with
    exit code 0."""
    import time
    env = dict(_scrubbed_env(), GU_TEST_HANG_RCCL='1', GU_RCCL_CHECK_TIMEOUT='3')
    t0 = time.time()
    proc = subprocess.run([sys.executable, BENCH_ON_STUB, '--gpus', '2'] + STUB, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          timeout=300)
    line = _one_json_line(proc)
    assert time.time() - t0 < 120
    assert line['n_gpus'] == 2 and line['value'] > 0 and line['strong_c4']['shards_equal_oracle'] is True
    assert line['rccl']['view_equals_shards'] is None and 'did not come back' in line['rccl']['error']
