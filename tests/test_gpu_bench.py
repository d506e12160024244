"""bench.py on the device, as the driver runs it (one JSON line on stdout), at sizes that finish in seconds: the contract
fields, the checks the line carries about itself, and the 1-rank RCCL view."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(tmp_path, *args):
    """Returns (the line, the side file)."""
    side = os.path.join(str(tmp_path), 'bench_detail.json')
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--detail', side] + list(args), cwd=ROOT, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=900)
    assert proc.returncode == 0, proc.stderr.decode()[-3000:]
    lines = [ln for ln in proc.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines  # exactly ONE line on stdout
    assert len(lines[0]) < 4096, len(lines[0])  # (the driver keeps a few KB of stdout: BENCH_r05's 20 KB line went unparsed)
    line = json.loads(lines[0])
    assert line['detail'] == side
    return line, json.load(open(side))


def test_default_workload_line_carries_the_contract_and_its_own_checks(tmp_path):
    """Config 3 at full size (65 536 envs x 1000 steps per launch), few launches: metric / unit / config as BASELINE.json names
    them, value = envs x steps x launches / time, the roofline object consistent with the HIP-event launch time, the first
    launch equal to the reference's digest, the final state equal to the oracle, and the CPU baseline beside it."""
    line, detail = run_bench(tmp_path, '--gpus', '1', '--steps', '5', '--warmup', '2', '--min-seconds', '0.05', '--c4-envs', '16384', '--gather-view')
    base = json.load(open(os.path.join(ROOT, 'BASELINE.json')))
    assert line['metric'] == base['metric'] and line['unit'] == 'env-steps/s' and line['n_gpus'] == 1
    assert line['steps'] == 5 and line['warmup'] == 2 and line['higher_is_better'] is True and line['scaling'] == 'weak'
    assert line['vs_baseline'] is None and line['dtype'] == 'int32' and line['data'] == 'synthetic'
    assert line['config']['workload'].startswith('c3: 65536 envs per GPU') and line['config']['envs_per_gpu'] == 65536
    assert abs(line['value'] - 65536 * 1000 / (line['ms_per_step'] / 1e3)) < 1e-5 * line['value']
    roof = line['roofline']
    assert roof['bound'] == 'hbm' and roof['unit'] == 'GB/s' and roof['peak'] == 8000.0
    # HBM traffic measured for THIS run: two short rocprofv3 --pmc child runs of the script (WRITE_SIZE, FETCH_SIZE) after the timed region
    assert roof['traffic_measured_by_child_runs'] is True and detail['roofline']['traffic_live']['dispatches_counted'] >= 4
    assert 0.98 < roof['traffic'] / roof['algorithmic_bytes_per_launch'] < 1.03 and abs(roof['traffic_over_algorithmic'] - roof['traffic'] / roof['algorithmic_bytes_per_launch']) < 1e-5
    assert roof['algorithmic_bytes_per_launch'] == 12 * 65536 * 1000
    assert abs(roof['achieved'] - roof['algorithmic_bytes_per_launch'] / (roof['launch_ms'] / 1e3) / 1e9) < 1e-5 * roof['achieved']
    assert abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-5 and 0.3 < roof['frac'] < 1.0
    assert roof['launch_ms'] <= line['ms_per_step'] * 1.02  # the kernel cannot take longer than the step that contains it
    assert line['bit_exact_vs_reference_digest'] is True and line['bit_exact_vs_oracle'] is True
    assert line['final_state_vs_oracle']['equal'] is True and line['final_state_vs_oracle']['launches'] == line['timing']['launches_total'] == detail['timing']['launches_total']
    assert line['engine'] == 'griduniverse_amd.engine.Engine'
    cpu = line['cpu_baseline']
    assert cpu['kind'] == 'port' and cpu['cores'] == 1 and cpu['value'] > 1e5 and cpu['unit'] == 'env-steps/s'
    assert line['rccl']['nranks'] == 1 and line['rccl']['view_equals_shards'] is True
    c4 = line['strong_c4']
    assert c4['scaling'] == 'strong' and c4['total_envs'] == 16384 and c4['shards_equal_oracle'] is True
    other = line['other_modes']
    assert other['stats_only']['returns_vs_oracle'] is True and other['stats_only']['value'] > line['value']
    assert detail['other_modes']['packed_rows']['bytes_per_env_step'] == 4 and 0.1 < other['packed_rows']['frac_of_hbm_peak'] < 1.0
    assert detail['other_modes']['rollout_sample_policy_traj']['bytes_per_env_step'] == 12 and 0.3 < other['rollout_sample_policy_traj']['frac_of_hbm_peak'] < 1.0
    # frac_wall: the same bytes over the wall time per launch the driver's clock sees
    assert abs(roof['frac_wall'] - 12 * 65536 * 1000 / (line['ms_per_step'] / 1e3) / 1e9 / 8000.0) < 1e-5 and roof['frac_wall'] <= roof['frac'] * 1.02
    # every other BASELINE config beside the headline, each with its own parity bit
    cfg, full = line['configs'], detail['configs']
    assert set(cfg) == set(full) == {'c2', 'c4_shard', 'c5', 'c3_distinct_1024', 'c3_distinct_65536'}
    for name in cfg:
        assert cfg[name]['bit_exact'] is True and cfg[name]['env_steps_per_s'] > 1e9 and full[name]['bound'] and full[name]['check'], name
    assert full['c2']['workload'].startswith('c2: 4096 envs') and full['c4_shard']['workload'].startswith('c4 shard 1 of 8: 32768 envs')
    assert cfg['c5']['form'] == 'per-XCD' and cfg['c5']['us_per_round'] < 3.0
    for name in ('c2', 'c4_shard', 'c3_distinct_1024', 'c3_distinct_65536'):
        assert full[name]['hbm_gbps'] < 8000.0 and 0.0 < cfg[name]['frac_of_hbm_peak'] < 1.0
    # the shard's launch split into a fixed part and a per-step slope (T = 1000 against T = 4000)
    # (a slope from two noisy launch times: 0.84 .. 0.96 on the boxes of round 6; the bounds only catch nonsense)
    assert 0.5 < cfg['c4_shard']['asymptote_frac_of_hbm_peak'] < 1.25 and -10.0 < cfg['c4_shard']['fixed_us_per_launch'] < 25.0
    # one maze per env streams its rows like the shared maze does (round 5: 0.53 of the peak on private byte planes)
    assert cfg['c3_distinct_65536']['frac_of_hbm_peak'] > 0.65 and cfg['c3_distinct_1024']['frac_of_hbm_peak'] > 0.65  # (measured 0.78 .. 0.82 / 0.84 .. 0.91)
    topo = detail['topology']
    assert topo['hip_device_count'] >= 1 and topo['devices'][0]['pci'] and topo['rccl_library']
    assert detail['device'] and detail['roofline']['store_pacing'] and detail['roofline']['trajectory_placement']


def test_other_workloads_and_switches(tmp_path):
    """Config 2 (4096 envs, 8x8) with the checks, and a run with everything optional switched off."""
    line, _ = run_bench(tmp_path, '--workload', 'c2', '--envs', '4096', '--steps', '3', '--warmup', '1', '--min-seconds', '0.02', '--no-strong-c4',
                     '--no-cpu-baseline', '--no-live-traffic')
    assert line['roofline']['traffic_measured_by_child_runs'] is False and line['roofline']['traffic'] is None  # (the committed profile is of the default launch)
    assert line['config']['envs_per_gpu'] == 4096 and line['bit_exact_vs_reference_digest'] is True  # 4096 x 1000 is a captured run
    assert line['final_state_vs_oracle']['equal'] is True and 'cpu_baseline' not in line and 'strong_c4' not in line
    line, _ = run_bench(tmp_path, '--steps', '2', '--warmup', '1', '--min-seconds', '0.02', '--no-strong-c4', '--no-cpu-baseline', '--no-checks',
                     '--no-other-modes', '--envs', '1000', '--T', '77', '--no-live-traffic', '--no-configs')
    assert 'other_modes' not in line and 'configs' not in line and 'bit_exact_vs_reference_digest' not in line and line['value'] > 0
