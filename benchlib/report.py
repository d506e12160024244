"""Reporting of bench.py: ONE compact JSON line on stdout (under 4 KB: the driver parses it) and everything else -- timing
spread, store pacing, trajectory placement, device and topology, per-config detail -- in a side file (bench_detail.json)."""
import ctypes
import glob
import json
import os
import sys

import numpy as np

import griduniverse_amd as gua
from griduniverse_amd import _lib

LINE_LIMIT = 4096


def topology_block(engine_cls):
    """What the node looks like, for the first run on more than one GPU to be self-diagnosing: HIP's device count, every
    device's PCI id, the xGMI link matrix as sysfs (or rocm-smi) shows it, the RCCL library the gathered view would load."""
    out = {}
    try:
        n = _lib.device_count() if engine_cls is gua.Engine else 1
        out['hip_device_count'] = n
        out['devices'] = []
        for d in range(n):
            info = engine_cls.device_info(d) if hasattr(engine_cls, 'device_info') else {}
            out['devices'].append({k: info.get(k) for k in ('name', 'arch', 'pci', 'cus') if k in info})
    except Exception as err:  # noqa: BLE001 -- reporting only
        out['error'] = str(err)
    links = {}
    for path in sorted(glob.glob('/sys/class/kfd/kfd/topology/nodes/*/io_links/*/properties')):
        try:
            props = dict(line.split(None, 1) for line in open(path).read().splitlines() if ' ' in line)
        except OSError:
            continue
        if props.get('type', '').strip() == '11':  # HSA_IOLINK_TYPE_XGMI
            node = path.split('/nodes/')[1].split('/')[0]
            links.setdefault(node, []).append(dict(to=props.get('node_to', '').strip(), weight=props.get('weight', '').strip(),
                                                   max_bandwidth=props.get('max_bandwidth', '').strip()))
    out['xgmi_links_by_kfd_node'] = links or None
    out['xgmi_hives'] = sorted({open(p).read().strip() for p in glob.glob('/sys/class/drm/card*/device/xgmi_hive_info/xgmi_hive_id')
                                if os.access(p, os.R_OK)}) or None
    rccl = os.environ.get('GU_RCCL_LIB') or '/opt/rocm/lib/librccl.so'
    out['rccl_library'] = os.path.realpath(rccl) if os.path.exists(rccl) else None
    out['visible_devices_env'] = {k: os.environ[k] for k in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES') if k in os.environ}
    return out



def device_block(engine_cls, device):
    """What the device looked like during the run: gu_device_info (name, arch, CUs, clocks as HIP reports them) plus the sysfs
    view of the same PCI function -- current sclk / mclk, power cap, memory and compute partition -- so that a slow run can be
    told from a differently configured box."""
    if not hasattr(engine_cls, 'device_info'):
        return None
    try:
        info = dict(engine_cls.device_info(device))
    except Exception as err:  # noqa: BLE001 -- reporting only
        return {'error': str(err)}
    pci = str(info.get('pci', '')).lower()
    base = '/sys/bus/pci/devices/' + pci
    sysfs = {}

    def read(rel):
        try:
            with open(os.path.join(base, rel)) as f:
                return f.read().strip()
        except OSError:
            return None

    if pci and os.path.isdir(base):
        for key, rel in (('memory_partition', 'current_memory_partition'), ('compute_partition', 'current_compute_partition'),
                         ('perf_level', 'power_dpm_force_performance_level'), ('vbios', 'vbios_version'),
                         ('gpu_busy_percent', 'gpu_busy_percent'), ('mem_busy_percent', 'mem_busy_percent')):
            v = read(rel)
            if v is not None:
                sysfs[key] = v
        for key, rel in (('sclk', 'pp_dpm_sclk'), ('mclk', 'pp_dpm_mclk'), ('fclk', 'pp_dpm_fclk')):
            v = read(rel)
            if v is not None:
                levels = [ln.strip() for ln in v.splitlines() if ln.strip()]
                sysfs[key + '_levels'] = levels
                sysfs[key + '_current'] = next((ln.rstrip(' *').split(':', 1)[-1].strip() for ln in levels if ln.endswith('*')), None)
        for hw in glob.glob(os.path.join(base, 'hwmon', 'hwmon*')):
            for key, rel in (('power_cap_uW', 'power1_cap'), ('power_cap_max_uW', 'power1_cap_max'), ('power_average_uW', 'power1_average'),
                             ('power_input_uW', 'power1_input'), ('temp_edge_mC', 'temp1_input'), ('temp_hbm_mC', 'temp3_input')):
                try:
                    with open(os.path.join(hw, rel)) as f:
                        sysfs[key] = int(f.read().strip())
                except (OSError, ValueError):
                    pass
    info['sysfs'] = sysfs or None
    return info



def pacing_block(eng):
    """Where the closed loop of the rollout kernel's store pacing stands for the bench launch kind (DESIGN.md section 6;
    gu_rollout.hpp: GuPacer): its period in 10 ns ticks per 16 steps, and the records of its last launches."""
    if not hasattr(eng, 'rollout_pacing'):
        return None
    info = eng.rollout_pacing('uniform', True)
    totals = eng.rollout_pacing_totals() if hasattr(eng, 'rollout_pacing_totals') else None
    if info is None:
        return {'paced': False, 'totals': totals}
    info['paced'] = True
    info['totals'] = totals
    if hasattr(eng, 'rollout_pace_log'):
        lg = eng.rollout_pace_log('uniform', True)
        iv = lg['interval'][lg['interval'] > 0]
        info['last_launches'] = {
            'launches_of_the_kind': int(lg['launches']), 'periods': [round(float(x), 2) for x in lg['period'][-16:]],
            'phase': [int(x) for x in lg['phase'][-16:]],
            'launches_in_log': int(len(lg['seq'])), 'launches_behind_in_log': int((lg['verdict'] == 2).sum()),
            'waves_behind_share_in_log': float(lg['ended_late'].sum()) / max(1, int(lg['waves'].sum())),
            'start_to_start_us_median': float(np.median(iv)) / 100.0 if len(iv) else None}
    return info


def placement_block(eng, post_probe_ms, launch_ms):
    """What gu_reserve_trajectory's candidate search did for the bench buffer, and the same bare store
    probe once more on the kept buffer right after the timed region."""
    if not hasattr(eng, 'trajectory_placement'):
        return None
    n, best, worst = eng.trajectory_placement()
    out = {'candidates_probed': n, 'probe_ms_kept': best, 'probe_ms_slowest': worst}
    if hasattr(eng, 'trajectory_placement_detail'):
        out.update(eng.trajectory_placement_detail())
    out['probe_ms_kept_after_timed_region'] = post_probe_ms
    if post_probe_ms and best:
        out['probe_drift'] = post_probe_ms / best
    if post_probe_ms and launch_ms:
        out['kernel_over_probe_after'] = launch_ms / post_probe_ms
    return out


# --------------------------------------------------------------------------------------- the line
def _num(x, digits=6):
    """Floats at 6 significant digits (the line is for machines; the side file keeps full precision)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float('%.*g' % (digits, x))
    if isinstance(x, dict):
        return {k: _num(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_num(v, digits) for v in x]
    return x


def _pick(d, keys):
    return None if not d else {k: d[k] for k in keys if k in d}


CONTRACT = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
            'dtype', 'data')


def compact_line(detail, detail_path=None):
    """The driver's line: contract keys, config, roofline, cpu_baseline, the parity bits, and every other figure as numbers
    only.  Everything in it is copied from `detail` (nothing is computed here)."""
    line = {k: detail[k] for k in CONTRACT}
    line['config'] = _pick(detail['config'], ('workload', 'envs_per_gpu', 'env_steps_per_launch', 'global_envs', 'parallelism', 'devices'))
    line['roofline'] = _pick(detail['roofline'], ('bound', 'achieved', 'peak', 'unit', 'frac', 'frac_wall', 'traffic', 'traffic_over_algorithmic',
                                                  'traffic_measured_by_child_runs', 'kernel', 'launch_ms', 'kernel_avg_us', 'kernel_avg_us_profile', 'algorithmic_bytes_per_launch',
                                                  'vs_measured_copy_rate'))
    if 'cpu_baseline' in detail:
        cpu = _pick(detail['cpu_baseline'], ('value', 'unit', 'cores', 'kind', 'sample', 'host_cpu_model', 'host_cpu_count', 'host_usable_cores',
                                             'numpy_vectorised_value', 'c_oracle_value'))
        cpu['all_cores'] = _pick(detail['cpu_baseline'].get('all_cores'), ('value', 'cores'))
        line['cpu_baseline'] = cpu
    for key in ('bit_exact_vs_reference_digest', 'bit_exact_vs_oracle'):
        if key in detail:
            line[key] = detail[key]
    if 'final_state_vs_oracle' in detail:
        line['final_state_vs_oracle'] = _pick(detail['final_state_vs_oracle'], ('equal', 'envs_checked', 'env_steps_each', 'launches'))
    line['timing'] = _pick(detail['timing'], ('blocks', 'timed_seconds', 'ms_per_step_min', 'ms_per_step_max', 'launch_ms_min', 'launch_ms_max',
                                              'launches_total'))
    if detail.get('mode'):
        line['mode'] = detail['mode']
    line['engine'] = detail['engine']
    line['per_rank'] = _pick(detail.get('per_rank'), ('value',))
    if detail.get('rccl') is not None:
        line['rccl'] = _pick(detail['rccl'], ('nranks', 'comm_init_ms', 'allgather_ms', 'view_envs', 'view_equals_shards', 'error'))
        if line['rccl'].get('error'):
            line['rccl']['error'] = str(line['rccl']['error'])[:240]
    c4 = detail.get('strong_c4')
    if c4 is not None:
        line['strong_c4'] = _pick(c4, ('value', 'scaling', 'total_envs', 'envs_per_gpu', 'n_gpus', 'ms_per_step', 'launch_ms', 'hbm_gbps_per_gpu',
                                       'shards_equal_oracle', 'bit_exact_vs_reference_digest', 'skipped'))
    others = detail.get('other_modes')
    if others:
        line['other_modes'] = {name: _pick(o, ('value', 'ms_per_launch', 'frac_of_hbm_peak', 'returns_vs_oracle')) for name, o in others.items()}
    configs = detail.get('configs')
    if configs:
        line['configs'] = {name: _pick(c, ('us_per_launch', 'us_per_round', 'env_steps_per_s', 'frac_of_hbm_peak', 'asymptote_frac_of_hbm_peak',
                                           'fixed_us_per_launch', 'form', 'bit_exact')) for name, c in configs.items()}
    if detail_path:
        line['detail'] = detail_path
    return _num(line)


def emit_report(detail, emit, detail_path):
    """Writes the side file (best effort) and hands the ONE line to `emit`."""
    written = None
    if detail_path:
        try:
            with open(detail_path, 'w') as f:
                json.dump(detail, f, indent=1)
                f.write('\n')
            written = detail_path
        except OSError as err:
            sys.stderr.write('bench.py: could not write %s: %s\n' % (detail_path, err))
    line = compact_line(detail, written)
    text = json.dumps(line, separators=(',', ':'))
    for optional in ('timing', 'per_rank', 'other_modes', 'configs', 'strong_c4'):  # (never needed so far: a line always goes out)
        if len(text) < LINE_LIMIT:
            break
        line.pop(optional, None)
        text = json.dumps(line, separators=(',', ':'))
    if len(text) >= LINE_LIMIT:
        raise RuntimeError('bench.py: the JSON line is %d bytes; the driver needs it under %d' % (len(text), LINE_LIMIT))
    ctypes.CDLL(None).fflush(None)  # whatever native code still holds in its stdout buffer comes BEFORE the line, never after
    emit(text)
    sys.stdout.flush()
