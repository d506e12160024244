"""ctypes binding of libgu.so (include/gu.h, and the introspection calls of include/gu_diag.h) -- the only door to the HIP kernels.

There is deliberately NO fallback: if the library is missing, or no MI355X is
visible, every compute entry point raises `GuError`.  Host-only logic (argument
validation, level loading, maze generation, ASCII rendering) never touches this
module, so it keeps working on a machine without a GPU.
"""
import ctypes
import os
import re
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GU_LIB_PATH selects another build of the same sources (the A/B tools use lib/libgu_exp.so, `make -C csrc exp`)
LIB_PATH = os.environ.get('GU_LIB_PATH') or os.path.join(_HERE, 'lib', 'libgu.so')
CSRC = os.path.join(_HERE, 'csrc')

GU_OK = 0
ERR_NAMES = {-1: 'GU_ERR_INVALID', -2: 'GU_ERR_HIP', -3: 'GU_ERR_NOMEM', -4: 'GU_ERR_STATE',
             -5: 'GU_ERR_COMM', -6: 'GU_ERR_UNSUPPORTED'}

F_AUTO_RESET, F_TRAJECTORY, F_STATS, F_PINNED_IO, F_PACKED = 1, 2, 4, 8, 16
POLICY_UNIFORM, POLICY_STREAM, POLICY_GREEDY, POLICY_SAMPLE = 0, 1, 2, 3
COMM_ID_BYTES = 128
OPT_UNSET = -2 ** 63
# gu_set_option / gu_get_option (include/gu.h "options"): name -> id
OPTIONS = {'rollout_block': 1, 'rollout_rows': 2, 'rows_copies': 3, 'rollout_multi': 4, 'rollout_multi_k': 5,
           'rollout_multi_copies': 6, 'rollout_xcd': 7, 'vi_path': 8, 'mc_scratch_mb': 9, 'mc_lane_returns': 10,
           'mc_global_walk': 11, 'step_sync': 12, 'traj_candidates': 13, 'traj_far_candidates': 14, 'traj_stride_mib': 15,
           'traj_far_mib': 16, 'traj_probe_all': 17, 'rollout_pace': 18, 'vi_xcd_block': 19, 'pace_target': 20, 'pace_bar_num': 21,
           'pace_gain_q': 22, 'pace_dec_q': 23, 'traj_layout': 24, 'pace_record': 25, 'pace_probe_every': 26, 'pace_adapt': 27, 'rollout_half_waves': 28, 'rollout_entry': 29,
           'sync_spin_us': 30,
           # experiments: refused by libgu.so, accepted by libgu_exp.so only
           'x_traj_uncached': 100, 'x_traj_poison': 101, 'x_mc_poison': 102}

_c = ctypes
_vp, _i32, _i64, _u32, _u64, _f64 = _c.c_void_p, _c.c_int32, _c.c_int64, _c.c_uint32, _c.c_uint64, _c.c_double

# name -> argtypes (restype is int everywhere).  Must list every symbol of include/gu.h and include/gu_diag.h;
# tests/test_abi.py cross-checks this table against the header and the built library.
SIGNATURES = {
    'gu_version': [],
    'gu_last_error': [_c.c_char_p, _c.c_size_t],
    'gu_device_count': [_c.POINTER(_c.c_int)],
    'gu_source_hash': [_c.c_char_p, _c.c_size_t],
    'gu_device_info': [_c.c_int, _c.c_char_p, _c.c_size_t],
    'gu_set_option': [_vp, _i32, _i64],
    'gu_get_option': [_vp, _i32, _c.POINTER(_i64)],
    'gu_create': [_c.c_int, _i64, _i64, _c.POINTER(_vp)],
    'gu_destroy': [_vp],
    'gu_set_grid': [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32],
    'gu_set_grids': [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32],
    'gu_generate_mazes': [_vp, _i32, _i32, _i32, _u64],
    'gu_get_cells': [_vp, _i32, _vp, _vp, _vp, _vp],
    'gu_seed': [_vp, _u64],
    'gu_reset': [_vp, _vp, _vp, _vp],
    'gu_reset_done': [_vp],
    'gu_step': [_vp, _vp, _u32, _vp, _vp, _vp],
    'gu_upload_actions': [_vp, _vp, _i64],
    'gu_step_device': [_vp, _i64, _u32],
    'gu_step_graph': [_vp, _i64, _i64, _u32],
    'gu_read_outputs': [_vp, _vp, _vp, _vp],
    'gu_reserve_trajectory': [_vp, _i64],
    'gu_trajectory_placement': [_vp, _vp, _vp, _vp],
    'gu_trajectory_placement_detail': [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    'gu_probe_trajectory': [_vp, _c.POINTER(_c.c_float)],
    'gu_rollout': [_vp, _i64, _i32, _u32],
    'gu_rollout_pacing': [_vp, _i32, _u32, _vp, _vp, _vp, _vp, _vp],
    'gu_rollout_calibrate': [_vp, _i64, _i32, _u32],
    'gu_rollout_pacing_totals': [_vp, _vp, _vp, _vp, _vp, _vp],
    'gu_rollout_pace_log': [_vp, _i32, _u32, _i32, _vp, _vp, _vp],
    'gu_rollout_pace_waves': [_vp, _i32, _u32, _i32, _vp, _vp],
    'gu_read_trajectory': [_vp, _i64, _i64, _vp, _vp, _vp],
    'gu_read_trajectory_packed': [_vp, _i64, _i64, _vp],
    'gu_read_stats': [_vp, _vp, _vp],
    'gu_get_state': [_vp, _vp, _vp, _vp, _vp],
    'gu_set_state': [_vp, _vp, _vp, _vp, _vp],
    'gu_done_indices': [_vp, _vp, _vp],
    'gu_look_step_ahead': [_vp, _i64, _vp, _vp, _i32, _vp, _vp, _vp],
    'gu_vi_set': [_vp, _vp, _vp],
    'gu_vi_sweep': [_vp, _f64, _i32, _i32, _vp],
    'gu_vi_run': [_vp, _f64, _f64, _i32, _vp, _vp],
    'gu_vi_eval_run': [_vp, _f64, _f64, _i32, _vp, _vp],
    'gu_vi_greedy': [_vp, _f64],
    'gu_vi_get': [_vp, _vp, _vp],
    'gu_vi_sweep_step': [_vp, _f64, _u32, _vp],
    'gu_vi_sweep_step_run': [_vp, _f64, _i32, _u32, _vp],
    'gu_vi_last_form': [_vp],
    'gu_vi_last_dp_form': [_vp],
    'gu_vi_xcd_torn_words': [_vp, _vp],
    'gu_vi_last_clusters': [_vp, _vp],
    'gu_mc_evaluate': [_vp, _i64, _vp, _i32, _i32, _i32, _f64, _vp, _vp, _vp, _vp],
    'gu_mc_walk_lengths': [_vp, _i64, _vp, _i64, _i32, _vp, _i32, _vp, _vp],
    'gu_mc_walk_episodes': [_vp, _i64, _vp, _vp, _vp, _vp, _i32, _i64],
    'gu_shortest_paths': [_vp, _i32, _vp, _vp, _vp],
    'gu_render_rgb': [_vp, _i64, _i64, _i32, _vp],
    'gu_trail_enable': [_vp, _i32],
    'gu_trail_read': [_vp, _i64, _i64, _vp, _vp],
    'gu_render_policy_rgb': [_vp, _i32, _vp],
    'gu_host_alloc': [_c.c_size_t, _c.POINTER(_vp)],
    'gu_host_free': [_vp],
    'gu_sync': [_vp],
    'gu_timer_begin': [_vp],
    'gu_timer_end': [_vp, _c.POINTER(_c.c_float)],
    'gu_timer_mark': [_vp],
    'gu_timer_laps': [_vp, _vp, _i32, _vp],
    'gu_comm_unique_id': [_vp],
    'gu_comm_init': [_vp, _i32, _i32, _vp],
    'gu_comm_destroy': [_vp],
    'gu_allgather_view': [_vp, _vp, _vp, _vp],
    'gu_comm_init_all': [_vp, _i32],
    'gu_allgather_view_all': [_vp, _i32, _vp, _vp, _vp],
}


class GuError(RuntimeError):
    def __init__(self, code, message):
        super().__init__('{} ({}): {}'.format(ERR_NAMES.get(code, 'GU_ERR'), code, message))
        self.code = code


_lib = None


def build(verbose=False):
    """Compile libgu.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ['make', '-j8', '-C', CSRC] + ([] if verbose else ['-s'])
    subprocess.check_call(cmd)
    return LIB_PATH


def source_hash():
    """sha256 (16 hex digits) over the library's sources as they are on disk -- the Makefile's SRCHASH."""
    import hashlib
    names = sorted(n for n in os.listdir(CSRC) if n.endswith(('.hip', '.hpp')))
    h = hashlib.sha256()
    for path in [os.path.join(CSRC, n) for n in names] + [os.path.join(os.path.dirname(_HERE), 'include', n) for n in ('gu.h', 'gu_diag.h')]:
        with open(path, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


_MARKER = re.compile(rb'GU_SRCHASH=([0-9a-f]{16}|unknown);GU_BUILD=(\w+);')


def build_marker(path=None):
    """(source hash, build kind) a libgu*.so carries in its bytes, or None when there is no such file.

    Read from the FILE, never through dlopen: a library mapped once cannot be replaced in the process (glibc answers a
    later dlopen of the same path with the old image), so looking at the hash must not map it -- otherwise
    is_stale() -> build() -> load() in one process would keep seeing the library it set out to replace."""
    path = path or LIB_PATH
    try:
        with open(path, 'rb') as f:
            blob = f.read()
    except OSError:
        return None
    m = _MARKER.search(blob)
    return (m.group(1).decode(), m.group(2).decode()) if m else ('unknown', 'unknown')


def built_hash():
    """The source hash compiled into libgu.so, or None when there is no library."""
    marker = build_marker()
    return None if marker is None else marker[0]


def is_stale():
    """True when libgu.so is missing or was built from sources other than those on disk."""
    return built_hash() != source_hash()


def load():
    """dlopen libgu.so.  Raises (never falls back, never builds) when it is missing or stale."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GuError(-2, 'libgu.so not found at {}; build it with `python -c "import __graft_entry__ as g; '
                              'g.build()"` or `make -C griduniverse_amd/csrc`'.format(LIB_PATH))
        # the hash is read from the file's bytes BEFORE anything is mapped: a stale library that was dlopen'ed once would be
        # handed back by glibc for every later dlopen of the path, and the rebuilt one could never be loaded by this process
        built = built_hash()
        if os.path.isdir(CSRC) and built != source_hash() and not os.environ.get('GU_ALLOW_STALE_LIB'):
            raise GuError(-2, 'libgu.so at {} was built from other sources (built {}, on disk {}): rebuild with '
                              '`make -C griduniverse_amd/csrc`'.format(LIB_PATH, built, source_hash()))
        lib = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        _lib = lib
    return _lib


def last_error():
    buf = ctypes.create_string_buffer(512)
    load().gu_last_error(buf, 512)
    return buf.value.decode('utf-8', 'replace')


def check(rc):
    if rc != GU_OK:
        raise GuError(rc, last_error())


def set_default_option(name, value):
    """Process-wide default of a launch-shape / search option (every engine without a value of its own uses it).
    value None returns it to the built-in default.  Results never depend on options; tests and tools flip them."""
    check(load().gu_set_option(None, OPTIONS[name], OPT_UNSET if value is None else int(value)))


def get_default_option(name):
    """The value in force for engines without one of their own."""
    v = ctypes.c_int64(0)
    check(load().gu_get_option(None, OPTIONS[name], ctypes.byref(v)))
    return v.value


def device_info(device=0):
    """dict of what gu_device_info reports (name, arch, pci, cus, lds_per_cu, clocks, memory sizes)."""
    buf = ctypes.create_string_buffer(1024)
    rc = load().gu_device_info(int(device), buf, 1024)
    if rc < 0:
        check(rc)
    out = {}
    for item in buf.value.decode('utf-8', 'replace').split(';'):
        if '=' in item:
            k, v = item.split('=', 1)
            out[k] = int(v) if v.lstrip('-').isdigit() else v
    return out


def device_count():
    """Number of visible HIP devices; 0 when there is none (does not raise)."""
    n = ctypes.c_int(0)
    rc = load().gu_device_count(ctypes.byref(n))
    return n.value if rc == GU_OK else 0


def ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class PinnedArray(object):
    """numpy array living in page-locked host memory (gu_host_alloc); freed with the object."""

    def __init__(self, shape, dtype=np.int32):
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._ptr = ctypes.c_void_p()
        check(load().gu_host_alloc(max(self.nbytes, 1), ctypes.byref(self._ptr)))
        buf = (ctypes.c_char * max(self.nbytes, 1)).from_address(self._ptr.value)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def free(self):
        if self._ptr is not None and self._ptr.value:
            self.array = None
            load().gu_host_free(self._ptr)
            self._ptr = ctypes.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass


def as_array(a, dtype, shape=None, name='array'):
    out = np.ascontiguousarray(a, dtype=dtype)
    if shape is not None and tuple(out.shape) != tuple(shape):
        raise ValueError('{} must have shape {}, got {}'.format(name, tuple(shape), tuple(out.shape)))
    return out
