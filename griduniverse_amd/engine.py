"""Engine: numpy-facing wrapper of one libgu handle (one device, one HIP stream).

Thin by design -- every method is one C-ABI call (include/gu.h) plus array
marshalling; all grid / argument semantics of the reference live in
`griduniverse_amd.envs` (host Python) and in the kernels (device).
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import check, ptr
from .grid import GridSpec

_POLICIES = {'uniform': _lib.POLICY_UNIFORM, 'stream': _lib.POLICY_STREAM, 'greedy': _lib.POLICY_GREEDY,
             'sample': _lib.POLICY_SAMPLE}


class Engine(object):
    def __init__(self, num_envs, spec, device=0, env_id0=0, seed=0):
        if not isinstance(spec, GridSpec):
            raise TypeError('spec must be a GridSpec')
        self._h = ctypes.c_void_p()
        self.lib = _lib.load()
        self.N = int(num_envs)
        self.env_id0 = int(env_id0)
        self.device = int(device)
        check(self.lib.gu_create(self.device, self.N, self.env_id0, ctypes.byref(self._h)))
        self.spec = None
        self._pinned = {}
        self._pinned_io = None
        try:
            self.set_grid(spec)
            self.seed(seed)
        except Exception:
            self.close()
            raise

    # ------------------------------------------------------------------ lifetime
    def close(self):
        for p in getattr(self, '_pinned', {}).values():
            p.free()
        self._pinned = {}
        self._pinned_io = None
        if getattr(self, '_h', None) is not None and self._h.value:
            self.lib.gu_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @staticmethod
    def device_info(device=0):
        """dict: name, arch, pci, cus, lds_per_cu, sclk_khz, mclk_khz, bus_bits, l2_bytes, hbm_bytes, hbm_free (gu_device_info)."""
        return _lib.device_info(device)

    # ------------------------------------------------------------------ options
    def set_option(self, name, value):
        """Launch-shape / search option of THIS engine (include/gu.h "options"; `_lib.OPTIONS` names them); None returns
        it to the process default.  Results never depend on options."""
        check(self.lib.gu_set_option(self._h, _lib.OPTIONS[name], _lib.OPT_UNSET if value is None else int(value)))

    def get_option(self, name):
        v = ctypes.c_int64(0)
        check(self.lib.gu_get_option(self._h, _lib.OPTIONS[name], ctypes.byref(v)))
        return v.value

    # ------------------------------------------------------------------ configuration
    def set_grid(self, spec):
        p = spec.planes()
        starts = np.asarray(spec.starts, dtype=np.int32)
        check(self.lib.gu_set_grid(self._h, spec.W, spec.H, spec.words_per_row, ptr(p['wall']), ptr(p['goal']),
                                   ptr(p['lava']), ptr(p['rplus']), ptr(p['rminus']), ptr(starts), len(starts)))
        self.spec = spec
        self._n_grids = 1

    def set_grids(self, specs):
        """Several distinct grids of one shape: env e uses specs[e // (N // len(specs))]."""
        G = len(specs)
        W, H = specs[0].W, specs[0].H
        if any((sp.W, sp.H) != (W, H) for sp in specs):
            raise ValueError('all grids of one engine must have the same shape')
        planes = [sp.planes() for sp in specs]
        stack = {k: np.ascontiguousarray(np.stack([p[k] for p in planes])) for k in planes[0]}
        max_starts = max(len(sp.starts) for sp in specs)
        starts = np.zeros((G, max_starts), np.int32)
        n_starts = np.zeros(G, np.int32)
        for g, sp in enumerate(specs):
            starts[g, :len(sp.starts)] = sp.starts
            n_starts[g] = len(sp.starts)
        check(self.lib.gu_set_grids(self._h, G, W, H, specs[0].words_per_row, ptr(stack['wall']), ptr(stack['goal']),
                                    ptr(stack['lava']), ptr(stack['rplus']), ptr(stack['rminus']), ptr(starts),
                                    ptr(n_starts), max_starts))
        self.spec = specs[0]
        self.specs = list(specs)
        self._n_grids = len(specs)

    def generate_mazes(self, n_grids, W, H, maze_seed):
        """n_grids random mazes carved on the device (one per env group of N // n_grids envs)."""
        check(self.lib.gu_generate_mazes(self._h, int(n_grids), int(W), int(H), int(maze_seed) & 0xFFFFFFFFFFFFFFFF))
        self.spec = GridSpec(W, H, [0], [W * H - 1], [], [])  # shape holder; the real grids live on the device
        self.specs = None
        self._n_grids = int(n_grids)

    def get_cells(self, grid_index=0):
        """(flags uint8[S], reward int8[S], starts int32[n]) of one grid as compiled on the device."""
        S = self.spec.S
        flags, reward = np.empty(S, np.uint8), np.empty(S, np.int8)
        n = ctypes.c_int32(0)
        check(self.lib.gu_get_cells(self._h, int(grid_index), None, None, None, ctypes.byref(n)))
        starts = np.empty(max(n.value, 1), np.int32)  # a start list may repeat cells and be longer than the grid
        check(self.lib.gu_get_cells(self._h, int(grid_index), ptr(flags), ptr(reward), ptr(starts), ctypes.byref(n)))
        return flags, reward, starts[:n.value].copy()

    def seed(self, seed):
        self.seed_value = int(seed) & 0xFFFFFFFFFFFFFFFF
        check(self.lib.gu_seed(self._h, self.seed_value))

    # ------------------------------------------------------------------ reset / step
    def reset(self, mask=None, start_choice=None):
        m = None if mask is None else _lib.as_array(np.asarray(mask).astype(bool), np.uint8, (self.N,), 'mask')
        c = None if start_choice is None else _lib.as_array(start_choice, np.int32, (self.N,), 'start_choice')
        obs = np.empty(self.N, np.int32)
        check(self.lib.gu_reset(self._h, ptr(m), ptr(c), ptr(obs)))
        return obs

    def reset_done(self):
        check(self.lib.gu_reset_done(self._h))

    def step(self, actions, auto_reset=False):
        a = _lib.as_array(actions, np.int32, (self.N,), 'actions')
        obs, rew, don = (np.empty(self.N, np.int32) for _ in range(3))
        check(self.lib.gu_step(self._h, ptr(a), _lib.F_AUTO_RESET if auto_reset else 0, ptr(obs), ptr(rew), ptr(don)))
        return obs, rew, don

    def _pin(self, key, shape, dtype=np.int32):
        p = self._pinned.get(key)
        if p is None or p.array.shape != tuple(shape):
            self._pinned_io = None
            if p is not None:
                p.free()
            p = self._pinned[key] = _lib.PinnedArray(tuple(shape), dtype)
        return p.array

    @property
    def pinned_actions(self):
        """int32[N] page-locked action buffer: fill it in place, then call step_pinned()."""
        return self._pin('act', (self.N,))

    def step_pinned(self, auto_reset=False):
        """gu_step on the engine's page-locked buffers (no bounce copies).  Returns views that stay
        valid -- and are overwritten -- until the next step_pinned()."""
        io = self._pinned_io
        if io is None:  # views and their C pointers are built once: per call they would cost more than the launch
            a, out = self.pinned_actions, self._pin('out', (3, self.N))
            io = self._pinned_io = ((ptr(a), ptr(out[0]), ptr(out[1]), ptr(out[2])), (out[0], out[1], out[2]))
        flags = (_lib.F_AUTO_RESET if auto_reset else 0) | _lib.F_PINNED_IO
        p = io[0]
        rc = self.lib.gu_step(self._h, p[0], flags, p[1], p[2], p[3])
        if rc:
            check(rc)
        return io[1]

    def upload_actions(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.int32)
        if a.ndim != 2 or a.shape[1] != self.N:
            raise ValueError('actions must have shape (T, {}), got {}'.format(self.N, a.shape))
        check(self.lib.gu_upload_actions(self._h, ptr(a), a.shape[0]))

    def step_device(self, t, auto_reset=False):
        check(self.lib.gu_step_device(self._h, int(t), _lib.F_AUTO_RESET if auto_reset else 0))

    def step_graph(self, t0, T, auto_reset=False):
        check(self.lib.gu_step_graph(self._h, int(t0), int(T), _lib.F_AUTO_RESET if auto_reset else 0))

    def read_outputs(self):
        obs, rew, don = (np.empty(self.N, np.int32) for _ in range(3))
        check(self.lib.gu_read_outputs(self._h, ptr(obs), ptr(rew), ptr(don)))
        return obs, rew, don

    # ------------------------------------------------------------------ rollout
    def reserve_trajectory(self, T):
        check(self.lib.gu_reserve_trajectory(self._h, int(T)))

    def trajectory_placement(self):
        """(candidate allocations tried, probe ms of the kept one, probe ms of the slowest) of the trajectory buffer."""
        n, best, worst = ctypes.c_int32(0), ctypes.c_float(0.0), ctypes.c_float(0.0)
        check(self.lib.gu_trajectory_placement(self._h, ctypes.byref(n), ctypes.byref(best), ctypes.byref(worst)))
        return n.value, best.value, worst.value

    def trajectory_placement_detail(self):
        """Everything the placement search of the trajectory buffer tried: dict(probe_ms=[...], address=[...], kept=index,
        search_ms=wall time of the search, peak_bytes=most device memory it held at once)."""
        n, kept = ctypes.c_int32(0), ctypes.c_int32(-1)
        search, peak = ctypes.c_float(0.0), ctypes.c_uint64(0)
        check(self.lib.gu_trajectory_placement_detail(self._h, 0, None, None, ctypes.byref(n), ctypes.byref(kept),
                                                      ctypes.byref(search), ctypes.byref(peak)))
        ms, addr = np.zeros(max(n.value, 1), np.float32), np.zeros(max(n.value, 1), np.uint64)
        check(self.lib.gu_trajectory_placement_detail(self._h, ms.size, ptr(ms), ptr(addr), ctypes.byref(n), None, None, None))
        return dict(probe_ms=[float(x) for x in ms[:n.value]], address=['0x%x' % int(a) for a in addr[:n.value]],
                    kept=kept.value, search_ms=float(search.value), peak_bytes=int(peak.value))

    def probe_trajectory(self):
        """ms of one full write of the trajectory buffer the engine holds, in the rollout's store shape (overwrites it)."""
        ms = ctypes.c_float(0.0)
        check(self.lib.gu_probe_trajectory(self._h, ctypes.byref(ms)))
        return ms.value

    def rollout(self, T, policy='uniform', auto_reset=True, trajectory=True, stats=False):
        """trajectory: False / True (three int32 rows per step) / 'packed' (one uint32 per env-step)."""
        tflag = _lib.F_PACKED if trajectory == 'packed' else (_lib.F_TRAJECTORY if trajectory else 0)
        flags = (_lib.F_AUTO_RESET if auto_reset else 0) | tflag | (_lib.F_STATS if stats else 0)
        check(self.lib.gu_rollout(self._h, int(T), _POLICIES[policy], flags))

    def calibrate_rollout(self, T, policy='uniform', auto_reset=True, trajectory=True, stats=False):
        """= rollout(...).  Rounds 3 and 4 searched the store-pacing period here; the launches choose it themselves now
        (include/gu.h: gu_rollout_calibrate), so nothing is left to ask for."""
        flags = (_lib.F_AUTO_RESET if auto_reset else 0) | (_lib.F_STATS if stats else 0)
        flags |= _lib.F_PACKED if trajectory == 'packed' else (_lib.F_TRAJECTORY if trajectory else 0)
        check(self.lib.gu_rollout_calibrate(self._h, int(T), _POLICIES[policy], flags))

    def rollout_pacing_totals(self):
        """Over all launch kinds of this engine: dict(calibration_ms, launches_spent, kinds_paced, kinds_from_cache, kinds_waiting) --
        calibration_ms and launches_spent are 0: no launch is ever spent on a search."""
        ms = ctypes.c_float(0.0)
        n = [ctypes.c_int32(0) for _ in range(4)]
        check(self.lib.gu_rollout_pacing_totals(self._h, ctypes.byref(ms), *[ctypes.byref(x) for x in n]))
        return dict(calibration_ms=ms.value, launches_spent=n[0].value, kinds_paced=n[1].value, kinds_from_cache=n[2].value, kinds_waiting=n[3].value)

    def rollout_pacing(self, policy='uniform', auto_reset=True, packed=False):
        """The schedule of this launch kind on the current trajectory buffer (include/gu.h: gu_rollout_pacing): dict(period = 10 ns
        ticks per 16 steps of its last launch, ms_paced = that launch on the device's clock, evaluated = launches of the kind so
        far; ms_unpaced and calibration_ms are 0), or None when the kind keeps no schedule (not launched yet, or too small)."""
        period, n = ctypes.c_int32(0), ctypes.c_int32(0)
        a, b, c = ctypes.c_float(0.0), ctypes.c_float(0.0), ctypes.c_float(0.0)
        rc = self.lib.gu_rollout_pacing(self._h, _POLICIES[policy], (_lib.F_AUTO_RESET if auto_reset else 0) | (_lib.F_PACKED if packed else 0), ctypes.byref(period),
                                        ctypes.byref(a), ctypes.byref(b), ctypes.byref(n), ctypes.byref(c))
        if rc == -4:
            return None
        check(rc)
        return dict(period=period.value, ms_unpaced=a.value, ms_paced=b.value, evaluated=n.value, calibration_ms=c.value)

    def rollout_pace_log(self, policy='uniform', auto_reset=True, packed=False):
        """The records of the last (at most 61) launches of this kind, oldest first, as a dict of arrays: seq, period (ticks, float;
        0 = the launch ran without the limiter), verdict (of the launch behind it: 0 none yet, 1 on schedule, 2 behind), phase (0
        limiter on, 1 probing without it, 2 limiter off, 3 probing with it), waves, elapsed (ticks from start to report, slowest
        wave), ended_late (waves), max_behind (ticks), interval (ticks to the next launch's start; 0 for the last); plus 'launches'
        (of the kind on the current shape).  None when the kind keeps no schedule."""
        buf = np.zeros((61, 8), dtype=np.uint64)
        n, launches = ctypes.c_int32(0), ctypes.c_uint32(0)
        rc = self.lib.gu_rollout_pace_log(self._h, _POLICIES[policy], (_lib.F_AUTO_RESET if auto_reset else 0) | (_lib.F_PACKED if packed else 0), 61,
                                          buf.ctypes.data, ctypes.byref(n), ctypes.byref(launches))
        if rc == -4:
            return None
        check(rc)
        e = buf[:n.value].astype(np.int64)
        return dict(seq=e[:, 0], period=e[:, 1] / 64.0, verdict=e[:, 2] & 0xFF, phase=(e[:, 2] >> 8) & 0xFF, dec_q=e[:, 2] >> 16, waves=e[:, 3], elapsed=e[:, 4],
                    ended_late=e[:, 5], max_behind=e[:, 6], interval=e[:, 7], launches=launches.value)

    def rollout_pace_waves(self, policy='uniform', auto_reset=True, packed=False):
        """MEASUREMENT AID: what every wave of this kind's last launch reported: int64[waves] ticks (10 ns) from the wave's start to
        its report a few groups before the end of the launch (0 = did not report).  None when the kind keeps no schedule."""
        cap = (self.N + 31) // 32  # (the most a launch can have: the transition-row kernel's half waves)
        buf = np.zeros(cap, dtype=np.uint32)
        n = ctypes.c_int32(0)
        rc = self.lib.gu_rollout_pace_waves(self._h, _POLICIES[policy], (_lib.F_AUTO_RESET if auto_reset else 0) | (_lib.F_PACKED if packed else 0), cap,
                                            buf.ctypes.data, ctypes.byref(n))
        if rc == -4:
            return None
        check(rc)
        return buf[:n.value].astype(np.int64)

    def read_trajectory(self, t0, T, pinned=False):
        """Rows t0..t0+T-1 of the trajectory as obs/reward/done int32[T, N].  pinned=True returns views of
        page-locked buffers owned by the engine (full PCIe rate; overwritten by the next pinned read)."""
        if pinned:
            buf = self._pin('traj', (3, T, self.N))
            obs, rew, don = buf[0], buf[1], buf[2]
        else:
            obs, rew, don = (np.empty((T, self.N), np.int32) for _ in range(3))
        check(self.lib.gu_read_trajectory(self._h, int(t0), int(T), ptr(obs), ptr(rew), ptr(don)))
        return dict(obs=obs, reward=rew, done=don)

    def read_trajectory_packed(self, t0, T, unpack=True, pinned=False):
        """After rollout(trajectory='packed'): uint32[T, N] words obs | (reward & 0xFF) << 16 | done << 24, or --
        unpack=True -- the same dict of int32 arrays read_trajectory returns."""
        words = self._pin('trajp', (T, self.N), np.uint32) if pinned else np.empty((T, self.N), np.uint32)
        check(self.lib.gu_read_trajectory_packed(self._h, int(t0), int(T), ptr(words)))
        if not unpack:
            return words
        return dict(obs=(words & 0xFFFF).astype(np.int32), reward=((words >> 16) & 0xFF).astype(np.int8).astype(np.int32),
                    done=((words >> 24) & 1).astype(np.int32))

    def read_stats(self):
        ret = np.empty(self.N, np.int64)
        eps = np.empty(self.N, np.int32)
        check(self.lib.gu_read_stats(self._h, ptr(ret), ptr(eps)))
        return ret, eps

    # ------------------------------------------------------------------ state
    def get_state(self):
        pos, don = np.empty(self.N, np.int32), np.empty(self.N, np.int32)
        ep, tc = np.empty(self.N, np.uint32), np.empty(self.N, np.uint64)  # (step counts have 64 bits: include/gu.h, state)
        check(self.lib.gu_get_state(self._h, ptr(pos), ptr(don), ptr(ep), ptr(tc)))
        return dict(pos=pos, done=don, episode=ep, tcount=tc)

    def set_state(self, pos=None, done=None, episode=None, tcount=None):
        pos = None if pos is None else _lib.as_array(pos, np.int32, (self.N,), 'pos')
        done = None if done is None else _lib.as_array(done, np.int32, (self.N,), 'done')
        episode = None if episode is None else _lib.as_array(episode, np.uint32, (self.N,), 'episode')
        tcount = None if tcount is None else _lib.as_array(tcount, np.uint64, (self.N,), 'tcount')
        check(self.lib.gu_set_state(self._h, ptr(pos), ptr(done), ptr(episode), ptr(tcount)))

    def done_indices(self):
        idx = np.empty(self.N, np.int32)
        count = ctypes.c_int32(0)
        check(self.lib.gu_done_indices(self._h, ptr(idx), ctypes.byref(count)))
        return idx[:count.value].copy()

    def look_step_ahead(self, states, actions, care_about_terminal=True):
        s = np.ascontiguousarray(states, dtype=np.int32).ravel()
        a = np.ascontiguousarray(actions, dtype=np.int32).ravel()
        if s.size != a.size:
            raise ValueError('states and actions must have the same number of elements')
        nxt, rew, don = (np.empty(s.size, np.int32) for _ in range(3))
        check(self.lib.gu_look_step_ahead(self._h, s.size, ptr(s), ptr(a), 1 if care_about_terminal else 0,
                                          ptr(nxt), ptr(rew), ptr(don)))
        return nxt, rew, don

    # ------------------------------------------------------------------ tabular DP
    def vi_set(self, v, pi):
        S = self.spec.S
        v = _lib.as_array(v, np.float64, (S,), 'value_function')
        pi = _lib.as_array(pi, np.float64, (S, 4), 'policy')
        check(self.lib.gu_vi_set(self._h, ptr(v), ptr(pi)))

    def vi_sweep(self, gamma=1.0, iters=1, greedy_update=True):
        deltas = np.empty(int(iters), np.float64)
        check(self.lib.gu_vi_sweep(self._h, float(gamma), int(iters), 1 if greedy_update else 0, ptr(deltas)))
        return deltas

    def vi_run(self, gamma=1.0, threshold=1e-5, max_steps=1000):
        """value_iteration's loop in one call (stopping rule on the device).  Returns (rounds done, deltas)."""
        done = ctypes.c_int32(0)
        deltas = np.full(max(int(max_steps), 1), np.nan)
        check(self.lib.gu_vi_run(self._h, float(gamma), float(threshold), int(max_steps), ctypes.byref(done), ptr(deltas)))
        return done.value, deltas[:done.value].copy()

    def vi_eval_run(self, gamma=1.0, threshold=1e-5, max_steps=1000):
        """policy_iteration's evaluation loop: V1 sweeps on the fixed policy until delta < threshold (or max_steps).
        Returns (sweeps done, deltas)."""
        done = ctypes.c_int32(0)
        deltas = np.full(max(int(max_steps), 1), np.nan)
        check(self.lib.gu_vi_eval_run(self._h, float(gamma), float(threshold), int(max_steps), ctypes.byref(done), ptr(deltas)))
        return done.value, deltas[:done.value].copy()

    def vi_greedy(self, gamma=1.0):
        check(self.lib.gu_vi_greedy(self._h, float(gamma)))

    def vi_get(self):
        S = self.spec.S
        v, pi = np.empty(S, np.float64), np.empty((S, 4), np.float64)
        check(self.lib.gu_vi_get(self._h, ptr(v), ptr(pi)))
        return v, pi

    def vi_sweep_step(self, gamma=1.0, auto_reset=False, want_delta=True):
        d = ctypes.c_double(0.0)
        check(self.lib.gu_vi_sweep_step(self._h, float(gamma), _lib.F_AUTO_RESET if auto_reset else 0,
                                        ctypes.byref(d) if want_delta else None))
        return d.value if want_delta else None

    def vi_sweep_step_run(self, gamma=1.0, iters=1, auto_reset=False):
        """`iters` rounds of {V1 + V2 sweep; every env steps greedily on the updated policy} -- one launch (synchronised per XCD, or
        chip-wide) when table and batch fit the resident workgroups, see vi_last_form().  Returns the per-round deltas."""
        deltas = np.empty(int(iters), np.float64)
        check(self.lib.gu_vi_sweep_step_run(self._h, float(gamma), int(iters), _lib.F_AUTO_RESET if auto_reset else 0, ptr(deltas)))
        return deltas

    def vi_last_form(self):
        """Which form the last vi_sweep_step_run took: 1 = one launch synchronised per XCD, 2 = one launch with a chip-wide
        barrier per round, 3 = one launch per round (0: none yet)."""
        return int(self.lib.gu_vi_last_form(self._h))

    def vi_last_dp_form(self):
        """Which form finished the last vi_sweep / vi_run / vi_eval_run: 1 = one XCD's workgroups in one launch, 2 = one workgroup,
        3 = chip-wide cluster, 4 = one launch per round (0: none yet)."""
        return int(self.lib.gu_vi_last_dp_form(self._h))

    def vi_xcd_torn_words(self):
        """A -DGU_VI_XCD_TORN build: exchange words of the per-XCD launches found with the right tag and the wrong payload, summed over
        this engine's launches; None on the product library."""
        n = ctypes.c_int64(0)
        rc = self.lib.gu_vi_xcd_torn_words(self._h, ctypes.byref(n))
        return None if rc else int(n.value)

    def vi_last_clusters(self):
        """Workgroups per XCC id in the last per-XCD launch of vi_sweep_step_run, as the hardware reported them (list of 8)."""
        m = np.zeros(8, np.int32)
        check(self.lib.gu_vi_last_clusters(self._h, ptr(m)))
        return m.tolist()

    def mc_walk_lengths(self, u, n_offsets, start_states, cap, cdf):
        """include/gu.h: gu_mc_walk_lengths.  uint16[n_starts, n_offsets]: the length of the reference's episode that begins at uniform
        i of `u` in start cell start_states[c] (0xFFFF: the uniforms ran out before it ended)."""
        u = _lib.as_array(u, np.float64, None, 'u')
        starts = _lib.as_array(start_states, np.int32, None, 'start_states')
        cdf = _lib.as_array(cdf, np.float64, (self.spec.S, 4), 'cdf')  # (the library copies S * 32 bytes from it)
        out = np.empty((starts.size, int(n_offsets)), np.uint16)
        check(self.lib.gu_mc_walk_lengths(self._h, u.size, ptr(u), int(n_offsets), starts.size, ptr(starts), int(cap), ptr(cdf), ptr(out)))
        return out

    def mc_walk_episodes(self, u, cdf, offsets, first_state, cap, T):
        """include/gu.h: gu_mc_walk_episodes: episode e from uniform offsets[e] and cell first_state[e] into rows 0 .. T-1 of the trajectory."""
        u = _lib.as_array(u, np.float64, None, 'u')
        cdf = _lib.as_array(cdf, np.float64, (self.spec.S, 4), 'cdf')
        off = _lib.as_array(offsets, np.int64, (self.N,), 'offsets')
        first = _lib.as_array(first_state, np.int32, (self.N,), 'first_state')
        check(self.lib.gu_mc_walk_episodes(self._h, u.size, ptr(u), ptr(cdf), ptr(off), ptr(first), int(cap), int(T)))

    def mc_evaluate(self, T, first_state, discount_pow, keep, every_visit=False, incremental_mean=True,
                    stationary_env=True, alpha=0.001):
        """Monte-Carlo evaluation over the trajectory rows 0..T-1 of the last rollout (env e = episode e).
        Returns (value_function[S], total_visit_counter[S])."""
        first = _lib.as_array(first_state, np.int32, (self.N,), 'first_state')
        pw = _lib.as_array(discount_pow, np.float64, (T,), 'discount_pow')
        kp = _lib.as_array(np.asarray(keep).astype(bool), np.uint8, (T,), 'keep')
        S = self.spec.S
        value, visits = np.empty(S, np.float64), np.empty(S, np.float64)
        check(self.lib.gu_mc_evaluate(self._h, int(T), ptr(first), 1 if every_visit else 0, 1 if incremental_mean else 0,
                                      1 if stationary_env else 0, float(alpha), ptr(pw), ptr(kp), ptr(value), ptr(visits)))
        return value, visits

    def shortest_paths(self, max_path=None):
        """Breadth-first shortest path from each grid's first start cell to the first terminal state the FIFO search
        dequeues (maze_solving.py semantics).  Returns a list with one int8 action array (or None) per grid, plus the
        terminal states reached."""
        G, S = self._n_grids, self.spec.S
        max_path = int(max_path or S)
        path = np.zeros((G, max_path), np.int8)
        plen, term = np.zeros(G, np.int32), np.zeros(G, np.int32)
        check(self.lib.gu_shortest_paths(self._h, max_path, ptr(path), ptr(plen), ptr(term)))
        if (plen == -2).any():
            raise ValueError('a path is longer than max_path={}'.format(max_path))
        return [None if n < 0 else path[g, :n].copy() for g, n in enumerate(plen)], term

    def render_policy_rgb(self, cell_px=52):
        """uint8[H*cell_px, W*cell_px, 3]: the tiles of grid 0 with the current policy table drawn as arrows."""
        out = np.empty((self.spec.H * cell_px, self.spec.W * cell_px, 3), np.uint8)
        check(self.lib.gu_render_policy_rgb(self._h, int(cell_px), ptr(out)))
        return out

    def render_rgb(self, env0=0, n_envs=1, cell_px=8):
        """uint8[n_envs, H*cell_px, W*cell_px, 3] frames of envs env0 .. env0+n_envs-1 (rendered on the device)."""
        W, H = self.spec.W, self.spec.H
        out = np.empty((int(n_envs), H * cell_px, W * cell_px, 3), np.uint8)
        check(self.lib.gu_render_rgb(self._h, int(env0), int(n_envs), int(cell_px), ptr(out)))
        return out

    def trail_enable(self, capacity=500):
        """Keep the reference viewer's agent trail per env (env:92-93, 182-184, 190: the cell after every step, newest `capacity`
        <= 500, emptied by reset) and blend it into render_rgb frames (rendering.py:287-311); 0 switches it off again."""
        check(self.lib.gu_trail_enable(self._h, int(capacity)))
        self._trail_cap = int(capacity)

    def trail_read(self, env0=0, n_envs=1):
        """The trails of envs env0 .. env0 + n_envs - 1 as lists of cells, oldest first (the reference's last_n_states as states)."""
        cap = getattr(self, '_trail_cap', 0)
        cells, length = np.empty((int(n_envs), max(cap, 1)), np.int32), np.empty(int(n_envs), np.int32)
        check(self.lib.gu_trail_read(self._h, int(env0), int(n_envs), ptr(cells), ptr(length)))
        return [cells[k, :length[k]].tolist() for k in range(int(n_envs))]

    # ------------------------------------------------------------------ stream / timing
    def sync(self):
        check(self.lib.gu_sync(self._h))

    def timer_begin(self):
        check(self.lib.gu_timer_begin(self._h))

    def timer_end(self):
        ms = ctypes.c_float(0.0)
        check(self.lib.gu_timer_end(self._h, ctypes.byref(ms)))
        return ms.value

    def timer_mark(self):
        check(self.lib.gu_timer_mark(self._h))

    def timer_laps(self, max_laps=65536):
        """Milliseconds between consecutive timer_mark() events (waits for the last one)."""
        ms = np.empty(int(max_laps), np.float32)
        n = ctypes.c_int32(0)
        check(self.lib.gu_timer_laps(self._h, ptr(ms), int(max_laps), ctypes.byref(n)))
        return ms[:n.value].astype(np.float64)

    # ------------------------------------------------------------------ RCCL gathered view
    @staticmethod
    def comm_unique_id():
        buf = np.zeros(_lib.COMM_ID_BYTES, np.uint8)
        check(_lib.load().gu_comm_unique_id(ptr(buf)))
        return buf.tobytes()

    def comm_init(self, nranks, rank, unique_id):
        buf = np.frombuffer(bytes(unique_id), dtype=np.uint8).copy()
        if buf.size != _lib.COMM_ID_BYTES:
            raise ValueError('unique id must be {} bytes'.format(_lib.COMM_ID_BYTES))
        check(self.lib.gu_comm_init(self._h, int(nranks), int(rank), ptr(buf)))
        self.nranks, self.rank = int(nranks), int(rank)

    def comm_destroy(self):
        check(self.lib.gu_comm_destroy(self._h))

    # one process, one engine per device (ncclCommInitAll + one grouped all-gather; every engine on its own device)
    @staticmethod
    def comm_init_all(engines):
        handles = (ctypes.c_void_p * len(engines))(*[e._h for e in engines])
        check(_lib.load().gu_comm_init_all(handles, len(engines)))
        for rank, e in enumerate(engines):
            e.nranks, e.rank = len(engines), rank

    @staticmethod
    def allgather_view_all(engines):
        """(obs, reward, done) of all engines' envs, env-major: one grouped RCCL all-gather, read from the first device."""
        handles = (ctypes.c_void_p * len(engines))(*[e._h for e in engines])
        total = sum(e.N for e in engines)
        obs, rew, don = (np.empty(total, np.int32) for _ in range(3))
        check(_lib.load().gu_allgather_view_all(handles, len(engines), ptr(obs), ptr(rew), ptr(don)))
        return obs, rew, don

    def allgather_view(self):
        total = self.nranks * self.N
        obs, rew, don = (np.empty(total, np.int32) for _ in range(3))
        check(self.lib.gu_allgather_view(self._h, ptr(obs), ptr(rew), ptr(don)))
        return obs, rew, don
