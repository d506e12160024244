"""Batch sharding over several MI355X: one process per GPU, no data-path collective.

The reference is single-process; env instances never read each other
(core/envs/griduniverse_env.py:136-193), so a batch of `total_envs` splits into contiguous
env-index blocks, rank g owning envs [g*n, (g+1)*n).  The grid is replicated, per-env RNG
streams are keyed by GLOBAL env id, hence the union of the shards is byte-identical to the
single-device batch and stepping needs no communication at all.

The one exchange is optional: `gathered_view()` returns the single-array (obs, reward, done)
of ALL envs on every rank -- one RCCL all-gather of each rank's packed int32[3n] block over
xGMI (csrc/gu_comm.hip).  RCCL's 128-byte unique id travels from rank 0 to the others over
a host channel: by default the package's own socket rendezvous (rendezvous.py: torchrun-style
RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT, no PyTorch anywhere in the product); any
callable `broadcast_bytes(payload, src)` will do -- `torch_broadcast_bytes` is one for callers
that already run a `torch.distributed` process group.
"""
import os

import numpy as np

from .vec_env import VecGridUniverse


def shard_range(total_envs, world_size, rank):
    """(first global env id, number of envs) of `rank`.  Equal blocks; `total_envs` must divide."""
    total_envs, world_size, rank = int(total_envs), int(world_size), int(rank)
    if world_size <= 0 or not 0 <= rank < world_size:
        raise ValueError('bad rank {} of {}'.format(rank, world_size))
    if total_envs <= 0 or total_envs % world_size:
        raise ValueError('total_envs={} must be a positive multiple of world_size={} (the gathered view '
                         'all-gathers equal blocks)'.format(total_envs, world_size))
    per = total_envs // world_size
    return rank * per, per


def env_launch_info():
    """(rank, local_rank, world_size) from the torchrun-style environment."""
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')),
            int(os.environ.get('WORLD_SIZE', '1')))


def rendezvous_broadcast_bytes(payload, src=0):
    """Broadcast a byte string from `src` over the package's socket rendezvous (created from the environment on first use)."""
    from . import rendezvous
    return rendezvous.broadcast_bytes(payload, src)


def torch_broadcast_bytes(payload, src=0):
    """The same over an initialised torch.distributed process group (optional; the product never imports torch itself)."""
    import torch
    import torch.distributed as dist
    buf = torch.zeros(len(payload), dtype=torch.uint8)
    if dist.get_rank() == src:
        buf = torch.frombuffer(bytearray(payload), dtype=torch.uint8).clone()
    dist.broadcast(buf, src=src)
    return bytes(buf.numpy().tobytes())


def unpack_view(blocks, n):
    """[world][3n] packed (obs|reward|done) blocks -> three env-major arrays of world*n."""
    blocks = np.asarray(blocks, dtype=np.int32).reshape(-1, 3, n)
    return tuple(np.ascontiguousarray(blocks[:, k, :]).reshape(-1) for k in range(3))


class ShardedVecGridUniverse(object):
    """This rank's shard of a `total_envs` batch, plus the gathered view.

    Grid kwargs are those of `GridUniverseEnv` / `VecGridUniverse`.  NOTE for `random_maze=True`:
    every rank must build the SAME grid, so seed both global RNGs identically on all ranks before
    constructing (random.seed(k); np.random.seed(k)), exactly as for the reference.
    """

    def __init__(self, total_envs, *, rank=None, world_size=None, device=None, seed=0, auto_reset=False,
                 broadcast_bytes=rendezvous_broadcast_bytes, engine_factory=None, **grid_kwargs):
        env_rank, env_local, env_world = env_launch_info()
        self.rank = env_rank if rank is None else int(rank)
        self.world_size = env_world if world_size is None else int(world_size)
        self.total_envs = int(total_envs)
        self.env_id0, self.num_envs = shard_range(total_envs, self.world_size, self.rank)
        kw = dict(grid_kwargs)
        if engine_factory is not None:
            kw['engine_factory'] = engine_factory
        self.local = VecGridUniverse(self.num_envs, seed=seed, device=env_local if device is None else device,
                                     env_id0=self.env_id0, auto_reset=auto_reset, **kw)
        self._broadcast = broadcast_bytes
        self._comm_ready = False

    # local shard: no communication ------------------------------------------------------
    def reset(self, mask=None, start_choice=None):
        return self.local.reset(mask, start_choice)

    def step(self, actions):
        return self.local.step(actions)

    def rollout(self, T, **kw):
        return self.local.rollout(T, **kw)

    def global_ids(self):
        return np.arange(self.env_id0, self.env_id0 + self.num_envs, dtype=np.int64)

    # the single-array view: one all-gather -----------------------------------------------
    def _ensure_comm(self):
        if self._comm_ready:
            return
        eng = self.local.engine
        uid = eng.comm_unique_id() if self.rank == 0 else bytes(128)
        if self.world_size > 1:
            uid = self._broadcast(uid, 0)
        eng.comm_init(self.world_size, self.rank, uid)
        self._comm_ready = True

    def gathered_view(self):
        """(obs, reward, done) of all `total_envs` envs, env-major, identical on every rank."""
        self._ensure_comm()
        obs, reward, done = self.local.engine.allgather_view()
        return obs, reward, done.astype(bool)

    def close(self):
        if self._comm_ready:
            self.local.engine.comm_destroy()
            self._comm_ready = False
        self.local.close()


class MultiDeviceVecGridUniverse(object):
    """One host process driving several GPUs: one engine (handle + HIP stream) per device, contiguous env-index
    shards, launches issued asynchronously device after device so the GPUs run concurrently.
    No collective at all: the single-array view is assembled on the host from one D2H copy per device.

    `devices` may repeat an index (e.g. [0, 0] on a one-GPU box): results do not depend on the placement."""

    def __init__(self, total_envs, devices, *, seed=0, auto_reset=False, engine_factory=None, **grid_kwargs):
        self.devices = list(devices)
        self.total_envs = int(total_envs)
        kw = dict(grid_kwargs)
        if engine_factory is not None:
            kw['engine_factory'] = engine_factory
        first = VecGridUniverse(shard_range(total_envs, len(self.devices), 0)[1], seed=seed, device=self.devices[0],
                                env_id0=0, auto_reset=auto_reset, **kw)
        self.shards = [first]
        for g, dev in enumerate(self.devices[1:], start=1):
            id0, n = shard_range(total_envs, len(self.devices), g)
            # every shard shares the first shard's grid (built once: random mazes must not be re-drawn per device)
            share = {k: v for k, v in kw.items() if k == 'engine_factory'}
            if first.template is not None:
                self.shards.append(VecGridUniverse(n, template=first.template, seed=seed, device=dev, env_id0=id0,
                                                   auto_reset=auto_reset, **share))
            else:
                self.shards.append(VecGridUniverse(n, seed=seed, device=dev, env_id0=id0, auto_reset=auto_reset, **kw))
        self.auto_reset = bool(auto_reset)

    def reset(self):
        return np.concatenate([s.reset() for s in self.shards])

    def step(self, actions):
        actions = np.ascontiguousarray(actions, dtype=np.int32)
        outs, lo = [], 0
        for s in self.shards:
            outs.append(s.step(actions[lo:lo + s.num_envs]))
            lo += s.num_envs
        return (np.concatenate([o[0] for o in outs]), np.concatenate([o[1] for o in outs]),
                np.concatenate([o[2] for o in outs]), {})

    def rollout(self, T, policy='uniform', trajectory=True, stats=False):
        """Enqueue the fused rollout on every device first, then collect: the devices overlap."""
        for s in self.shards:
            if trajectory:
                s.engine.reserve_trajectory(T)
            s.engine.rollout(T, policy, self.auto_reset, trajectory, stats)
        out = {}
        if trajectory:
            parts = [s.engine.read_trajectory(0, T) for s in self.shards]
            out = {k: np.concatenate([p[k] for p in parts], axis=1) for k in parts[0]}
        if stats:
            parts = [s.engine.read_stats() for s in self.shards]
            out['ret'] = np.concatenate([p[0] for p in parts])
            out['episodes'] = np.concatenate([p[1] for p in parts])
        return out

    def view(self, rccl=False):
        """(obs, reward, done) of all envs, env-major.  Default: one D2H copy per device, concatenated on the host.
        rccl=True: one grouped RCCL all-gather over xGMI (needs every shard on its own device), then one copy."""
        if rccl:
            engines = [s.engine for s in self.shards]
            cls = type(engines[0])
            if not getattr(self, '_comm_all', False):
                cls.comm_init_all(engines)
                self._comm_all = True
            return tuple(cls.allgather_view_all(engines))
        parts = [s.engine.read_outputs() for s in self.shards]
        return tuple(np.concatenate([p[k] for p in parts]) for k in range(3))

    def close(self):
        for s in self.shards:
            s.close()
