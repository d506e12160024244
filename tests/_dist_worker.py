"""Worker for tests/test_multiprocess.py: launched by torch.distributed.run with world_size 2, the way the driver launches
bench.py -- but the ranks themselves never import torch: their host channel is griduniverse_amd/rendezvous.py.
Each rank runs its shard of a 4096-env batch on the oracle-backed engine stub, then checks the shard against
the single-process run of the whole batch and the gathered view against the concatenation of all shards."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from griduniverse_amd import rendezvous  # noqa: E402
from griduniverse_amd.parallel import ShardedVecGridUniverse, env_launch_info, shard_range  # noqa: E402
from oracle import c_oracle as C  # noqa: E402
from tests._oracle_engine import OracleEngine  # noqa: E402


def main():
    out_dir = sys.argv[1]
    rank, local_rank, world = env_launch_info()
    channel = rendezvous.default()  # (what ShardedVecGridUniverse uses for RCCL's id by default)
    assert (rank, world) == (channel.rank, channel.world)
    OracleEngine.host_channel = channel
    total, T, seed = 4096, 200, 77
    lava = [16 + 32 * r for r in range(24)]  # config-4 grid
    env = ShardedVecGridUniverse(total, seed=seed, auto_reset=True, engine_factory=OracleEngine,
                                 grid_shape=(32, 32), lava_states=lava)
    assert (env.env_id0, env.num_envs) == shard_range(total, world, rank) == (rank * total // world, total // world)
    env.reset()
    traj = env.rollout(T)

    # the whole batch in one piece (what a single GPU would compute)
    grid = C.Grid.from_lists(32, 32, lava=lava)
    whole = C.State(total)
    C.reset(grid, seed, whole)
    want = C.rollout(grid, seed, whole, T, True)
    lo, hi = env.env_id0, env.env_id0 + env.num_envs
    ok_shard = all(np.array_equal(traj[k], want[k][:, lo:hi]) for k in ('obs', 'reward', 'done'))

    obs, reward, done = env.gathered_view()
    ok_view = (obs.shape == (total,) and np.array_equal(obs, want['obs'][-1]) and np.array_equal(reward, want['reward'][-1])
               and np.array_equal(done, want['done'][-1].astype(bool)) and done.dtype == bool)

    # the bench.py timing reduction: MAX over ranks of a per-rank scalar
    channel.barrier()
    ok_max = channel.reduce([1.0 + rank, -float(rank)], 'MAX') == [float(world), 0.0] and channel.reduce([1.0 + rank], 'MIN') == [1.0]

    with open(os.path.join(out_dir, 'rank%d.json' % rank), 'w') as f:
        json.dump(dict(rank=rank, world=world, ok_shard=bool(ok_shard), ok_view=bool(ok_view), ok_max=bool(ok_max),
                       ids=[int(env.global_ids()[0]), int(env.global_ids()[-1])]), f)
    env.close()
    channel.barrier()
    channel.close()
    rendezvous._default = None
    assert 'torch' not in sys.modules, 'the product\'s N > 1 path imported torch'

    # bench.py's multi-rank flow (blocks, max-over-ranks, gathered-view check, strong-scaling config 4) on the stub engine
    import bench
    args = bench.parse_args(['--gpus', str(world), '--envs', '512', '--T', '40', '--steps', '2', '--warmup', '1',
                             '--min-seconds', '0.02', '--c4-envs', '2048', '--detail', os.path.join(out_dir, 'bench_detail.json')])
    lines = []
    bench.run(args, engine_cls=OracleEngine, emit=lines.append)
    assert len(lines) == (1 if rank == 0 else 0)
    if rank == 0:
        with open(os.path.join(out_dir, 'bench_line.json'), 'w') as f:
            f.write(lines[0])
    assert 'torch' not in sys.modules, 'bench.py imported torch'


if __name__ == '__main__':
    main()
