"""Per-engine options with a process default, gu_device_info, several engines of one process with large LDS tables."""
import numpy as np
import pytest

from griduniverse_amd import Engine, GridSpec, _lib
from oracle import c_oracle as C
from tests import _golden as G
import griduniverse_amd as gua

pytestmark = pytest.mark.gpu

def spec_of(meta):
    return GridSpec(meta['W'], meta['H'], meta['starts'], meta['goals'], meta['lava'], meta['walls'], meta['reward'])


def test_options_are_per_engine_with_a_process_default(gu_option):
    """Two engines of one process on different kernel paths at the same time (the environment switches of rounds 1 / 2 could not do
    that), identical results; an engine's own value wins over the process default, None hands it back."""
    meta, _ = G.load_traj('c3_maze32')
    N, T = 4096, 300
    for name in ('rollout_rows', 'rollout_block', 'rollout_multi'):
        gu_option(name, None)  # (a GU_TEST_OPTIONS session may have set them: this test is about the built-in defaults)
    with Engine(N, spec_of(meta), seed=11) as a, Engine(N, spec_of(meta), seed=11) as b:
        assert a.get_option('rollout_rows') == -1 and a.get_option('rollout_block') == 256
        a.set_option('rollout_rows', 0)
        a.set_option('rollout_multi', 0)      # a: the general kernel for everything
        b.set_option('rollout_rows', 1)       # b: the transition-row kernel wherever eligible
        b.set_option('rollout_block', 512)
        gu_option('rollout_block', 128)       # process default: a uses it, b keeps its own 512
        assert a.get_option('rollout_block') == 128 and b.get_option('rollout_block') == 512
        outs = []
        for e in (a, b):
            e.reset()
            e.reserve_trajectory(T)
            e.rollout(T, 'uniform', True, True, stats=True)
            outs.append((e.read_trajectory(0, T), e.read_stats(), e.get_state()))
        for k in ('obs', 'reward', 'done'):
            assert np.array_equal(outs[0][0][k], outs[1][0][k]), k
        assert all(np.array_equal(x, y) for x, y in zip(outs[0][1], outs[1][1]))
        assert all(np.array_equal(outs[0][2][k], outs[1][2][k]) for k in outs[0][2])
        grid, st = C.Grid.from_lists(**meta), C.State(N)
        C.reset(grid, 11, st)
        want = C.rollout(grid, 11, st, T, True)
        assert all(np.array_equal(outs[0][0][k], want[k]) for k in want)
        b.set_option('rollout_block', None)
        assert b.get_option('rollout_block') == 128
        with pytest.raises(gua.GuError):
            a.set_option('x_traj_uncached', 1)  # compiled out of the product library


def test_device_info_names_the_device():
    info = _lib.device_info(0)
    assert str(info['arch']).startswith('gfx950') and info['cus'] >= 1 and info['lds_per_cu'] >= 65536
    assert info['hbm_bytes'] > 2 ** 34 and 0 < info['hbm_free'] <= info['hbm_bytes'] and len(str(info['pci'])) >= 7
    assert Engine.device_info(0)['name'] == info['name']


def test_several_engines_of_one_process_with_large_lds_tables():
    """parallel.MultiDeviceVecGridUniverse on [0, 0, 0] (the box has one GPU; on a node the same code puts each engine on its own
    device, where the dynamic-LDS limit of a kernel has to be raised once PER DEVICE -- the latch is a per-device mask now):
    statistics-only launches on a 32x32 grid need 80 KiB of LDS for the K = 2 table; all shards against the oracle."""
    from griduniverse_amd.parallel import MultiDeviceVecGridUniverse
    meta, _ = G.load_traj('c3_maze32')
    total, T = 3 * 1024, 500
    env = MultiDeviceVecGridUniverse(total, [0, 0, 0], seed=5, auto_reset=True, template=spec_of(meta))
    try:
        env.reset()
        out = env.rollout(T, trajectory=False, stats=True)
        grid, st = C.Grid.from_lists(**meta), C.State(total)
        C.reset(grid, 5, st)
        want = C.rollout(grid, 5, st, T, True, trajectory=False, stats=True)
        assert np.array_equal(out['ret'], want['ret']) and np.array_equal(out['episodes'], want['episodes'])
        view = env.view()
        assert np.array_equal(view[0], st.pos) and np.array_equal(view[2], st.done)
    finally:
        env.close()
