"""Several distinct grids in one engine, and mazes generated on the device."""
import numpy as np
import pytest

import griduniverse_amd as gua
from griduniverse_amd.engine import Engine
from griduniverse_amd.grid import GridSpec
from oracle import c_oracle as C
from tests.test_maze_structure import check_maze_structure

pytestmark = pytest.mark.gpu


def oracle_grid(spec):
    return C.Grid(spec.W, spec.H, spec.wall, spec.lava, spec.goal, spec.reward, spec.starts)


def random_specs(rs, n, W, H):
    S, specs = W * H, []
    for _ in range(n):
        pick = lambda k: [int(x) for x in rs.choice(S, size=int(k), replace=False)]  # noqa: E731
        specs.append(GridSpec(W, H, pick(rs.randint(1, 4)), pick(rs.randint(1, 3)), pick(rs.randint(0, 5)), pick(rs.randint(0, S // 4))))
    return specs


@pytest.mark.parametrize('group', [128, 50, 1])
def test_multi_grid_engine_equals_per_grid_oracle(group):
    rs = np.random.RandomState(group)
    n_grids, W, H, T = 6, 13, 9, 150
    specs = random_specs(rs, n_grids, W, H)
    N, seed, id0 = n_grids * group, 17, 3 * n_grids * group
    with Engine(N, specs[0], env_id0=id0, seed=seed) as eng:
        eng.set_grids(specs)
        first = eng.reset()
        eng.reserve_trajectory(T)
        eng.rollout(T, 'uniform', auto_reset=True, trajectory=True, stats=True)
        got = eng.read_trajectory(0, T)
        acts = rs.randint(0, 4, (5, N)).astype(np.int32)
        steps = [eng.step(a, auto_reset=True) for a in acts]
        eng.upload_actions(acts)
        eng.rollout(5, 'stream', auto_reset=False, trajectory=False)
        final = eng.get_state()
        for g, spec in enumerate(specs):
            fl, rw, st = eng.get_cells(g)
            with Engine(1, spec) as single:
                f1, r1, s1 = single.get_cells(0)
            assert np.array_equal(fl, f1) and np.array_equal(rw, r1) and np.array_equal(st, s1)
    for g, spec in enumerate(specs):
        sl = slice(g * group, (g + 1) * group)
        grid, st = oracle_grid(spec), C.State(group, id0 + g * group)
        assert np.array_equal(C.reset(grid, seed, st), first[sl])
        want = C.rollout(grid, seed, st, T, True)
        for k in ('obs', 'reward', 'done'):
            assert np.array_equal(got[k][:, sl], want[k]), (g, k)
        for i in range(5):
            w = C.rollout(grid, seed, st, 1, True, actions=acts[i:i + 1, sl])
            assert np.array_equal(steps[i][0][sl], w['obs'][0]) and np.array_equal(steps[i][2][sl], w['done'][0])
        C.rollout(grid, seed, st, 5, False, actions=acts[:, sl])
        assert np.array_equal(final['pos'][sl], st.pos) and np.array_equal(final['episode'][sl], st.episode)


@pytest.mark.parametrize('W,H', [(32, 32), (5, 3), (40, 12), (1, 7), (64, 64), (90, 80)])  # (64 x 64: one wave per workgroup; 90 x 80: too big for LDS, records from L2)
def test_one_grid_per_env_on_the_four_bit_image(W, H):
    """One distinct grid PER ENV (the N x GridUniverseEnv(random_maze=True) shape, griduniverse_env.py:318-321): the rollout keeps
    every lane's grid in LDS at four bits per cell and tests the candidate cell (gu_rollout.hpp, MAP 5).  Grids with every quirk
    of SURVEY 8(a): starts on terminal cells and on walls, goals that are walls, cells both goal and lava, several starts; uniform
    and caller-supplied actions, with and without auto-reset, int32 / packed rows and statistics."""
    rs = np.random.RandomState(W * 100 + H)
    S, N, T, seed = W * H, 192, 200, 23
    specs = random_specs(rs, N, W, H) if S > 8 else [GridSpec(W, H, [int(rs.randint(S))], [int(rs.randint(S))], [], []) for _ in range(N)]
    for i, sp in enumerate(specs[:64]):  # the quirks, on purpose
        cells = [int(c) for c in rs.choice(S, size=min(S, 4), replace=False)]
        goals, lava, walls = (np.flatnonzero(m).tolist() for m in (sp.goal, sp.lava, sp.wall))
        kind = i % 4
        if kind == 0:
            specs[i] = GridSpec(W, H, [cells[0]], [cells[0]], lava, walls)                                   # start on a goal: absorbing at once
        elif kind == 1:
            specs[i] = GridSpec(W, H, [cells[0]], goals, [], sorted(set(walls) | {cells[0]}))               # start on a wall
        elif kind == 2:
            specs[i] = GridSpec(W, H, list(sp.starts), [cells[-1]], [cells[-1]], sorted(set(walls) | {cells[-1]}))  # goal = lava = wall
    with Engine(N, specs[0], seed=seed) as eng:
        eng.set_grids(specs)
        eng.reserve_trajectory(T)
        acts = rs.randint(0, 4, (T, N)).astype(np.int32)
        eng.upload_actions(acts)
        runs = []
        for auto in (True, False):
            eng.seed(seed)
            first = eng.reset()
            for policy, traj, stats in (('uniform', True, True), ('stream', True, False), ('uniform', 'packed', False), ('uniform', False, True)):
                eng.rollout(T, policy, auto_reset=auto, trajectory=traj, stats=stats)
                rows = eng.read_trajectory_packed(0, T) if traj == 'packed' else eng.read_trajectory(0, T) if traj else None  # (dicts, both)
                runs.append((auto, policy, traj, stats, first, rows, eng.read_stats() if stats else None, eng.get_state()))
    for g, spec in enumerate(specs):
        grid = oracle_grid(spec)
        st = None
        for auto, policy, traj, stats, first, rows, st_out, state in runs:
            if policy == 'uniform' and traj is True and stats:  # (the first run of an `auto` setting: from seed + reset)
                st = C.State(1, g)
                assert C.reset(grid, seed, st)[0] == first[g], g
            want = C.rollout(grid, seed, st, T, auto, actions=acts[:, g:g + 1] if policy == 'stream' else None, stats=stats)
            if traj:
                for k in ('obs', 'reward', 'done'):
                    assert np.array_equal(rows[k][:, g], want[k][:, 0]), (g, auto, policy, k)
            if stats:
                assert st_out[0][g] == want['ret'][0] and st_out[1][g] == want['episodes'][0], (g, auto, policy)
            assert state['pos'][g] == st.pos[0] and state['done'][g] == st.done[0] and state['episode'][g] == st.episode[0], (g, auto, policy)


def test_one_grid_per_env_across_2_pow_32_steps():
    """The four-bit image (MAP 5) in the launch during which the step counts pass 2^32 (every step asks for the RNG prefix of its
    own epoch, gu_rollout.hpp: prefix_at) and in launches that lie wholly in epoch 1."""
    rs = np.random.RandomState(77)
    W, H, N, T, seed = 9, 7, 128, 120, 31
    specs = random_specs(rs, N, W, H)
    with Engine(N, specs[0], seed=seed) as eng:
        eng.set_grids(specs)
        first = eng.reset()
        tc = (2 ** 32 - 50 + rs.randint(0, 30, N)).astype(np.uint64)
        eng.set_state(tcount=tc)
        eng.reserve_trajectory(T)
        runs = []
        for _ in range(3):
            eng.rollout(T, 'uniform', auto_reset=True, trajectory=True)
            runs.append(eng.read_trajectory(0, T))
        final = eng.get_state()
    for g, spec in enumerate(specs):
        grid, st = oracle_grid(spec), C.State(1, g)
        assert C.reset(grid, seed, st)[0] == first[g]
        st.tcount[:] = tc[g]
        for got in runs:
            want = C.rollout(grid, seed, st, T, True)
            assert all(np.array_equal(got[k][:, g], want[k][:, 0]) for k in ('obs', 'reward', 'done')), g
        assert final['pos'][g] == st.pos[0] and final['tcount'][g] == st.tcount[0] == tc[g] + 3 * T


def test_multi_grid_restrictions():
    specs = random_specs(np.random.RandomState(0), 3, 8, 8)
    with Engine(30, specs[0]) as eng:
        eng.set_grids(specs)
        with pytest.raises(gua.GuError):
            eng.vi_set(np.zeros(64), np.ones((64, 4)) / 4)
        with pytest.raises(gua.GuError):
            eng.set_grids(specs[:2] + specs[:2])  # 4 does not divide 30
        with pytest.raises(ValueError):
            eng.set_grids([specs[0], GridSpec(9, 8, [0], [1], [], [])])
        eng.set_grids(specs[:1])  # back to a single grid: LDS path and DP tables work again
        eng.vi_set(np.zeros(64), np.ones((64, 4)) / 4)


@pytest.mark.parametrize('W,H,n_grids,group', [(32, 32, 64, 64), (11, 7, 40, 3), (64, 64, 8, 256), (4, 1, 5, 2)])
def test_device_generated_mazes_equal_oracle(W, H, n_grids, group):
    N, seed, maze_seed, T = n_grids * group, 5, 2026, 120
    id0 = 2 * N  # as if this were the third shard: grid ids continue globally
    with Engine(N, GridSpec(W, H, [0], [W * H - 1], [], []), env_id0=id0, seed=seed) as eng:
        eng.generate_mazes(n_grids, W, H, maze_seed)
        first = eng.reset()
        eng.reserve_trajectory(T)
        eng.rollout(T, 'uniform', auto_reset=True)
        got = eng.read_trajectory(0, T)
        cells = [eng.get_cells(g) for g in range(n_grids)]
    for g in range(n_grids):
        wall, start, goal = C.generate_maze(maze_seed, id0 // group + g, W, H)
        check_maze_structure(wall.reshape(H, W), start, goal)
        spec = GridSpec(W, H, [start], [goal], [], np.flatnonzero(wall).tolist())
        with Engine(1, spec) as single:
            f1, r1, s1 = single.get_cells(0)
        assert np.array_equal(cells[g][0], f1) and np.array_equal(cells[g][1], r1) and np.array_equal(cells[g][2], s1), g
        if g % 7 == 0:
            sl = slice(g * group, (g + 1) * group)
            grid, st = oracle_grid(spec), C.State(group, id0 + g * group)
            assert np.array_equal(C.reset(grid, seed, st), first[sl])
            want = C.rollout(grid, seed, st, T, True)
            for k in ('obs', 'reward', 'done'):
                assert np.array_equal(got[k][:, sl], want[k]), (g, k)


def test_device_mazes_through_vec_env_and_errors():
    env = gua.VecGridUniverse(4096, grid_shape=(32, 32), device_mazes=64, maze_seed=9, seed=1, auto_reset=True)
    obs = env.reset()
    assert len(set(obs.tolist())) > 20  # distinct mazes -> distinct start cells
    out = env.rollout(50, stats=True)
    assert out['obs'].shape == (50, 4096)
    env.close()
    with Engine(64, GridSpec(3, 3, [0], [8], [], [])) as eng:
        with pytest.raises(gua.GuError):
            eng.generate_mazes(4, 3, 3, 1)   # no room for a corridor
        with pytest.raises(gua.GuError):
            eng.generate_mazes(5, 8, 8, 1)   # 5 does not divide 64


def _oracle_env(spec):
    from oracle.ref_env import OracleGridUniverseEnv
    S = spec.S
    return OracleGridUniverseEnv(grid_shape=(spec.W, spec.H), initial_state=list(spec.starts),
                                 goal_states=np.flatnonzero(spec.goal).tolist(), lava_states=np.flatnonzero(spec.lava).tolist(),
                                 walls=np.flatnonzero(spec.wall).tolist())


def test_batched_shortest_paths_equal_the_restated_reference_search():
    """gu_shortest_paths (one lane per grid) against oracle/bfs.py on open grids with lava / walls / several goals
    (where FIFO tie-breaking decides the path) and on device-generated mazes (unique paths)."""
    from oracle import bfs
    rs = np.random.RandomState(8)
    specs = random_specs(rs, 12, 10, 7)
    specs[3] = GridSpec(10, 7, [0], [69], [], [1, 10, 11])   # start walled in: no terminal reachable
    specs[5] = GridSpec(10, 7, [33], [33], [], [])           # start is terminal: empty path
    with Engine(12 * 4, specs[0]) as eng:
        eng.set_grids(specs)
        paths, terms = eng.shortest_paths()
    for g, spec in enumerate(specs):
        want, term = bfs.breadth_first_search(_oracle_env(spec), spec.starts[0])
        if want is None:
            assert paths[g] is None and terms[g] == -1, g
        else:
            assert paths[g].tolist() == want and terms[g] == term, (g, paths[g].tolist(), want)
    W = H = 21
    with Engine(64, GridSpec(W, H, [0], [W * H - 1], [], []), seed=1) as eng:
        eng.generate_mazes(64, W, H, 3)
        paths, terms = eng.shortest_paths()
        # following every maze's path from its start reaches its goal: feed the paths as an action stream
        T = max(len(p) for p in paths)
        acts = np.zeros((T, 64), np.int32)
        for g, p in enumerate(paths):
            acts[:len(p), g] = p
        first = eng.reset()
        eng.upload_actions(acts)
        eng.reserve_trajectory(T)
        eng.rollout(T, 'stream', auto_reset=False)
        traj = eng.read_trajectory(0, T)
        for g, p in enumerate(paths):
            wall, start, goal = C.generate_maze(3, g, W, H)
            assert first[g] == start and terms[g] == goal
            assert traj['obs'][len(p) - 1, g] == goal and traj['done'][len(p) - 1, g] == 1
            assert not traj['done'][:len(p) - 1, g].any()
            if g < 6:
                spec = GridSpec(W, H, [start], [goal], [], np.flatnonzero(wall).tolist())
                assert p.tolist() == bfs.breadth_first_search(_oracle_env(spec), start)[0]
        with pytest.raises(ValueError):
            eng.shortest_paths(max_path=3)


@pytest.mark.parametrize('case', __import__('tests._golden', fromlist=['x']).load_json('bfs.json'), ids=lambda c: c['name'])
def test_shortest_paths_equal_the_reference_search(case):
    """gu_shortest_paths and the host counterpart algorithms.maze_solving against tests/golden/bfs.json -- paths produced
    by the reference's own create_graph / breadth_first_search / calculate_action / construct_path
    (core/algorithms/maze_solving.py:43-50, 113-193; lifted out of the demo script by tests/golden/make_golden.py)."""
    from griduniverse_amd.algorithms import maze_solving
    spec = GridSpec(case['W'], case['H'], [case['start']], case['goals'], case['lava'], case['walls'], case['reward'])
    with Engine(4, spec) as eng:
        paths, terms = eng.shortest_paths()
    if case['path'] is None:  # unreachable; or the reference's KeyError on a wall start (documented difference)
        assert paths[0] is None and terms[0] == -1
    else:
        assert paths[0].tolist() == case['path'] and terms[0] == case['terminal']
    env = gua.GridUniverseEnv(grid_shape=(case['W'], case['H']), initial_state=list(case['starts']), goal_states=list(case['goals']),
                              lava_states=list(case['lava']), walls=list(case['walls']))
    graph = maze_solving.create_graph(env)
    assert len(graph) == case['graph_nodes'] and sum(len(v) for v in graph.values()) == case['graph_edges']
    assert maze_solving.breadth_first_search(env, case['start']) == case['path']
    env.close()
