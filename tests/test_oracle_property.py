"""Property tests (hypothesis, CPU): the two independent restatements of the transition -- per-instance Python
(oracle/ref_env.py, pinned to the reference by goldens) and batched C (oracle/gu_oracle.c) -- agree on random
grids, so the C oracle can stand in for the reference at sizes the goldens do not cover."""
import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

from oracle import c_oracle as C
from oracle.ref_env import OracleGridUniverseEnv


@st.composite
def grids(draw):
    W, H = draw(st.integers(1, 12)), draw(st.integers(1, 12))
    S = W * H
    cells = st.lists(st.integers(0, S - 1), max_size=max(1, S // 2), unique=True)
    walls, lava = draw(cells), draw(st.lists(st.integers(0, S - 1), max_size=4, unique=True))
    goals = draw(st.lists(st.integers(0, S - 1), min_size=1, max_size=3, unique=True))
    starts = draw(st.lists(st.integers(0, S - 1), min_size=1, max_size=3, unique=True))
    return dict(W=W, H=H, walls=walls, lava=lava, goals=goals, starts=starts)


@settings(max_examples=60, deadline=None)
@given(grids(), st.integers(0, 2 ** 63 - 1))
def test_c_oracle_equals_python_oracle(g, seed):
    env = OracleGridUniverseEnv(grid_shape=(g['W'], g['H']), initial_state=list(g['starts']), goal_states=list(g['goals']),
                                lava_states=list(g['lava']), walls=list(g['walls']))
    grid = C.Grid.from_env(env)
    S = g['W'] * g['H']
    s, a = np.repeat(np.arange(S), 4), np.tile(np.arange(4), S)
    for care in (True, False):
        nxt, rew, don = C.look_step_ahead(grid, s, a, care)
        for i in range(0, 4 * S, max(1, S // 8)):
            want = env.look_step_ahead(int(s[i]), int(a[i]), care)
            assert (nxt[i], rew[i], bool(don[i])) == (want[0], want[1], want[2])
    state = C.State(3, 7)
    first = C.reset(grid, seed, state)
    out = C.rollout(grid, seed, state, 40, True)
    from oracle import gu_rng
    for j in range(3):
        ep = 0
        env.current_state = g['starts'][gu_rng.start_index(seed, 7 + j, ep, len(g['starts']))]
        assert env.current_state == first[j]
        done = False
        for t in range(40):
            if done:
                ep += 1
                env.current_state = g['starts'][gu_rng.start_index(seed, 7 + j, ep, len(g['starts']))]
            o, r, done, _ = env.step(gu_rng.action(seed, 7 + j, t))
            assert (o, r, done) == (out['obs'][t, j], out['reward'][t, j], bool(out['done'][t, j]))
