"""Pins oracle/bfs.py (restatement of the breadth-first path search of the reference's demo script
core/algorithms/maze_solving.py:43-50, 113-193) to tests/golden/bfs.json: paths, terminal states and graph sizes
produced by the reference's OWN function definitions, lifted out of the script with `ast` and executed against the
real reference env by tests/golden/make_golden.py (capture_bfs)."""
import pytest

from oracle import bfs
from oracle.ref_env import OracleGridUniverseEnv
from tests import _golden as G

CASES = G.load_json('bfs.json')


def env_of(case):
    env = OracleGridUniverseEnv(grid_shape=(case['W'], case['H']), initial_state=list(case['starts']),
                                goal_states=list(case['goals']), lava_states=list(case['lava']), walls=list(case['walls']))
    assert [int(r) for r in env.reward_matrix] == case['reward']
    return env


def test_fixture_covers_the_edge_cases():
    by = {c['name']: c for c in CASES}
    assert by['start_is_terminal']['path'] == []
    assert by['start_walled_in']['path'] is None and by['goal_behind_walls']['path'] is None
    assert by['start_on_wall']['error'] == 'KeyError'
    assert by['lava_nearer_than_goal']['terminal'] in by['lava_nearer_than_goal']['lava']
    # W = 1: the reference names vertical moves LEFT / RIGHT (difference of 1 is tested first, :116-121)
    assert by['column1x9']['path'] == [1] * 8
    assert len(by['maze_101x101_level']['path']) == 240 and by['maze_101x101_level']['terminal'] == 10098
    assert len(CASES) >= 20


@pytest.mark.parametrize('case', CASES, ids=lambda c: c['name'])
def test_restated_search_equals_the_reference_search(case):
    env = env_of(case)
    graph = bfs.create_graph(env)
    assert len(graph) == case['graph_nodes'] and sum(len(v) for v in graph.values()) == case['graph_edges']
    path, terminal = bfs.breadth_first_search(env, case['start'])
    # the one documented difference: a wall start is KeyError in the reference, (None, None) in the restatement
    assert path == case['path'] and terminal == case['terminal']
    if path:  # walking the path on the env ends on the terminal, and not before
        env.current_state = case['start']
        dones = [env.step(a) for a in path]
        if case['W'] > 1:  # (the W = 1 quirk yields actions that do not move the agent)
            assert dones[-1][0] == terminal and dones[-1][2] and not any(d[2] for d in dones[:-1])
