"""Shortest path to a terminal state, importable counterpart of the reference's demo script
`core/algorithms/maze_solving.py` (everything there sits under `if __name__ == '__main__':`, :12).

Same construction: the graph has an edge s -> s' for every action whose `look_step_ahead(s, a,
care_about_terminal=False)` moves the agent (:43-50, the only user of that flag), breadth-first search from
`env.initial_state` with a FIFO queue, children visited in action order, stopping when a terminal state (goal OR
lava, `env.is_terminal`) is dequeued (:123-169); the result is the list of action indices along the tree path
(:171-193).  The transition table comes from the HIP kernel gu_look_step_ahead (one launch for all S x 4 pairs);
the search itself is host Python -- it runs once per grid and is not a batch workload.
"""
from collections import deque


def create_graph(env):
    """{state: [next states that differ from it, in action order]} for every non-wall state (:43-50)."""
    nxt = env._transition_table(False)[0]
    graph = {}
    for s in range(env.world.size):
        if not env._is_wall(s):
            graph[s] = [int(n) for n in nxt[s] if n != s]
    return graph


def calculate_action(parent_state, next_state, x_max=None):
    """Action index that leads from parent_state to the adjacent next_state (:113-127 returns the names)."""
    diff = next_state - parent_state
    if diff == 1:
        return 1  # RIGHT
    if diff == -1:
        return 3  # LEFT
    return 2 if diff > 1 else 0  # DOWN / UP


def breadth_first_search(env, start_state=None):
    """List of actions from `start_state` (default env.initial_state) to the first terminal state the FIFO search
    dequeues, or None when no terminal state is reachable."""
    graph = create_graph(env)
    start = env.initial_state if start_state is None else start_state
    parent = {start: None}
    queue = deque([start])
    while queue:
        state = queue.popleft()
        if env.is_terminal(state):
            actions = []
            while parent[state] is not None:
                actions.append(calculate_action(parent[state], state))
                state = parent[state]
            return actions[::-1]
        for child in graph.get(state, []):
            if child not in parent:  # == "not in closed_set and not in open_set" of :151-160
                parent[child] = state
                queue.append(child)
    return None
