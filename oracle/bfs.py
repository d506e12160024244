"""CPU restatement of the reference's breadth-first path search (test infrastructure).

Follows the demo script core/algorithms/maze_solving.py (all of it sits under `if __name__ == '__main__':`, so it
cannot be imported; restated operation for operation): create_graph :43-50, breadth_first_search :123-169,
calculate_action :113-127, construct_path :171-193.  `env` is any object with the reference env's surface.
"""


def create_graph(env):
    graph = {}
    for state in range(env.world.size):
        if not env._is_wall(state):
            graph[state] = []
            for action in range(4):
                nxt, _, _ = env.look_step_ahead(state, action, False)
                if nxt != state:
                    graph[state].append(nxt)
    return graph


def calculate_action(parent_state, next_state):
    diff = parent_state - next_state
    if diff == 1:
        return 3   # 'LEFT'
    if diff == -1:
        return 1   # 'RIGHT'
    if diff < -1:
        return 2   # 'DOWN'
    if diff > 1:
        return 0   # 'UP'
    raise ValueError('not adjacent')


def breadth_first_search(env, start_state):
    """Returns (action list, terminal state) or (None, None) if the queue empties.  One deliberate difference: a
    start cell that is a wall is not a node of the graph (:45), and the reference then dies with KeyError at
    `graph[parent_state]` (:140); here that case returns (None, None)."""
    graph = create_graph(env)
    open_set = [start_state]
    closed_set = set()
    meta = {start_state: (None, None)}
    while open_set:
        parent_state = open_set.pop(0)
        if env.is_terminal(parent_state):
            actions, state = [], parent_state
            while meta[state][0] is not None:
                actions.append(meta[state][1])
                state = meta[state][0]
            return actions[::-1], parent_state
        for child in graph.get(parent_state, []):
            if child in closed_set:
                continue
            if child not in open_set:
                meta[child] = (parent_state, calculate_action(parent_state, child))
                open_set.append(child)
        closed_set.add(parent_state)
    return None, None
