"""VecGridUniverse -- N lock-stepped GridUniverse instances on one MI355X.

The batched counterpart of the reference's per-instance loop

    obs = env.reset()
    for t in range(T):
        obs, reward, done, info = env.step(action)      # core/envs/griduniverse_env.py:176-185
        if done: env.reset()                            # harness policy, e.g. monte_carlo.py:25

with the same grid construction kwargs as `GridUniverseEnv` (one shared grid), array
in / array out.  `rollout(T)` fuses the whole caller loop into one kernel launch.
Sharding across GPUs: give every rank its slice via `env_id0`; per-env RNG streams are
keyed by GLOBAL env id, so the union of the shards equals the single-device batch byte
for byte (see `parallel.py`).
"""

from .engine import Engine
from .envs.griduniverse_env import GridUniverseEnv
from .grid import GridSpec


class VecGridUniverse(object):
    def __init__(self, num_envs, grid_shape=(4, 4), *, initial_state=0, goal_states=None, lava_states=None,
                 walls=None, custom_world_fp=None, random_maze=False, template=None, templates=None,
                 device_mazes=None, maze_seed=0, seed=0, device=0, env_id0=0, auto_reset=False, engine_factory=Engine):
        """`template`: an existing GridUniverseEnv (or any object with the reference env's grid
        attributes) to take the grid from; otherwise the grid kwargs are validated and built
        exactly like GridUniverseEnv's (same exceptions, same RNG consumption).
        `templates=[...]`: several distinct grids of one shape, env e using grid e // (num_envs // len);
        `device_mazes=G, maze_seed=k`: G random mazes of `grid_shape` generated on the GPU."""
        self.num_envs = int(num_envs)
        if templates is not None or device_mazes is not None:
            # several distinct grids of one shape; env e uses grid e // (num_envs // n_grids)
            self.env_id0, self.auto_reset, self.info = int(env_id0), bool(auto_reset), {}
            if templates is not None:
                specs = [t if isinstance(t, GridSpec) else GridSpec.from_env(t) for t in templates]
                self.template, self.spec = templates[0], specs[0]
                self.engine = engine_factory(self.num_envs, specs[0], device=device, env_id0=env_id0, seed=seed)
                self.engine.set_grids(specs)
            else:
                # `device_mazes` random mazes carved on the GPU (csrc/gu_maze.hip) instead of one host maze
                W, H = grid_shape
                self.template = None
                self.spec = GridSpec(W, H, [0], [W * H - 1], [], [])
                self.engine = engine_factory(self.num_envs, self.spec, device=device, env_id0=env_id0, seed=seed)
                self.engine.generate_mazes(int(device_mazes), W, H, maze_seed)
            return
        if template is None:
            template = GridUniverseEnv(grid_shape, initial_state=initial_state, goal_states=goal_states,
                                       lava_states=lava_states, walls=walls, custom_world_fp=custom_world_fp,
                                       random_maze=random_maze, device=device)
        self.template = template
        self.spec = template if isinstance(template, GridSpec) else GridSpec.from_env(template)
        self.env_id0 = int(env_id0)
        self.auto_reset = bool(auto_reset)
        self.engine = engine_factory(self.num_envs, self.spec, device=device, env_id0=env_id0, seed=seed)
        self.info = {}

    # gym-like surface -----------------------------------------------------------------
    def seed(self, seed):
        self.engine.seed(seed)
        return [seed]

    def reset(self, mask=None, start_choice=None):
        """Reset all (or the masked) envs; start cell of multi-start levels from RNG stream 1
        unless `start_choice` gives an index into starting_states per env.  Returns obs int32[N]."""
        return self.engine.reset(mask, start_choice)

    def step(self, actions, zero_copy=False):
        """actions int32[N] in 0..3 -> (obs int32[N], reward int32[N], done bool[N], info).

        zero_copy=True: `actions` is copied into the engine's page-locked buffer (or pass None after filling
        `self.actions_buffer` in place) and the kernel writes the results straight into page-locked host memory;
        the returned obs / reward arrays are VIEWS that the next zero-copy step overwrites."""
        if zero_copy:
            if actions is not None:
                self.engine.pinned_actions[:] = actions
            obs, reward, done = self.engine.step_pinned(self.auto_reset)
            return obs, reward, done.astype(bool), self.info
        obs, reward, done = self.engine.step(actions, self.auto_reset)
        return obs, reward, done.astype(bool), self.info

    @property
    def actions_buffer(self):
        """int32[N] page-locked array read directly by the step kernel in zero-copy mode."""
        return self.engine.pinned_actions

    def rollout(self, T, policy='uniform', actions=None, auto_reset=None, trajectory=True, stats=False):
        """T fused steps.  policy: 'uniform' (device RNG), 'stream' (give `actions` int32[T,N]) or
        'greedy' (argmax of the policy table set through `engine.vi_set`).  Returns a dict with
        obs/reward/done int32[T,N] when `trajectory` (True, or 'packed' to move 4 instead of 12 bytes per
        env-step through HBM and PCIe), plus ret/episodes when `stats`."""
        auto = self.auto_reset if auto_reset is None else bool(auto_reset)
        if actions is not None:
            policy = 'stream'
            self.engine.upload_actions(actions)
        if trajectory:
            self.engine.reserve_trajectory(T)
        self.engine.rollout(T, policy, auto, trajectory, stats)
        if trajectory == 'packed':  # one uint32 per env-step on the device, unpacked to the same three arrays here
            out = self.engine.read_trajectory_packed(0, T)
        else:
            out = self.engine.read_trajectory(0, T) if trajectory else {}
        if stats:
            out['ret'], out['episodes'] = self.engine.read_stats()
        return out

    def done_indices(self):
        return self.engine.done_indices()

    def get_state(self):
        return self.engine.get_state()

    def set_state(self, **kw):
        self.engine.set_state(**kw)

    def close(self):
        self.engine.close()

    @property
    def observations(self):
        return self.engine.read_outputs()[0]
