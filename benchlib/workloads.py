"""The workloads of BASELINE.json's configs, built by the product's own host code."""
import random

import numpy as np

import griduniverse_amd as gua

METRIC = 'env-steps/sec at N_envs on 32×32 grid, 1/2/4/8 MI355X; bit-exact vs CPU'
BYTES_PER_ENV_STEP = 12       # fused rollout writing the int32 (obs, reward, done) trajectory
HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
HBM_COPY_GBPS = 6290.0
C4_TOTAL_ENVS = 262144        # BASELINE.json config 4
WORKLOAD_SEED = {'c2': 2, 'c3': 123, 'c4': 4, 'c5': 5}
REFERENCE_DIGEST = {'c2': 'c2_open8x8_4096x1000', 'c3': 'c3_maze32_65536x1000', 'c4': 'c4_lava32_262144x250'}


def build_workload(name):
    """Returns (template env, description)."""
    if name == 'c3':
        random.seed(123)
        np.random.seed(123)
        env = gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True)
        return env, '32x32 generator maze (random.seed(123); np.random.seed(123))'
    if name == 'c4':
        env = gua.GridUniverseEnv(grid_shape=(32, 32), lava_states=[16 + 32 * r for r in range(24)])
        return env, '32x32 open grid, start 0, goal 1023, lava column [16+32r, r<24]'
    if name == 'c2':
        return gua.GridUniverseEnv(grid_shape=(8, 8)), 'default 8x8 grid'
    if name == 'c5':
        random.seed(5)
        np.random.seed(5)
        return gua.GridUniverseEnv(grid_shape=(64, 64), random_maze=True), '64x64 generator maze (random.seed(5); np.random.seed(5))'
    raise SystemExit('unknown workload ' + name)


def workload_line(name, N, grid_desc, T):
    """config.workload of the JSON line (kept under 200 characters)."""
    return '%s: %d envs per GPU on the %s, uniform device-RNG actions, auto-reset, %d env-steps per launch, int32 (obs,reward,done) rows' \
        % (name, N, grid_desc, T)
