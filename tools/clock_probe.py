#!/usr/bin/env python3
"""Launch time of the bench kernel next to the clocks / power rocm-smi reports WHILE it runs (the box-to-box spread of the
unchanged kernel, 117-141 us per launch, is the largest term in every comparison across sessions).  Usage: python tools/clock_probe.py"""
import os
import random
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import griduniverse_amd as gua  # noqa: E402

random.seed(123)
np.random.seed(123)
env = gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True)
eng = gua.Engine(65536, gua.GridSpec.from_env(env), seed=123)
eng.reset()
eng.reserve_trajectory(1000)
t0, smi = time.time(), ''
while time.time() - t0 < 3.0:
    for _ in range(200):
        eng.rollout(1000, 'uniform', True, True)
    if not smi and time.time() - t0 > 1.2:  # sampled while 200 launches are queued
        smi = subprocess.run('rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -i "sclk\\|mclk\\|fclk\\|power (W)\\|junction\\|memory)"',
                             shell=True, stdout=subprocess.PIPE).stdout.decode()
    eng.sync()
eng.timer_begin()
for _ in range(200):
    eng.rollout(1000, 'uniform', True, True)
us = eng.timer_end() * 1e3 / 200
print('bench kernel: %.1f us per launch = %.2f TB/s' % (us, 786.432e6 / us / 1e6))
print(smi)
print(subprocess.run('rocm-smi --showmemorypartition --showcomputepartition --showxgmierr 2>&1 | grep -i "partition\|xgmi" | head -6; rocm-smi --showhw 2>&1 | tail -4; cat /sys/class/drm/card*/device/current_memory_partition 2>/dev/null | head -2; cat /sys/class/drm/card*/device/mem_info_vram_total 2>/dev/null | head -2',
                     shell=True, stdout=subprocess.PIPE).stdout.decode())
