"""Loaders for the committed golden fixtures (captured from the real reference by
tests/golden/make_golden.py; data only)."""
import glob
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def traj_names():
    return sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLDEN, 'traj_*.npz')))


def dp_names():
    return sorted(os.path.basename(p)[3:-4] for p in glob.glob(os.path.join(GOLDEN, 'dp_*.npz')))


def load_npz(prefix, name):
    z = np.load(os.path.join(GOLDEN, '%s_%s.npz' % (prefix, name)))
    meta = json.loads(str(z['meta']))
    return meta, {k: z[k] for k in z.files if k != 'meta'}


def load_traj(name):
    return load_npz('traj', name)


def load_dp(name):
    return load_npz('dp', name)


def level_path(name):
    return os.path.join(GOLDEN, 'levels', name)


def mc_names():
    return sorted(os.path.basename(p)[3:-4] for p in glob.glob(os.path.join(GOLDEN, 'mc_*.npz')))


def load_mc(name):
    return load_npz('mc', name)
