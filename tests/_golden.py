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


_LEVEL_DIR = None


def level_path(name):
    """Path of a level text file materialised from the parsed level in levels.json (the fixtures hold the levels
    as data -- W, H and the index lists the reference's loader produced -- not as copies of its text files)."""
    global _LEVEL_DIR
    if _LEVEL_DIR is None:
        import atexit
        import shutil
        import tempfile
        _LEVEL_DIR = tempfile.mkdtemp(prefix='gu_levels_')
        atexit.register(shutil.rmtree, _LEVEL_DIR, True)
        for fn, sp in load_json('levels.json').items():
            glyph = {}
            for ch, key in (('#', 'walls'), ('L', 'lava'), ('G', 'goals'), ('x', 'starts')):
                for s in sp[key]:
                    glyph[s] = ch
            with open(os.path.join(_LEVEL_DIR, fn), 'w') as f:
                for y in range(sp['H']):
                    f.write(''.join(glyph.get(y * sp['W'] + x, 'o') for x in range(sp['W'])) + '\n')
    return os.path.join(_LEVEL_DIR, name)


def mc_names():
    return sorted(os.path.basename(p)[3:-4] for p in glob.glob(os.path.join(GOLDEN, 'mc_*.npz')))


def load_mc(name):
    return load_npz('mc', name)
