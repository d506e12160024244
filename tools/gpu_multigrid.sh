# one maze per env (rollout MAP 5): parity tests + launch times of the distinct-grid configs
cd $GRAFT_REPO_ROOT
TAG=${1:-r06f}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_multigrid.py -m gpu -q 2>&1 | tail -3
python - <<'PY' 2>&1 | tee gpurun_out/${TAG}_distinct_grids.txt
import json, sys
sys.path.insert(0, '.')
import griduniverse_amd as gua
from benchlib.configs import baseline_configs
from griduniverse_amd import _lib
for rep in range(3):
    for layout in (None, 1, 0):
        _lib.set_default_option('traj_layout', layout)
        out = baseline_configs(gua.Engine, 0, 20, rep == 0, only=('c3_distinct',))
        print('traj_layout', layout, json.dumps({k: {kk: round(v[kk], 4) if isinstance(v[kk], float) else v[kk] for kk in ('us_per_launch', 'frac_of_hbm_peak', 'bit_exact')} for k, v in out.items()}))
PY
