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
for rep in range(2):
    out = baseline_configs(gua.Engine, 0, 20, rep == 0, only=('c3_distinct',))
    print(json.dumps({k: {kk: v[kk] for kk in ('us_per_launch', 'env_steps_per_s', 'frac_of_hbm_peak', 'bit_exact')} for k, v in out.items()}))
PY
