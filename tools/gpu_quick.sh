cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/pace_aim.py; python tools/pace_aim.py
timeout 900 python -m pytest tests/test_gpu_store_pacing.py -q -m gpu -x 2>&1 | tail -3
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
ll=d['roofline']['store_pacing']['last_launches']
print('bench: ms_per_step %.5f launch_ms %.5f frac %.3f frac_wall %.3f periods %.1f..%.1f behind share %.3f s2s %.1f' % (d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'], d['roofline']['frac_wall'], min(ll['periods']), max(ll['periods']), ll['waves_behind_share_in_log'], ll['start_to_start_us_median']))
"; done
