# scratch session for gpurun (edited per experiment)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do
for lib in prev this; do
if [ $lib = prev ]; then export GU_ALLOW_STALE_LIB=1 GU_LIB_PATH=$PWD/griduniverse_amd/lib/libgu_prev.so; else unset GU_ALLOW_STALE_LIB GU_LIB_PATH; fi
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$lib: ms_per_step %.5f launch_ms %.5f  gap per block of 20: %.1f us  frac %.3f frac_wall %.3f' % (d['ms_per_step'], d['roofline']['launch_ms'], (d['ms_per_step']-d['roofline']['launch_ms'])*20e3, d['roofline']['frac'], d['roofline']['frac_wall']))
"
done; done
