cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_rows_kernel.py tests/test_gpu_traj_layout.py tests/test_gpu_store_pacing.py tests/test_gpu_parity.py tests/test_gpu_kstep_kernel.py tests/test_gpu_mc.py tests/test_gpu_options.py tests/test_gpu_c_abi.py -q -m gpu -x > gpurun_out/r05x_pytest.txt 2>&1; grep -E "passed|failed|Error" gpurun_out/r05x_pytest.txt | tail -5
for rep in 1 2; do
echo "== build of this morning (old prologue, old loops)"; GU_ALLOW_STALE_LIB=1 GU_LIB_PATH=$PWD/griduniverse_amd/lib/libgu_prev.so python tools/rows_timing.py --settle 250 --reps 3
echo "== this build"; python tools/rows_timing.py --settle 250 --reps 3
done
python tools/layout_ab.py --half --only c2:uniform; python tools/layout_ab.py --half --only c4:uniform
python bench.py > gpurun_out/r05x_bench.json 2> gpurun_out/r05x_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05x_bench.json').read().strip().split('\n')[-1])
print('value %.4g ms_per_step %.5f frac %.3f frac_wall %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_wall']))
print('configs', {k: (v.get('us_per_launch') or v.get('us_per_round')) for k, v in (d.get('configs') or {}).items()})
print('other', {k: v.get('ms_per_launch') for k, v in (d.get('other_modes') or {}).items()})
PY
