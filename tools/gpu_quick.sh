cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_rows_kernel.py tests/test_gpu_traj_layout.py tests/test_gpu_store_pacing.py tests/test_gpu_parity.py tests/test_gpu_kstep_kernel.py tests/test_gpu_mc.py tests/test_gpu_options.py -q -m gpu -x > gpurun_out/r05t_pytest.txt 2>&1; grep -E "passed|failed" gpurun_out/r05t_pytest.txt | tail -3
for rep in 1 2; do
echo "== HEAD~ build (old prologue)"; GU_ALLOW_STALE_LIB=1 GU_LIB_PATH=$PWD/griduniverse_amd/lib/libgu_prev.so python tools/rows_timing.py --settle 250 --reps 3
echo "== this build"; python tools/rows_timing.py --settle 250 --reps 3
done
echo "== HEAD~ build (old prologue)"; GU_ALLOW_STALE_LIB=1 GU_LIB_PATH=$PWD/griduniverse_amd/lib/libgu_prev.so python tools/rows_intercept.py
echo "== this build"; python tools/rows_intercept.py
