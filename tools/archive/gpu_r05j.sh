# round 5, session j: MC numpy latency after the host-side trims; the torn-half detector under the stress tool
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_mc.py tests/test_gpu_compat_drivers.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/mc_numpy_latency.py > gpurun_out/r05j_mc_numpy_latency.txt 2>&1; cat gpurun_out/r05j_mc_numpy_latency.txt
GU_LIB_PATH=$GRAFT_REPO_ROOT/griduniverse_amd/lib/libgu_torn.so timeout 1500 python tools/xcd_stress.py 1200 gpurun_out/r05j_xcd_torn.txt 2>&1 | tail -5
