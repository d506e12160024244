cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_dp.py tests/test_gpu_compat_drivers.py tests/test_gpu_facade.py -x -q -m gpu > gpurun_out/r05k_pytest_dp.txt 2>&1; grep -E "passed|failed|rror" gpurun_out/r05k_pytest_dp.txt | tail -5; grep -B5 -A30 "Error" gpurun_out/r05k_pytest_dp.txt | head -60
timeout 300 python tools/api_latency.py > gpurun_out/r05k_api_latency.json 2>&1; grep -i "iteration\|vi_\|sweep" gpurun_out/r05k_api_latency.json | head -20
timeout 300 python tools/dp_forms.py > gpurun_out/r05k_dp_forms.txt 2>&1; tail -30 gpurun_out/r05k_dp_forms.txt | cut -c1-200
GU_LIB_PATH=$GRAFT_REPO_ROOT/griduniverse_amd/lib/libgu_torn.so timeout 1500 python tools/xcd_stress.py 1200 gpurun_out/r05k_xcd_torn.txt 2>&1 | tail -5
