cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_dp.py tests/test_gpu_compat_drivers.py tests/test_gpu_facade.py tests/test_gpu_step_api.py -q -m gpu > gpurun_out/r05l_pytest_dp.txt 2>&1; grep -E " passed| failed|rror" gpurun_out/r05l_pytest_dp.txt | tail -5
timeout 600 python tools/c5_forms.py > gpurun_out/r05l_c5_forms.txt 2>&1; tail -25 gpurun_out/r05l_c5_forms.txt | cut -c1-300
GU_LIB_PATH=$GRAFT_REPO_ROOT/griduniverse_amd/lib/libgu_torn.so timeout 400 python tools/xcd_stress.py 300 gpurun_out/r05l_xcd_torn.txt 2>&1 | tail -3
timeout 400 python tools/xcd_stress.py 240 gpurun_out/r05l_xcd_stress.txt 2>&1 | tail -3
