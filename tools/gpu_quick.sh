cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/dp_call_breakdown.py > gpurun_out/r05n_dp_call_breakdown.txt 2>&1; cat gpurun_out/r05n_dp_call_breakdown.txt
timeout 900 python -m pytest tests/test_gpu_dp.py tests/test_gpu_compat_drivers.py tests/test_gpu_facade.py tests/test_gpu_mc.py -q -m gpu > gpurun_out/r05n_pytest.txt 2>&1; grep -E " passed| failed|rror" gpurun_out/r05n_pytest.txt | tail -5
