# round 5, session i: layout tests, the reference-RNG Monte-Carlo walk on the device, the whole suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_traj_layout.py tests/test_gpu_mc.py tests/test_gpu_compat_drivers.py -x -q -m gpu > gpurun_out/r05i_pytest_a.txt 2>&1; grep -E "passed|failed|rror" gpurun_out/r05i_pytest_a.txt | tail -8; grep -B5 -A25 "Error\|assert" gpurun_out/r05i_pytest_a.txt | head -80
timeout 300 python tools/mc_numpy_latency.py > gpurun_out/r05i_mc_numpy_latency.txt 2>&1; cat gpurun_out/r05i_mc_numpy_latency.txt
timeout 1800 python -m pytest tests -q -m gpu > gpurun_out/r05i_pytest.txt 2>&1; grep -E "passed|failed|rror" gpurun_out/r05i_pytest.txt | tail -8
