cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/layout_ab.py --half --sizes 4096 8192 16384 32768 --json gpurun_out/r05m_half_sizes.json > gpurun_out/r05m_half_sizes.txt 2>&1; cat gpurun_out/r05m_half_sizes.txt | cut -c1-400
timeout 900 python tools/layout_ab.py --half --json gpurun_out/r05m_half_ab.json > gpurun_out/r05m_half_ab.txt 2>&1; cat gpurun_out/r05m_half_ab.txt | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_rows_kernel.py tests/test_gpu_traj_layout.py tests/test_gpu_parity.py tests/test_gpu_store_pacing.py -q -m gpu > gpurun_out/r05m_pytest.txt 2>&1; grep -E " passed| failed|rror" gpurun_out/r05m_pytest.txt | tail -5
