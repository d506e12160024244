cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_dp.py -q -m gpu -x -k "tables_a_dp_call" 2>&1 | tail -5
python tools/api_latency.py > gpurun_out/r05zz_api_latency.txt 2>&1; tail -30 gpurun_out/r05zz_api_latency.txt | cut -c1-250
