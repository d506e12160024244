# round 6: MAP 5 (one grid per env at four bits per cell) -- parity tests, then the bench line with the distinct-grid configs
cd $GRAFT_REPO_ROOT
TAG=${1:-r06d}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_step_api.py tests/test_gpu_parity.py -m gpu -q > gpurun_out/${TAG}_pytest.log 2>&1; tail -8 gpurun_out/${TAG}_pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --detail gpurun_out/${TAG}_bench_detail.json > gpurun_out/${TAG}_bench_stdout.txt 2> gpurun_out/${TAG}_bench.err
echo "bench rc $? bytes $(wc -c < gpurun_out/${TAG}_bench_stdout.txt)"; cat gpurun_out/${TAG}_bench_stdout.txt; tail -3 gpurun_out/${TAG}_bench.err
