# round 6, first session: the compact bench line as the driver runs it, the bench / mc GPU tests
cd $GRAFT_REPO_ROOT
TAG=r06a
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/${TAG}_bench_detail.json > gpurun_out/${TAG}_bench_stdout.txt 2> gpurun_out/${TAG}_bench.err
echo "bench rc $? stdout bytes $(wc -c < gpurun_out/${TAG}_bench_stdout.txt) lines $(wc -l < gpurun_out/${TAG}_bench_stdout.txt)"
cat gpurun_out/${TAG}_bench_stdout.txt
tail -3 gpurun_out/${TAG}_bench.err
timeout 1500 python -m pytest tests/test_gpu_bench.py tests/test_gpu_mc.py tests/test_gpu_options.py tests/test_gpu_store_pacing.py -m gpu -q -x > gpurun_out/${TAG}_pytest.log 2>&1; tail -5 gpurun_out/${TAG}_pytest.log
