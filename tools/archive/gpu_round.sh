# One GPU session: smoke, the -m gpu suite, the bench line, API latencies, then the rocprofv3 passes.
# Usage (through gpurun): bash tools/gpu_round.sh <tag> [skip-tests]
cd $GRAFT_REPO_ROOT
TAG=${1:-r02a}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
if [ "$2" != "skip-tests" ]; then
  timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1; grep -E "passed|failed|rror" gpurun_out/${TAG}_pytest_gpu.log | tail -5
fi
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench_line.json; cat gpurun_out/${TAG}_bench_line.json; tail -5 gpurun_out/${TAG}_bench.err
timeout 600 python tools/api_latency.py > gpurun_out/${TAG}_api_latency.json 2>&1; cat gpurun_out/${TAG}_api_latency.json
bash tools/gpu_profile.sh $TAG > gpurun_out/profile_$TAG.log 2>&1; tail -25 gpurun_out/profile_$TAG.log
cd $GRAFT_REPO_ROOT
cp gpurun_out/prof_$TAG/summary/* gpurun_out/ 2>/dev/null
