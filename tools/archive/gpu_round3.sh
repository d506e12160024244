# One GPU session of round 3: smoke, the -m gpu suite, the driver's bench command, the two new ways to start bench.py with
# --gpus 2 (both sharing the one GPU of the box), then the rocprofv3 passes (kernel trace + per-mode PMC).
# Usage (through gpurun): bash tools/gpu_round3.sh <tag> [skip-tests]
cd $GRAFT_REPO_ROOT
TAG=${1:-r03a}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
if [ "$2" != "skip-tests" ]; then
  timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/${TAG}_pytest_gpu.log | tail -12
fi
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench_line.json; head -c 1500 gpurun_out/${TAG}_bench_line.json; echo; tail -3 gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/${TAG}_bench_2ranks.err | tail -1 > gpurun_out/${TAG}_bench_2ranks_one_gpu.json; head -c 600 gpurun_out/${TAG}_bench_2ranks_one_gpu.json; echo; tail -3 gpurun_out/${TAG}_bench_2ranks.err
timeout 600 python bench.py --gpus 2 --single-process --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/${TAG}_bench_sp.err | tail -1 > gpurun_out/${TAG}_bench_single_process_one_gpu.json; head -c 600 gpurun_out/${TAG}_bench_single_process_one_gpu.json; echo; tail -3 gpurun_out/${TAG}_bench_sp.err
bash tools/gpu_profile.sh $TAG > gpurun_out/profile_$TAG.log 2>&1; tail -25 gpurun_out/profile_$TAG.log
cd $GRAFT_REPO_ROOT
cp gpurun_out/prof_$TAG/summary/* gpurun_out/ 2>/dev/null
cat gpurun_out/rollout_pmc_latest.json | head -60
