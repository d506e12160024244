cd $GRAFT_REPO_ROOT
TAG=${1:-r01d}
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|rror" gpurun_out/pytest_gpu.log | tail -3
bash tools/gpu_profile.sh $TAG > gpurun_out/profile_$TAG.log 2>&1; tail -30 gpurun_out/profile_$TAG.log
cd $GRAFT_REPO_ROOT
cp gpurun_out/prof_$TAG/summary/rollout_pmc_latest.json profiles/rollout_pmc_latest.json
python bench.py 2>&1 | tail -1 > gpurun_out/bench_$TAG.json; cat gpurun_out/bench_$TAG.json
python tools/bench_configs.py gpurun_out/bench_configs_$TAG.json > gpurun_out/bench_configs_$TAG.log 2>&1; tail -3 gpurun_out/bench_configs_$TAG.log
