cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python tools/bench_configs.py gpurun_out/bench_configs_r01b.json 2>&1 | tail -120
bash tools/gpu_profile.sh r01b 2>&1 | tail -40
cd $GRAFT_REPO_ROOT && python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_r01b.json
