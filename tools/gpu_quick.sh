cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/pace_loop.py --kind c3 --summary > gpurun_out/r05q_pace_loop_c3.txt 2>&1; tail -15 gpurun_out/r05q_pace_loop_c3.txt
timeout 900 python -m pytest tests/test_gpu_store_pacing.py tests/test_gpu_traj_layout.py -q -m gpu > gpurun_out/r05q_pytest.txt 2>&1; grep -E " passed| failed|rror" gpurun_out/r05q_pytest.txt | tail -5
for i in 1 2 3; do python bench.py --steps 400 --warmup 50 2>/dev/null | tail -1 >> gpurun_out/r05q_bench.jsonl; done; cat gpurun_out/r05q_bench.jsonl | cut -c1-400
