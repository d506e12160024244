# round 5, session g: the slow loop around the rule (adaptive aim)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do
  timeout 300 python tools/pace_loop.py --kind c3 --launches 2900 --json gpurun_out/r05g_pace_c3_$i.json > gpurun_out/r05g_pace_c3_$i.txt 2>&1; grep -v "^      \|first 40" gpurun_out/r05g_pace_c3_$i.txt | cut -c1-700
done
timeout 300 python tools/pace_loop.py --kind sample --launches 1740 --no-search > gpurun_out/r05g_pace_sample.txt 2>&1; grep -v "^      \|first 40" gpurun_out/r05g_pace_sample.txt | cut -c1-700
timeout 300 python tools/placement_loop.py 10 --json gpurun_out/r05g_placement_loop.json > gpurun_out/r05g_placement_loop.txt 2>&1; cat gpurun_out/r05g_placement_loop.txt
timeout 900 python -m pytest tests/test_gpu_store_pacing.py -q -m gpu > gpurun_out/r05g_pytest.txt 2>&1; grep -E "passed|failed|rror" gpurun_out/r05g_pytest.txt | tail -8
