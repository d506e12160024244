# round 5, session c: closed-loop pacing v3 (early fetch + early report, proportional step up) + layout crossover
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python tools/pace_loop.py --kind c3 --launches 1800 --sweep 156 176 2 --json gpurun_out/r05c_pace_c3.json > gpurun_out/r05c_pace_c3.txt 2>&1; cut -c1-700 gpurun_out/r05c_pace_c3.txt
for k in sample c4 packed; do
  timeout 300 python tools/pace_loop.py --kind $k --launches 900 --json gpurun_out/r05c_pace_$k.json > gpurun_out/r05c_pace_$k.txt 2>&1; grep -v "^      " gpurun_out/r05c_pace_$k.txt | cut -c1-500
done
timeout 300 python tools/pace_loop.py --kind c3 --launches 1200 --inc 64 --dec 24 --no-search > gpurun_out/r05c_pace_c3_64_24.txt 2>&1; grep -v "^      " gpurun_out/r05c_pace_c3_64_24.txt | cut -c1-500
timeout 300 python tools/pace_loop.py --kind c3 --launches 1200 --inc 32 --dec 6 --no-search > gpurun_out/r05c_pace_c3_32_6.txt 2>&1; grep -v "^      " gpurun_out/r05c_pace_c3_32_6.txt | cut -c1-500
timeout 900 python tools/layout_ab.py --sizes 4096 8192 16384 24576 --json gpurun_out/r05c_layout_sizes.json > gpurun_out/r05c_layout_sizes.txt 2>&1; cat gpurun_out/r05c_layout_sizes.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rows_kernel.py -x -q -m gpu > gpurun_out/r05c_pytest.txt 2>&1; grep -E "passed|failed|rror" gpurun_out/r05c_pytest.txt | tail -5
