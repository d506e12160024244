# round 5, session b: closed-loop pacing v2 (bucketed reports, stochastic approximation) + layout A/B again
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python tools/pace_loop.py --kind c3 --launches 1800 --waves 300 --sweep 152 182 2 --json gpurun_out/r05b_pace_c3.json > gpurun_out/r05b_pace_c3.txt 2>&1; cut -c1-700 gpurun_out/r05b_pace_c3.txt
for k in sample c4 packed; do
  timeout 300 python tools/pace_loop.py --kind $k --launches 900 --json gpurun_out/r05b_pace_$k.json > gpurun_out/r05b_pace_$k.txt 2>&1; grep -v "^      " gpurun_out/r05b_pace_$k.txt | cut -c1-500
done
timeout 300 python tools/pace_loop.py --kind c3 --launches 1200 --inc 64 --dec 1 --no-search > gpurun_out/r05b_pace_c3_64_1.txt 2>&1; grep -v "^      " gpurun_out/r05b_pace_c3_64_1.txt | cut -c1-500
timeout 300 python tools/pace_loop.py --kind c3 --launches 1200 --inc 256 --dec 4 --no-search > gpurun_out/r05b_pace_c3_256_4.txt 2>&1; grep -v "^      " gpurun_out/r05b_pace_c3_256_4.txt | cut -c1-500
timeout 900 python tools/layout_ab.py --json gpurun_out/r05b_layout_ab.json > gpurun_out/r05b_layout_ab.txt 2>&1; cat gpurun_out/r05b_layout_ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rows_kernel.py -x -q -m gpu 2>&1 | tail -4
