# round 5, session a: closed-loop pacing traces + trajectory layout A/B (through gpurun)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rows_kernel.py -x -q -m gpu 2>&1 | tail -4
timeout 300 python tools/pace_loop.py --kind c3 --launches 600 --sweep 150 200 2 --json gpurun_out/r05a_pace_c3.json > gpurun_out/r05a_pace_c3.txt 2>&1; cat gpurun_out/r05a_pace_c3.txt
for k in sample stream c4 packed; do
  timeout 300 python tools/pace_loop.py --kind $k --launches 300 --json gpurun_out/r05a_pace_$k.json > gpurun_out/r05a_pace_$k.txt 2>&1; grep -v "^      " gpurun_out/r05a_pace_$k.txt | cut -c1-600
done
timeout 300 python tools/pace_loop.py --kind c3 --launches 300 --late 64 --ok 16 --no-search > gpurun_out/r05a_pace_c3_64_16.txt 2>&1; cut -c1-600 gpurun_out/r05a_pace_c3_64_16.txt
timeout 300 python tools/pace_loop.py --kind c3 --launches 300 --late 16 --ok 4 --no-search > gpurun_out/r05a_pace_c3_16_4.txt 2>&1; cut -c1-600 gpurun_out/r05a_pace_c3_16_4.txt
timeout 900 python tools/layout_ab.py --json gpurun_out/r05a_layout_ab.json > gpurun_out/r05a_layout_ab.txt 2>&1; cat gpurun_out/r05a_layout_ab.txt
