# round 5, session e: closed-loop pacing v4b (decide() with its loads in flight together)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python tools/pace_loop.py --kind c3 --launches 1800 --sweep 154 176 2 --json gpurun_out/r05e_pace_c3.json > gpurun_out/r05e_pace_c3.txt 2>&1; cut -c1-700 gpurun_out/r05e_pace_c3.txt
for k in sample c4 packed; do
  timeout 300 python tools/pace_loop.py --kind $k --launches 900 --json gpurun_out/r05e_pace_$k.json > gpurun_out/r05e_pace_$k.txt 2>&1; grep -v "^      " gpurun_out/r05e_pace_$k.txt | cut -c1-500
done
timeout 300 python tools/pace_loop.py --kind c3 --launches 1200 --inc 512 --dec 16 --no-search > gpurun_out/r05e_pace_c3_512_16.txt 2>&1; grep -v "^      " gpurun_out/r05e_pace_c3_512_16.txt | cut -c1-500
timeout 300 python tools/pace_loop.py --kind c3 --launches 1200 --inc 256 --dec 32 --no-search > gpurun_out/r05e_pace_c3_256_32.txt 2>&1; grep -v "^      " gpurun_out/r05e_pace_c3_256_32.txt | cut -c1-500
timeout 600 python tools/layout_ab.py --sizes 4096 8192 --json gpurun_out/r05e_layout_sizes.json > gpurun_out/r05e_layout_sizes.txt 2>&1; cat gpurun_out/r05e_layout_sizes.txt
