# round 5, session h: gain / dec / adaptive aim matrix on one box (fresh engine per run, placement search on)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2 3; do
for v in "256 16 0" "64 4 0" "128 8 0" "64 4 1" "256 16 1" "32 2 0"; do
  set -- $v
  timeout 300 python tools/pace_loop.py --kind c3 --launches 2900 --inc $1 --dec $2 --adapt $3 --no-search --summary 2>&1 | grep SUMMARY
done
done | tee gpurun_out/r05h_matrix.txt
timeout 300 python tools/pace_loop.py --kind c3 --launches 600 --sweep 154 170 2 > gpurun_out/r05h_sweep.txt 2>&1; grep -A14 "== sweep" gpurun_out/r05h_sweep.txt | cut -c1-200
