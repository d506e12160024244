set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocminfo | grep -E "Marketing|gfx9" | head -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25
timeout 600 python bench.py --steps 20 --warmup 3 2>&1 | tail -5 | tee gpurun_out/bench_first.log
