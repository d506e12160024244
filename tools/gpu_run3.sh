cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -o /tmp/store_ceiling tools/micro/store_ceiling.hip && /tmp/store_ceiling 65536 | tee gpurun_out/store_ceiling.txt
/tmp/store_ceiling 1048576 | tail -12 | tee -a gpurun_out/store_ceiling.txt
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
for bs in 64 128 256; do GU_ROLLOUT_BLOCK=$bs python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bs=$bs', d['value'], d['roofline']['achieved'], d['roofline']['launch_ms'])"; done | tee gpurun_out/bench_v2a.txt
