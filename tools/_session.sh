python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_round2.py tests/test_gpu_mc.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
for rep in 1 2; do
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1 > gpurun_out/r03n_bench_$rep.json
done
PACE_AB_K=50 timeout 600 python tools/pace_ab.py 10 65536 2>&1 | cut -c1-200 | tail -12 > gpurun_out/r03n_pace_auto.txt
