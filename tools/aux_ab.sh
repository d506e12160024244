python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_round2.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
for rep in 1 2 3; do for v in "" _aux16p; do
GU_LIB_PATH=$PWD/griduniverse_amd/lib/libgu$v.so timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1 > gpurun_out/r03m_bench${v}_$rep.json
done; done
