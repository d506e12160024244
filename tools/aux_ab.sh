python -m pytest tests -m gpu -x -q --durations=14 > gpurun_out/r03l_pytest_full2.txt 2>&1; grep -E "passed|failed" gpurun_out/r03l_pytest_full2.txt
GU_DEBUG=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r03l_bench2.err | tail -1 > gpurun_out/r03l_bench2.json
grep "store pacing" gpurun_out/r03l_bench2.err
