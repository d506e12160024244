# round 5, session d: closed-loop pacing v4 (plain-store slots, the first wave decides, share-of-waves rule)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python tools/pace_loop.py --kind c3 --launches 1800 --sweep 154 176 2 --json gpurun_out/r05d_pace_c3.json > gpurun_out/r05d_pace_c3.txt 2>&1; cut -c1-700 gpurun_out/r05d_pace_c3.txt
for k in sample c4 packed stream; do
  timeout 300 python tools/pace_loop.py --kind $k --launches 900 --json gpurun_out/r05d_pace_$k.json > gpurun_out/r05d_pace_$k.txt 2>&1; grep -v "^      " gpurun_out/r05d_pace_$k.txt | cut -c1-500
done
timeout 300 python tools/pace_loop.py --kind c3 --launches 1200 --inc 512 --dec 16 --no-search > gpurun_out/r05d_pace_c3_512_16.txt 2>&1; grep -v "^      " gpurun_out/r05d_pace_c3_512_16.txt | cut -c1-500
timeout 300 python tools/pace_loop.py --kind c3 --launches 1200 --inc 256 --dec 32 --no-search > gpurun_out/r05d_pace_c3_256_32.txt 2>&1; grep -v "^      " gpurun_out/r05d_pace_c3_256_32.txt | cut -c1-500
timeout 300 python tools/pace_loop.py --kind c3 --launches 1200 --inc 128 --dec 8 --no-search > gpurun_out/r05d_pace_c3_128_8.txt 2>&1; grep -v "^      " gpurun_out/r05d_pace_c3_128_8.txt | cut -c1-500
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rows_kernel.py tests/test_gpu_dp.py tests/test_gpu_render.py -x -q -m gpu > gpurun_out/r05d_pytest.txt 2>&1; grep -E "passed|failed|rror" gpurun_out/r05d_pytest.txt | tail -5
