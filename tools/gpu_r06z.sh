# round 6, full session: smoke, the -m gpu suite, rocprofv3 passes over bench.py, the bench line as the driver runs it
cd $GRAFT_REPO_ROOT
TAG=${1:-r06z}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1; grep -E "passed|failed|rror" gpurun_out/${TAG}_pytest_gpu.log | tail -5
bash tools/gpu_profile.sh $TAG > gpurun_out/profile_$TAG.log 2>&1; tail -30 gpurun_out/profile_$TAG.log | cut -c1-260
cd $GRAFT_REPO_ROOT
cp gpurun_out/prof_$TAG/summary/* gpurun_out/ 2>/dev/null
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/${TAG}_bench_detail.json 2> gpurun_out/${TAG}_bench.err > gpurun_out/${TAG}_bench_line.json
echo "bench rc $? bytes $(wc -c < gpurun_out/${TAG}_bench_line.json) lines $(wc -l < gpurun_out/${TAG}_bench_line.json)"
cat gpurun_out/${TAG}_bench_line.json
tail -3 gpurun_out/${TAG}_bench.err
rm -rf gpurun_out/prof_$TAG/kt gpurun_out/prof_$TAG/pmc_*   # (the raw CSVs: tens of MB; the summaries are kept)
