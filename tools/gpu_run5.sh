cd $GRAFT_REPO_ROOT
REPO=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_r01c
python tools/bench_configs.py gpurun_out/bench_configs_r01c.json > gpurun_out/bench_configs_r01c.log 2>&1
tail -12 gpurun_out/bench_configs_r01c.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_r01c/kt_all -- python3 $REPO/tools/bench_configs.py > $REPO/gpurun_out/prof_r01c/kt_all.log 2>&1
cat $REPO/gpurun_out/prof_r01c/kt_all/*/*kernel_stats.csv | head -40
cd $REPO && python examples/griduniverse_alg_examples.py > gpurun_out/example_alg.log 2>&1; tail -25 gpurun_out/example_alg.log
