# round 6, second session: the whole -m gpu suite on the 64-bit step count; two GU_TEST_OPTIONS sessions with a kernel trace each
cd $GRAFT_REPO_ROOT
TAG=r06b
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_pytest_gpu.log 2>&1; tail -15 gpurun_out/${TAG}_pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
for sw in "" "rollout_rows=1,traj_layout=1" "rollout_rows=0,rollout_multi=0"; do
  name=$(echo "default_$sw" | tr ',=' '__')
  GU_TEST_OPTIONS="$sw" rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kt_$name -- python3 -m pytest $GRAFT_REPO_ROOT/tests/test_gpu_parity.py -m gpu -q -x -k "golden_rollout or reference_digests" -p no:cacheprovider > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kt_$name.log 2>&1
  echo "== GU_TEST_OPTIONS=$sw"; grep -E "GU_TEST_OPTIONS|passed|failed" $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kt_$name.log | tail -3
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kt_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cut -d, -f1,2 "$f" | grep -i "rollout" | head -12
done 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/${TAG}_dispatch_by_options.txt
