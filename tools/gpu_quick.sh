# scratch session for gpurun (edited per experiment)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for sw in "GU_FUZZ_SEED=21" "GU_FUZZ_SEED=22 GU_ROLLOUT_ROWS=1 GU_ROLLOUT_MULTI=1" "GU_FUZZ_SEED=23 GU_ROLLOUT_ROWS=1 GU_ROWS_COPIES=2" "GU_FUZZ_SEED=24 GU_ROLLOUT_ROWS=3 GU_TRAJ_LAYOUT=1 GU_ROLLOUT_HALF_WAVES=1"; do
  echo "== GU_FUZZ_TRIALS=4000 $sw"
  env GU_FUZZ_TRIALS=4000 $sw timeout 1500 python -m pytest tests -m gpu -q -k "property" 2>&1 | grep -E "passed|failed|rror|^FAILED|^E  |assert" | tail -6
done 2>&1 | tee gpurun_out/r05zz_fuzz_4000_trials.txt
