# Long random-grid soak of the property tests (rollouts, table policies, DP, Monte-Carlo) under both rollout dispatches.
# Usage (through gpurun): bash tools/gpu_fuzz.sh <tag> <trials>
cd $GRAFT_REPO_ROOT
TAG=${1:-r02d}
TRIALS=${2:-1500}
for sw in "GU_FUZZ_SEED=11" "GU_FUZZ_SEED=12 GU_ROLLOUT_ROWS=1 GU_ROLLOUT_MULTI=1" "GU_FUZZ_SEED=13 GU_VI_CLUSTER=0" "GU_FUZZ_SEED=14 GU_ROLLOUT_ROWS=0" "GU_FUZZ_SEED=15 GU_ROLLOUT_ROWS=1 GU_ROLLOUT_ENTRY=0" "GU_FUZZ_SEED=16 GU_ROLLOUT_ROWS=3 GU_TRAJ_LAYOUT=1"; do
  echo "== GU_FUZZ_TRIALS=$TRIALS $sw"
  env GU_FUZZ_TRIALS=$TRIALS $sw timeout 3000 python -m pytest tests -m gpu -q -k "property" 2>&1 | grep -E "passed|failed|rror|^FAILED|^E  |assert" | tail -14
done 2>&1 | tee gpurun_out/${TAG}_fuzz_${TRIALS}_trials.txt
