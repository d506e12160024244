# Long random-grid soak of the property tests (rollouts, table policies, DP, Monte-Carlo) under several dispatches
# (GU_TEST_OPTIONS: see tools/gpu_soak_switches.sh).  Usage (through gpurun): bash tools/gpu_fuzz.sh <tag> <trials>
cd $GRAFT_REPO_ROOT
TAG=${1:-r06}
TRIALS=${2:-1500}
for sw in "11 " "12 rollout_rows=1,rollout_multi=1" "13 vi_path=1" "14 rollout_rows=0,rollout_multi=0" "15 rollout_rows=1,rollout_entry=0" "16 rollout_rows=1,traj_layout=1"; do
  set -- $sw
  echo "== GU_FUZZ_TRIALS=$TRIALS GU_FUZZ_SEED=$1 GU_TEST_OPTIONS=$2"
  GU_FUZZ_TRIALS=$TRIALS GU_FUZZ_SEED=$1 GU_TEST_OPTIONS="$2" timeout 3000 python -m pytest tests -m gpu -q -k "property" 2>&1 | grep -E "GU_TEST_OPTIONS|passed|failed|rror|^FAILED|^E  |assert" | tail -14
done 2>&1 | tee gpurun_out/${TAG}_fuzz_${TRIALS}_trials.txt
