# The whole -m gpu suite under several sets of launch-shape options of the rollout kernels.  tests/conftest.py turns
# GU_TEST_OPTIONS into process-wide gu_set_option defaults at session start and prints what gu_get_option reports for each;
# every eligible launch of every parity test then goes through that variant.  (Rounds 3-5 exported GU_ROLLOUT_ROWS=... here, which
# the product library had stopped reading in round 3: those logs are the default dispatch, see profiles/README.)
# Usage (through gpurun): bash tools/gpu_soak_switches.sh <tag> [sets...]
cd $GRAFT_REPO_ROOT
TAG=${1:-r06}
shift
SETS=("$@")
if [ ${#SETS[@]} -eq 0 ]; then
  SETS=("rollout_rows=1" "rollout_multi=1" "rollout_multi=0,rollout_rows=1" "rollout_rows=0,rollout_multi=0" "rollout_xcd=1" "rollout_block=1024"
        "rollout_block=64,rollout_xcd=1,rollout_rows=1" "rollout_rows=1,rollout_entry=0" "rollout_rows=1,traj_layout=1"
        "rollout_rows=1,traj_layout=1,rollout_half_waves=1" "rollout_rows=1,rows_copies=2" "rollout_rows=1,rows_copies=32,traj_layout=0"
        "rollout_rows=2,rollout_pace=150")
fi
for sw in "${SETS[@]}"; do
  echo "== GU_TEST_OPTIONS=$sw"
  skip=""  # (tests/test_gpu_store_pacing.py, about the DEFAULT pacing, skips itself under GU_TEST_OPTIONS)
  GU_TEST_OPTIONS="$sw" timeout 1500 python -m pytest tests -m gpu -q $skip 2>&1 | grep -E "GU_TEST_OPTIONS|passed|failed|rror|^FAILED|assert" | tail -8
done 2>&1 | tee gpurun_out/${TAG}_soak_switches.txt
