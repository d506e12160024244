# The whole -m gpu suite under every launch-shape switch of the rollout kernels (each is read per launch, so forcing it
# routes every eligible launch of every parity test through that variant).  Usage (through gpurun): bash tools/gpu_soak_switches.sh <tag>
cd $GRAFT_REPO_ROOT
TAG=${1:-r02b}
for sw in "GU_ROLLOUT_ROWS=1" "GU_ROLLOUT_MULTI=1" "GU_ROLLOUT_MULTI=0 GU_ROLLOUT_ROWS=1" "GU_ROLLOUT_ROWS=0" "GU_ROLLOUT_XCD=1" "GU_ROLLOUT_BLOCK=1024" "GU_ROLLOUT_BLOCK=64 GU_ROLLOUT_XCD=1 GU_ROLLOUT_ROWS=1" \
          "GU_ROLLOUT_ROWS=1 GU_ROLLOUT_ENTRY=0" "GU_ROLLOUT_ROWS=1 GU_TRAJ_LAYOUT=1" "GU_ROLLOUT_ROWS=3 GU_TRAJ_LAYOUT=1 GU_ROLLOUT_HALF_WAVES=1" "GU_ROLLOUT_ROWS=1 GU_ROWS_COPIES=1" "GU_ROLLOUT_ROWS=1 GU_ROWS_COPIES=2" "GU_ROLLOUT_ROWS=1 GU_ROWS_COPIES=32 GU_TRAJ_LAYOUT=0" "GU_ROLLOUT_ROWS=2 GU_ROLLOUT_PACE=150 GU_SOAK_SKIP_PACING_TESTS=1"; do
  echo "== $sw"
  skip=""; case "$sw" in *GU_SOAK_SKIP_PACING_TESTS*) skip="--deselect tests/test_gpu_store_pacing.py";; esac  # (those tests are about the DEFAULT pacing)
  env $sw timeout 1500 python -m pytest tests -m gpu -q -x $skip 2>&1 | grep -E "passed|failed|rror|^FAILED|assert" | tail -8
done 2>&1 | tee gpurun_out/${TAG}_soak_switches.txt
