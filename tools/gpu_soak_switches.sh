# The whole -m gpu suite under every launch-shape switch of the rollout kernels (each is read per launch, so forcing it
# routes every eligible launch of every parity test through that variant).  Usage (through gpurun): bash tools/gpu_soak_switches.sh <tag>
cd $GRAFT_REPO_ROOT
TAG=${1:-r02b}
for sw in "GU_ROLLOUT_ROWS=1" "GU_ROLLOUT_MULTI=1" "GU_ROLLOUT_MULTI=0 GU_ROLLOUT_ROWS=1" "GU_ROLLOUT_ROWS=0" "GU_ROLLOUT_XCD=1" "GU_ROLLOUT_BLOCK=1024" "GU_ROLLOUT_BLOCK=64 GU_ROLLOUT_XCD=1 GU_ROLLOUT_ROWS=1"; do
  echo "== $sw"
  env $sw timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|rror|^FAILED|assert" | tail -8
done 2>&1 | tee gpurun_out/${TAG}_soak_switches.txt
