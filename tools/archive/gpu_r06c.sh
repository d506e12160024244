# round 6: the whole -m gpu suite (no -x), RCCL refusal strings kept
cd $GRAFT_REPO_ROOT
TAG=${1:-r06c}
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -q -rP -k "not property or True" > gpurun_out/${TAG}_pytest_gpu.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR|RCCL with|ncclCommInitAll with" gpurun_out/${TAG}_pytest_gpu.log | tail -30
