# scratch session for gpurun (edited per experiment)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/sample_copies.py
