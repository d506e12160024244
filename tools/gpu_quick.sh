# scratch session for gpurun (edited per experiment): the tools of the last one, as an example
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
GU_LIB_PATH=$GRAFT_REPO_ROOT/griduniverse_amd/lib/libgu_torn.so timeout 1000 python tools/xcd_stress.py 840 gpurun_out/r05zz_xcd_torn.txt 2>&1 | tail -5
