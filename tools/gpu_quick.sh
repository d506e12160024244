cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
GU_LIB_PATH=$GRAFT_REPO_ROOT/griduniverse_amd/lib/libgu_torn.so timeout 3100 python tools/xcd_stress.py 2900 gpurun_out/r05w_xcd_torn.txt 2>&1 | tail -5
