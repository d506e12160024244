# sustained launch time (50 launches per figure) of the rollout kernel with sc1 stores at fixed idle amounts, several buffers
PACE_AB_K=50 GU_LIB_PATH=$PWD/griduniverse_amd/lib/libgu_aux16.so timeout 600 python tools/pace_ab.py 8 65536 6 7 8 9 10 11 12 13 14 16 18 20 24 2>&1 | tail -11 | cut -c1-40,150-400
