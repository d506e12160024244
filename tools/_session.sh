GU_DEBUG=2 PACE_AB_K=10 timeout 900 python tools/pace_ab.py 2 1048576 2>&1 | grep -v "traj\|placement" | cut -c1-220
GU_DEBUG=1 PACE_AB_K=10 timeout 900 python tools/pace_ab.py 2 524288 2>&1 | grep -v "traj\|placement" | cut -c1-220
