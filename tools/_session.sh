GU_DEBUG=2 PACE_AB_K=20 timeout 900 python tools/pace_ab.py 3 131072 420 400 385 370 360 350 340 330 320 310 2>&1 | grep -v "traj\|placement" | cut -c1-330
