for sp in 0 4 0 4; do echo "== split $sp"; PACE_AB_SPLIT=$sp PACE_AB_K=20 GU_DEBUG=1 timeout 600 python tools/pace_ab.py 4 262144 2>&1 | grep -v "traj\|placement" | cut -c1-170; done
