cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -o /tmp/store_ceiling tools/micro/store_ceiling.hip && /tmp/store_ceiling 65536 | tee gpurun_out/store_ceiling.txt
/tmp/store_ceiling 1048576 | tail -12 | tee -a gpurun_out/store_ceiling.txt
bash tools/gpu_profile.sh r01a 2>&1 | tail -60
