# Counters of the store loop per buffer, next to the buffer's write-rate class: one rocprofv3 --pmc pass per counter group
# (never combined with trace domains).  Usage (through gpurun): bash tools/placement_pmc.sh <tag>
cd $GRAFT_REPO_ROOT/tools/micro
TAG=${1:-r02h}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_place_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BIN=$GRAFT_REPO_ROOT/tools/archive/micro/placement_pmc
i=0
for group in "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_GMI_32B_sum TCC_EA0_WRREQ_WRITE_IO_32B_sum TCC_EA0_WRREQ_sum" \
             "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum" \
             "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
             "TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum GRBM_UTCL2_BUSY" \
             "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_WRREQ_LEVEL_sum" \
             "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCC_NORMAL_WRITEBACK_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $group -d $OUT/pass$i --output-format csv -- $BIN 10 > $OUT/pass$i.stdout 2> $OUT/pass$i.stderr
  tail -2 $OUT/pass$i.stderr | cut -c1-200
done
cd $GRAFT_REPO_ROOT
python tools/placement_pmc_parse.py $OUT | tee gpurun_out/${TAG}_placement_pmc.txt
