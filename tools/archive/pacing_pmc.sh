# Memory-side counters of the rollout launch unthrottled / on a schedule / on too short a schedule: one rocprofv3 --pmc pass per
# counter group (counters only, never combined with trace domains).  Usage (through gpurun): bash tools/pacing_pmc.sh <tag>
TAG=${1:-r03o}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_pacing_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for group in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
             "TCC_EA0_WRREQ_LEVEL_sum TCC_BUSY_sum TCC_TAG_STALL_sum" \
             "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCC_NORMAL_WRITEBACK_sum" \
             "TCP_PENDING_STALL_CYCLES_sum TA_BUSY_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
             "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $group -d $OUT/pass$i --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/pacing_pmc.py > $OUT/pass$i.stdout 2> $OUT/pass$i.stderr
  tail -1 $OUT/pass$i.stderr | cut -c1-160
done
cd $GRAFT_REPO_ROOT
python tools/pacing_pmc.py parse $OUT | tee gpurun_out/${TAG}_pacing_pmc.txt
