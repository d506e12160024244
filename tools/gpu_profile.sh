# rocprofv3 passes over bench.py (run on the GPU box via gpurun).  Kernel-trace/stats and each
# PMC group are SEPARATE runs (the pool refuses --pmc combined with the trace domains).
set -x
REPO=$GRAFT_REPO_ROOT
TAG=${1:-r01}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# kernel trace: the driver's own bench command (minus the CPU legs, which launch nothing); counters: a short timed region
KT="python3 $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic"
BENCH="python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-checks --no-strong-c4 --no-other-modes --min-seconds 0.02 --no-live-traffic"
# HBM traffic for EVERY launch form the line reports (headline, strong_c4, packed rows, statistics only): summarize_profile.py
# tells them apart by kernel name and launch size
TRAFFIC="python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-checks --min-seconds 0.02 --no-live-traffic"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $KT > $OUT/kt.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $TRAFFIC > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $TRAFFIC > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1
find $OUT -name "*.csv" | head -40
tail -3 $OUT/kt.log
python3 $REPO/tools/summarize_profile.py $OUT $TAG
