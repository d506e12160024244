# rocprofv3 over tools/latency_modes.py for the three kernels of the latency-bound modes (general; transition rows; K-step table
# for the statistics-only launches).
# Kernel trace and each PMC group are SEPARATE runs.  Usage (through gpurun): bash tools/gpu_profile_latency.sh <tag>
REPO=$GRAFT_REPO_ROOT
TAG=${1:-r02c}
OUT=$REPO/gpurun_out/lat_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for rows in 0 1 2; do
  if [ $rows = 2 ]; then export GU_ROLLOUT_ROWS=1 GU_ROLLOUT_MULTI=1; else export GU_ROLLOUT_ROWS=$rows GU_ROLLOUT_MULTI=0; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_rows$rows -- python3 $REPO/tools/latency_modes.py > $OUT/kt_rows$rows.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_rows$rows -- python3 $REPO/tools/latency_modes.py > $OUT/pmc_rows$rows.log 2>&1
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
out, tag = sys.argv[1], sys.argv[2]
res = {}
for rows in ('0', '1', '2'):
    per = defaultdict(list)
    for p in glob.glob(os.path.join(out, 'kt_rows' + rows, '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(p, newline='')):
            if 'rollout' in r['Kernel_Name']:
                per[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    pmc = defaultdict(lambda: defaultdict(list))
    for p in glob.glob(os.path.join(out, 'pmc_rows' + rows, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(p, newline='')):
            if 'rollout' in r['Kernel_Name']:
                pmc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    res[{'0': 'general kernel (GU_ROLLOUT_ROWS=0 GU_ROLLOUT_MULTI=0)', '1': 'row table (GU_ROLLOUT_ROWS=1 GU_ROLLOUT_MULTI=0)',
         '2': 'K-step table for statistics-only launches (defaults)'}[rows]] = {k: dict(calls=len(v), avg_us=sum(v) / len(v) / 1e3, min_us=min(v) / 1e3,
                                              pmc_avg_per_dispatch={c: sum(x) / len(x) for c, x in sorted(pmc[k].items())})
                                      for k, v in per.items()}
json.dump(res, open(os.path.join(out, tag + '_latency_modes_rocprof.json'), 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
