# round 5, final session: smoke, the -m gpu suite, rocprofv3 passes over bench.py, the bench line
cd $GRAFT_REPO_ROOT
TAG=r05z
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1; grep -E "passed|failed|rror" gpurun_out/${TAG}_pytest_gpu.log | tail -5
bash tools/gpu_profile.sh $TAG > gpurun_out/profile_$TAG.log 2>&1; tail -25 gpurun_out/profile_$TAG.log | cut -c1-300
cd $GRAFT_REPO_ROOT
cp gpurun_out/prof_$TAG/summary/* gpurun_out/ 2>/dev/null
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench_line.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05z_bench_line.json'))
print('value %.4g ms_per_step %.5f frac %.3f frac_wall %.3f traffic %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_wall'], d['roofline'].get('traffic_over_algorithmic')))
print('configs', {k: (v.get('us_per_launch') or v.get('us_per_round')) for k, v in (d.get('configs') or {}).items()})
print('other', {k: v.get('ms_per_launch') for k, v in (d.get('other_modes') or {}).items()})
print('checks', d.get('bit_exact_vs_reference_digest'), d.get('bit_exact_vs_oracle'), (d.get('final_state_vs_oracle') or {}).get('equal'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
print('pacing', json.dumps(d['roofline']['store_pacing'].get('last_launches'))[:600])
PY
tail -3 gpurun_out/${TAG}_bench.err
