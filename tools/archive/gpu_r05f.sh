# round 5, session f: closed-loop pacing v5 (does the limiter pay at all?) + the whole GPU suite + the bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for k in c3 packed c4 sample; do
  timeout 300 python tools/pace_loop.py --kind $k --launches 1740 --json gpurun_out/r05f_pace_$k.json > gpurun_out/r05f_pace_$k.txt 2>&1; grep -v "^      " gpurun_out/r05f_pace_$k.txt | cut -c1-600
done
timeout 1800 python -m pytest tests -q -m gpu > gpurun_out/r05f_pytest.txt 2>&1; grep -E "passed|failed|rror" gpurun_out/r05f_pytest.txt | tail -8
timeout 300 python tools/placement_loop.py 10 --json gpurun_out/r05f_placement_loop.json > gpurun_out/r05f_placement_loop.txt 2>&1; cat gpurun_out/r05f_placement_loop.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r05f_bench.err | tail -1 > gpurun_out/r05f_bench_line.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/r05f_bench_line.json'))
print('value %.4g ms_per_step %.5f frac %.3f frac_wall %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_wall']))
print('pacing', json.dumps(d['roofline']['store_pacing'])[:1500])
print('configs', {k: (v.get('us_per_launch') or v.get('us_per_round')) for k, v in (d.get('configs') or {}).items()})
print('other', {k: v.get('ms_per_launch') for k, v in (d.get('other_modes') or {}).items()})
print('checks', d.get('bit_exact_vs_reference_digest'), d.get('bit_exact_vs_oracle'), (d.get('final_state_vs_oracle') or {}).get('equal'))
PY
tail -5 gpurun_out/r05f_bench.err
