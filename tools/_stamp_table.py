import re, json, sys
rows = []; cur = None
for ln in open(sys.argv[1]):
    if ln.startswith('=='):
        cur = [ln.strip()]; rows.append(cur)
    m = re.search(r': ([0-9.]+),?$', ln)
    if m and cur is not None:
        cur.append(float(m.group(1)))
print("order: V1, keys, issue(+reduce), B1, wait, collect, toLDS, B2, V2, agents, total, polls")
for r in rows:
    print(r[0], ' '.join('%6.0f' % x for x in r[1:13]))
if len(sys.argv) > 2:
    d = json.load(open(sys.argv[2]))
    for k, v in d['forms'].items():
        print(k, v.get('form_taken'), v.get('us_per_round_median'))
