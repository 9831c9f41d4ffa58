"""Summarise rocprofv3 csv output of scripts/profile_1f.sh: per-kernel time and counter means."""
import csv, glob, os, sys, collections
out = sys.argv[1]
def find(sub, pat):
    return sorted(glob.glob(os.path.join(out, sub, '**', pat), recursive=True))
lines = []
for f in find('trace', '*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        lines.append(f"STATS {r.get('Name','')[:90]:90s} calls={r.get('Calls')} avg_ns={r.get('AverageNs')} total_ns={r.get('TotalDurationNs')} pct={r.get('Percentage')}")
for sub in ('sq', 'sq2', 'fetch', 'write'):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in find(sub, '*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:75]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        if 'picard' not in k and 'tridiag' not in k: continue
        for c, v in sorted(d.items()):
            v2 = v[len(v)//4:] if len(v) > 8 else v
            lines.append(f"PMC {k:75s} {c:24s} n={len(v)} mean={sum(v2)/len(v2):.6g}")
txt = '\n'.join(lines)
print(txt)
open(os.path.join(out, 'summary.txt'), 'w').write(txt + '\n')
