"""Summarise the rocprofv3 csv output of scripts/profile_bench.sh / profile_cmd.sh: per-kernel time (--stats pass) and
counter means per dispatch (one PMC pass per sub-directory).  Appends to <dir>/summary.txt and prints it."""
import collections
import csv
import glob
import os
import sys

out = sys.argv[1]
KEEP = ('picard', 'tridiag', 'seg_', 'pair_solve', 'block3', 'status_reduce', 'residual_reduce', 'stream_probe')


def find(sub, pat):
    return sorted(glob.glob(os.path.join(out, sub, '**', pat), recursive=True))


lines = []
for f in find('trace', '*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        lines.append(f"STATS {r.get('Name','')[:110]:110s} calls={r.get('Calls')} avg_ns={r.get('AverageNs')} total_ns={r.get('TotalDurationNs')} pct={r.get('Percentage')}")
# median duration per kernel from the trace itself: the average above includes the launches that run while the clock is
# still coming up (the first few of a run take up to 25 % longer), the median is the steady launch
dur = collections.defaultdict(list)
for f in find('trace', '*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if any(w in r['Kernel_Name'] for w in KEEP):
            dur[r['Kernel_Name'][:110]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in dur.items():
    v.sort()
    lines.append(f"MEDIAN {k:110s} calls={len(v)} median_ns={v[len(v) // 2]} min_ns={v[0]}")
for sub in ('sq', 'sqf', 'fetch', 'write'):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in find(sub, '*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:95]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        if not any(w in k for w in KEEP):
            continue
        for c, v in sorted(d.items()):
            v2 = v[len(v) // 4:] if len(v) > 8 else v      # skip the warm-up dispatches of a long series
            lines.append(f"PMC {k:95s} {c:26s} n={len(v)} mean={sum(v2)/len(v2):.6g}")
txt = '\n'.join(lines)
print(txt)
open(os.path.join(out, 'summary.txt'), 'a').write(txt + '\n')
