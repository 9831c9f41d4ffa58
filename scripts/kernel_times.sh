#!/bin/bash
# rocprofv3 kernel trace of one bench.py run, per-kernel average durations printed.
# usage: scripts/kernel_times.sh <tag> <bench.py arguments...>
tag=$1; shift
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 170 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline "$@" > $out/trace.log 2>&1
python3 - <<P
import csv,glob
for f in glob.glob("$out/trace/**/*kernel_stats.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"])>0.5: print(r["Name"].replace("(anonymous namespace)::","")[:100], r["Calls"], "%.1f us"%(float(r["AverageNs"])/1e3), r["Percentage"])
P
grep -h '^{' $out/trace.log | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('bench under rocprof: ms/step %.4f'%d['ms_per_step'])"
