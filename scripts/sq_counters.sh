#!/bin/bash
# Two SQ counter passes over one bench.py run (8 counters each), per-kernel means printed per wavefront.
# usage: scripts/sq_counters.sh <tag> <kernel substring> <bench.py arguments...>
tag=$1; key=$2; shift 2
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $out/sqa -- python3 bench.py --no-cpu-baseline "$@" > $out/sqa.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR --output-format csv -d $out/sqb -- python3 bench.py --no-cpu-baseline "$@" > $out/sqb.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_INSTS_BRANCH SQ_INSTS_SMEM --output-format csv -d $out/sqc -- python3 bench.py --no-cpu-baseline "$@" > $out/sqc.log 2>&1 || exit 1
python3 - <<P
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$out/sq[abc]/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "$key" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
w=acc["SQ_WAVES"]; waves=sum(w)/len(w)
print("kernel ~ $key   waves per dispatch", waves)
for c,v in sorted(acc.items()):
    v=v[len(v)//4:]
    print("  %-28s %14.4g   per wave %10.1f"%(c,sum(v)/len(v),sum(v)/len(v)/waves))
P
