#!/bin/bash
# rocprofv3 passes over bench.py itself: kernel trace with statistics, then one counter pass per group (FETCH_SIZE,
# WRITE_SIZE, eight SQ counters, the four fp64 instruction counters), each in a process of its own.
# usage: scripts/profile_bench.sh <tag> <bench.py arguments...>     e.g.  scripts/profile_bench.sh metric_k64 --iters-per-launch 64
set -e
tag=$1; shift
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
args="bench.py --no-cpu-baseline $*"
timeout -s KILL 170 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $args > $out/trace.log 2>&1
timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $args > $out/fetch.log 2>&1
timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $args > $out/write.log 2>&1
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/sq -- python3 $args > $out/sq.log 2>&1
timeout -s KILL 170 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $out/sqf -- python3 $args > $out/sqf.log 2>&1
echo "# rocprofv3 over: python3 $args" > $out/summary.txt
python3 scripts/prof_summary.py $out >> /dev/null
grep -h '"metric"' $out/trace.log | tail -1 > $out/bench_line.json || true
