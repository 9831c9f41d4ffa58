#!/bin/bash
# rocprofv3 passes (kernel trace with statistics, FETCH_SIZE, WRITE_SIZE, eight SQ counters) over any python script.
# usage: scripts/profile_cmd.sh <tag> <script.py> [arguments...]
set -e
tag=$1; shift
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 170 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $* > $out/trace.log 2>&1
timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $* > $out/fetch.log 2>&1
timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $* > $out/write.log 2>&1
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/sq -- python3 $* > $out/sq.log 2>&1
echo "# rocprofv3 over: python3 $*" > $out/summary.txt
python3 scripts/prof_summary.py $out >> /dev/null
grep -h '^{' $out/trace.log >> $out/summary.txt || true
