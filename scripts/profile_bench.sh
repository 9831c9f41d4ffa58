#!/bin/bash
# rocprofv3 passes over bench.py itself (kernel trace, then separate PMC passes), for one value of
# --iters-per-launch.  usage: scripts/profile_bench.sh <iters-per-launch> [steps]
set -e
k=$1; steps=${2:-96}
out=$PWD/gpurun_out/prof_bench_k$k
mkdir -p $out
export TMPDIR=/tmp
warm=$(( k > 16 ? k : 16 ))     # warm-up launches of the same shape as the timed ones
args="bench.py --steps $steps --warmup $warm --iters-per-launch $k --check-every $k --no-cpu-baseline"
timeout -s KILL 170 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $args > $out/trace.log 2>&1
timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $args > $out/fetch.log 2>&1
timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $args > $out/write.log 2>&1
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/sq -- python3 $args > $out/sq.log 2>&1
python3 scripts/prof_summary.py $out
grep -h '"metric"' $out/trace.log | tail -1 > $out/bench_line.json || true
