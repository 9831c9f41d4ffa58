#!/bin/bash
# rocprofv3 kernel trace of scripts/bench_configs.py (all single-GPU configurations)
set -e
out=$PWD/gpurun_out/prof_configs
mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 280 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 scripts/bench_configs.py > $out/trace.log 2>&1
python3 scripts/prof_summary.py $out
grep -h '"config"' $out/trace.log > $out/lines.json || true
