#!/bin/bash
# The round's standing GPU check: the whole GPU test-suite, then bench.py with the driver's arguments and with its defaults.
# usage (through gpurun): bash scripts/gpu_check.sh <tag>
tag=${1:-check}
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 170 --timeout-method thread --durations=8 > gpurun_out/gputest_$tag.log 2>&1
rc=$?
tail -15 gpurun_out/gputest_$tag.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 170 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_$tag.log 2>gpurun_out/bench_driver_$tag.err || { tail -5 gpurun_out/bench_driver_$tag.err; exit 1; }
timeout -k 10 170 python bench.py --no-cpu-baseline > gpurun_out/bench_default_$tag.log 2>/dev/null || exit 1
python - <<P
import json
for f in ('gpurun_out/bench_driver_$tag.log','gpurun_out/bench_default_$tag.log'):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d['roofline']
    print(f, 'value %.1f ms/step %.4f bound %s frac %.3f issue %s | k1 ms %.4f frac %.3f | rounds %s'%(d['value'],d['ms_per_step'],r['bound'],r['frac'] or 0,r.get('valu_issue_frac'),r.get('hbm',{}).get('kernel_ms',0),d['hbm_frac_headline'],d['config']['launch_shape']['all_reduce_rounds_in_timed_region']))
P
