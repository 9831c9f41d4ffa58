#!/bin/bash
# bench.py over the configurations, one summary line each.  usage: scripts/bench_configs.sh [extra bench args]
run() { timeout -k 10 200 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; k1=r.get('hbm',{})
        print('%-60s ms/step %.4f value %8.1f bound %-9s frac %.3f | k1 ms %s frac %s | issue %s invalid %d'%(' '.join(sys.argv[1:]), d['ms_per_step'],d['value'],r['bound'],r['frac'] or 0,
              ('%.4f'%k1['kernel_ms']) if k1 else '-', ('%.3f'%k1['frac']) if k1 else '-', ('%.3f'%r['valu_issue_frac']) if r.get('valu_issue_frac') else '-', d['invalid_profiles']))
" "$@" || exit 1; }
run --steps 256 --warmup 64 "$@"
run --steps 20 --warmup 5 "$@"
run --steps 256 --warmup 64 --theta convective "$@"
run --steps 256 --warmup 64 --iters-per-launch 1 --theta convective "$@"
run --config c2 --steps 256 --warmup 64 "$@"
run --config c3 --steps 192 --warmup 64 "$@"
run --config c4 --steps 192 --warmup 64 "$@"
run --config c5 --steps 128 --warmup 64 "$@"
