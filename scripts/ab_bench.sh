#!/bin/bash
# A/B of two builds of the library on the same box: bench.py with each, alternating, printing value and the k=1 rate.
# usage: scripts/ab_bench.sh libA.so libB.so [bench args...]
a=$1; b=$2; shift 2
for rep in 1 2 3; do
  for lib in $a $b; do
    TANGO_HIP_LIB=$PWD/$lib python bench.py --no-cpu-baseline "$@" > /tmp/ab.json 2>/tmp/ab.err || { tail -3 /tmp/ab.err; exit 1; }
    python - "$lib" <<'PY'
import json, sys
d = json.load(open('/tmp/ab.json'))
k1 = d['roofline'].get('hbm', {}).get('kernel_ms')
print(f"{sys.argv[1]:45s} value {d['value']:.1f}  ms/step {d['ms_per_step']:.4f}  k1_ms {k1}")
PY
  done
done
