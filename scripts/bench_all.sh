run() { echo "== $*"; timeout -k 10 200 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; k1=r.get('hbm',{})
        print('ms/step %.4f value %.1f bound %s frac %s k1_ms %s k1_frac %s issue %s'%(d['ms_per_step'],d['value'],r['bound'],r['frac'],k1.get('kernel_ms'),k1.get('frac'),r.get('valu_issue_frac')))
" || exit 1; }
run --steps 256 --warmup 64
run --steps 256 --warmup 64 --iters-per-launch 1
run --config c2 --steps 256 --warmup 64
run --config c2 --steps 256 --warmup 64 --iters-per-launch 8
run --config c3 --steps 192 --warmup 64
run --config c4 --steps 192 --warmup 64
run --config c4 --steps 192 --warmup 64 --iters-per-launch 64
run --config c4 --steps 96 --warmup 32 --iters-per-launch 1
run --profiles-per-gpu 32768 --n 1024 --steps 96 --warmup 32 --iters-per-launch 1
run --profiles-per-gpu 16384 --n 2048 --steps 96 --warmup 32 --iters-per-launch 1
run --profiles-per-gpu 8192 --n 4096 --steps 96 --warmup 32 --iters-per-launch 1
run --profiles-per-gpu 2048 --n 16384 --steps 96 --warmup 32 --iters-per-launch 16
