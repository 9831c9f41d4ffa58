"""Instruction mix of one kernel in a hipcc -S dump: python scripts/isa_mix.py file.s 'substring of mangled name'"""
import collections, re, sys
lines = open(sys.argv[1]).read().splitlines()
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and l.rstrip().split(';')[0].strip().endswith(':'))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('.end_amdhsa_kernel') or lines[i].strip().startswith('.section') and i > start + 5)
cnt = collections.Counter()
for l in lines[start + 1:end]:
    l = l.strip()
    if not l or l[0] in '.;/' or l.split(';')[0].strip().endswith(':'):
        continue
    cnt[l.split()[0]] += 1
g = collections.Counter()
for op, c in cnt.items():
    if op.startswith('v_') and 'f64' in op: g['valu_f64'] += c
    elif op.startswith('v_'): g['valu_other'] += c
    elif op.startswith('s_'): g['salu'] += c
    elif op.startswith('ds_'): g['lds'] += c
    elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): g['vmem'] += c
    else: g['other'] += c
print('total', sum(cnt.values()), dict(g))
for op, c in cnt.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
    print(f'{op:34s}{c}')
