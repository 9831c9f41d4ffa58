"""Instruction mix of one kernel in a `hipcc -S --cuda-device-only` dump.

    python scripts/isa_mix.py file.s 'substring of the mangled name' [--loop] [--json out.json] [--top N]

--loop restricts the count to the body of the kernel's largest loop (the span between a label and the last
backward branch to it), e.g. one Picard iteration of the multi-iteration kernel.  The fp64 flop count takes an
fma as two operations and every other fp64 arithmetic instruction (add, mul, rcp, sqrt, min/max, div_* helpers,
comparisons excluded) as one."""
import collections
import json
import re
import sys


def kernel_lines(lines, key):
    start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and l.split(';')[0].strip().endswith(':'))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('.end_amdhsa_kernel') or
               (lines[i].strip().startswith('.section') and i > start + 5))
    return lines[start + 1:end]


def largest_loop(body):
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r'^(\.?[A-Za-z_][\w$.]*):', l.strip())
        if m:
            labels[m.group(1)] = i
    best = (0, 0)
    for i, l in enumerate(body):
        m = re.match(r'\s*s_cbranch_\w+\s+(\S+)|\s*s_branch\s+(\S+)', l)
        if m:
            tgt = m.group(1) or m.group(2)
            j = labels.get(tgt)
            if j is not None and j < i and i - j > best[1] - best[0]:
                best = (j, i)
    return body[best[0]:best[1] + 1]


def mix(body):
    cnt = collections.Counter()
    for l in body:
        l = l.strip()
        if not l or l[0] in '.;/' or l.split(';')[0].strip().endswith(':'):
            continue
        cnt[l.split()[0]] += 1
    g = collections.Counter()
    flops = 0
    for op, c in cnt.items():
        if op.startswith('v_') and 'f64' in op:
            g['valu_f64'] += c
            if op.startswith('v_cmp') or op.startswith('v_cvt') or op.startswith('v_div_scale'):
                pass
            elif op.startswith('v_fma') or op.startswith('v_div_fmas'):
                flops += 2 * c
            else:
                flops += c
        elif op.startswith('v_'):
            g['valu_other'] += c
        elif op.startswith('s_waitcnt'):
            g['s_waitcnt'] += c
        elif op.startswith('s_'):
            g['salu'] += c
        elif op.startswith('ds_'):
            g['lds'] += c
        elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
            g['vmem'] += c
        else:
            g['other'] += c
    g['ds_bpermute_b32'] = cnt.get('ds_bpermute_b32', 0)
    g['dpp_moves'] = sum(c for op, c in cnt.items() if op.endswith('_dpp'))
    g['v_rcp_f64'] = cnt.get('v_rcp_f64_e32', 0) + cnt.get('v_rcp_f64_e64', 0)
    g['fp64_flops_per_lane'] = flops
    g['total'] = sum(cnt.values())
    return cnt, dict(g)


if __name__ == '__main__':
    args = sys.argv[1:]
    lines = open(args[0]).read().splitlines()
    body = kernel_lines(lines, args[1])
    if '--loop' in args:
        body = largest_loop(body)
    cnt, g = mix(body)
    print(json.dumps(g))
    if '--json' in args:
        json.dump(g, open(args[args.index('--json') + 1], 'w'), indent=1)
    top = int(args[args.index('--top') + 1]) if '--top' in args else 30
    for op, c in cnt.most_common(top):
        print(f'{op:34s}{c}')
