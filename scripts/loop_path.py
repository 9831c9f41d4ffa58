"""Basic blocks and branches of one kernel in a `hipcc -S --cuda-device-only` dump, with the instruction mix of each block:
the tool for finding what a hot loop executes along the path it actually takes.

    python scripts/loop_path.py file.s 'substring of the mangled kernel name' [first_line last_line]

Without a range: every label, every branch with its target (BACK marks a loop edge) and the block sizes.  With a range
(line numbers within the kernel, as printed): the instruction mix of that span."""
import collections
import re
import sys


def kernel(lines, key):
    start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and l.split(';')[0].strip().endswith(':'))
    end = next(i for i in range(start + 1, len(lines)) if '.end_amdhsa_kernel' in lines[i] or lines[i].strip().startswith('.section'))
    return lines[start:end]


def classify(op):
    if op.startswith('v_') and 'f64' in op and not op.startswith('v_cmp') and not op.startswith('v_cvt'):
        return 'valu_f64'
    if op.startswith('v_cmp'):
        return 'valu_cmp'
    if op.startswith('v_cndmask'):
        return 'valu_select'
    if op.startswith('v_mov') or op.startswith('v_accvgpr') or op.startswith('v_readlane') or op.startswith('v_readfirstlane') or op.startswith('v_writelane'):
        return 'valu_move'
    if op.startswith('v_'):
        return 'valu_int_other'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith('global_') or op.startswith('buffer_') or op.startswith('scratch_') or op.startswith('flat_'):
        return 'vmem'
    if op.startswith('s_waitcnt') or op.startswith('s_nop'):
        return 'wait_nop'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


def mix(span):
    ops, groups = collections.Counter(), collections.Counter()
    for l in span:
        l = l.strip()
        if not l or l[0] in '.;/' or l.split(';')[0].strip().endswith(':'):
            continue
        op = l.split()[0]
        ops[op] += 1
        groups[classify(op)] += 1
        if 'dpp' in l:
            groups['(of which dpp)'] += 1
    return ops, groups


def main():
    lines = open(sys.argv[1]).read().split('\n')
    body = kernel(lines, sys.argv[2])
    if len(sys.argv) >= 5:
        a, b = int(sys.argv[3]), int(sys.argv[4])
        ops, groups = mix(body[a:b + 1])
        print(dict(groups))
        for op, c in ops.most_common(45):
            print(f'{op:32s} {c}')
        return
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = i
    print(len(body), 'lines,', len(labels), 'labels')
    for i, l in enumerate(body):
        m = re.match(r'\s*(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)', l)
        if m:
            tgt = labels[m.group(2)]
            print(f'{i:6d} {m.group(1):18s} -> {tgt:6d} {"BACK" if tgt < i else ""}')


if __name__ == '__main__':
    main()
