"""Order of global/scratch loads, stores, barriers and `s_waitcnt vmcnt` waits of one kernel in a `hipcc -S
--cuda-device-only` dump, runs of the same instruction folded:

    python scripts/mem_order.py file.s 'substring of the mangled name' [--branches]

On gfx950 loads and stores share one counter and leave it in order, so a `vmcnt` wait that follows stores also waits
for those stores to reach memory.  A kernel is fine when no `s_waitcnt vmcnt` (and no scratch reload) shows up after
its first `global_store`."""
import re
import sys


def main():
    lines = open(sys.argv[1]).read().splitlines()
    key = sys.argv[2]
    start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l)
    end = next(i for i in range(start, len(lines)) if '.end_amdhsa_kernel' in lines[i])
    seq = []
    for i in range(start, end):
        l = lines[i].strip().split(';')[0].strip()
        if re.match(r'(global_load|global_store|scratch_|s_barrier|s_endpgm)', l) or ('s_waitcnt' in l and 'vmcnt' in l):
            seq.append(l.split()[0] if not l.startswith('s_waitcnt') else l)
        elif '--branches' in sys.argv and l.startswith('s_cbranch'):
            seq.append('branch')
    out, prev, cnt = [], None, 0
    for k in seq:
        if k == prev:
            cnt += 1
        else:
            if prev:
                out.append(f'{prev} x{cnt}')
            prev, cnt = k, 1
    out.append(f'{prev} x{cnt}')
    stored = False
    for o in out:
        late = stored and ('vmcnt' in o or o.startswith(('scratch_load', 'global_load')))
        stored = stored or o.startswith('global_store')
        print(('!! ' if late else '   ') + o)


if __name__ == '__main__':
    main()
