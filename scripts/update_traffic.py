"""Turn the summaries written by scripts/profile_bench.sh (gpurun_out/prof_<tag>/summary.txt) into the numbers
bench.py quotes beside its live timings (profiles/traffic.json) and copy the summaries to profiles/<round>_<tag>_summary.txt.

    python scripts/update_traffic.py r02 metric_k64:metric_65536x512:64 metric_k1:metric_65536x512:1 c2_k64:c2_65536x256:64 c3:c3_16384x512:1 c4_k1:c4_4096x8192:1 c4_k16:c4_4096x8192:16

Each argument is <profile tag>:<bench key prefix>:<iterations per launch of the timed kernel>.  HBM bytes per launch =
(2 x FETCH_SIZE + WRITE_SIZE) KiB summed over the kernels of one step: on gfx950 FETCH_SIZE counts half the bytes of a
wide coalesced read stream, WRITE_SIZE counts the bytes of 16-byte-per-lane stores exactly."""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEP_KERNELS = ('picard_1f_kernel', 'picard_3f_kernel', 'picard_seg_kernel', 'picard_seg_reduce_kernel', 'seg_halo_kernel')


def parse(path):
    pmc = {}
    for line in open(path):
        m = re.match(r'PMC (.+?)\s+(\w+)\s+n=(\d+) mean=(\S+)', line)
        if m:
            pmc.setdefault(m.group(1).strip(), {})[m.group(2)] = (int(m.group(3)), float(m.group(4)))
    return pmc


def main():
    rnd = sys.argv[1]
    tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
    tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
    tj = {k: v for k, v in tj.items() if not k.startswith('picard_1f_')}      # keys of an older layout
    for spec in sys.argv[2:]:
        tag, key, k = spec.split(':')
        k = int(k)
        src = os.path.join(ROOT, 'gpurun_out', f'prof_{tag}', 'summary.txt')
        pmc = parse(src)
        # kernels of the timed step: for k > 1 the multi-iteration form (last template flag true), else the others
        total = 0.0
        flops = valu = waves = None
        for name, c in pmc.items():
            if not any(name.startswith(s) or (' ' + s) in name for s in STEP_KERNELS):
                continue
            multi = bool(re.search(r'picard_1f_kernel<.*, true>\(', name))
            if ('picard_1f_kernel' in name) and (multi != (k > 1)):
                continue
            # the segmented form's k iterations per call are a chain: k reduce kernels, k - 1 combined kernels, one of the rest
            times = 1
            if k > 1 and 'picard_seg_kernel<2' in name:
                times = k - 1
            elif k > 1 and 'picard_seg_reduce_kernel' in name:
                times = k
            elif k == 1 and 'picard_seg_kernel<2' in name:
                continue
            if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
                total += times * (2 * c['FETCH_SIZE'][1] + c['WRITE_SIZE'][1]) * 1024
            if 'SQ_INSTS_VALU_FMA_F64' in c and ('picard_1f_kernel' in name):
                f = 2 * c['SQ_INSTS_VALU_FMA_F64'][1] + c['SQ_INSTS_VALU_MUL_F64'][1] + c['SQ_INSTS_VALU_ADD_F64'][1] + c['SQ_INSTS_VALU_TRANS_F64'][1]
                waves = c.get('SQ_WAVES', pmc[name].get('SQ_WAVES', (0, 0)))[1]
                flops, valu = f, c.get('SQ_INSTS_VALU', (0, 0))[1]
        tj[f'{key}_k{k}_hbm_bytes_per_launch'] = int(total)
        if k > 1 and flops and waves:
            tj[f'{key}_fp64_flops_per_wave_iteration'] = round(flops / waves / k, 2)
            tj[f'{key}_valu_insts_per_wave_iteration'] = round(valu / waves / k, 2)
        shutil.copy(src, os.path.join(ROOT, 'profiles', f'{rnd}_{tag}_summary.txt'))
        line = os.path.join(ROOT, 'gpurun_out', f'prof_{tag}', 'bench_line.json')
        if os.path.exists(line):
            with open(os.path.join(ROOT, 'profiles', f'{rnd}_{tag}_summary.txt'), 'a') as f:
                f.write("# bench.py's own line under the trace pass:\n" + open(line).read())
    tj['_note'] = ('HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB over the kernels of one step, fp64 operations = '
                   '2 FMA + MUL + ADD + TRANS wave-instructions (x 64 lanes when turned into flops), from the rocprofv3 counter '
                   'passes summarised in profiles/*_summary.txt (scripts/profile_bench.sh, scripts/update_traffic.py)')
    json.dump(tj, open(tpath, 'w'), indent=1, sort_keys=True)
    print(json.dumps(tj, indent=1, sort_keys=True))


if __name__ == '__main__':
    main()
