"""Does stepping an ensemble as S part-ensembles on S streams beat one ensemble on one stream?

    python scripts/perf_two_streams.py [--config metric|c2|c3|c4] [--k K] [--steps N]

Kernels queued on one stream run one after the other: the tail of a launch (the last wavefronts of the grid) and the
launch gap leave part of the GPU idle, and in the segmented form the small reduce kernel sits between the two big
ones.  Part-ensembles on separate streams can fill those holes with each other's work."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402

from bench import CONFIGS  # noqa: E402
from tango_amd.ensemble import OneFieldEnsemble, ThreeFieldEnsemble, shestakov_ensemble, threefield_ensemble  # noqa: E402


def build(fields, nb, n, a_prof, dt, lo, hi, rows, dev):
    alloc = lambda nbytes: torch.empty(nbytes, dtype=torch.uint8, device=dev)
    if fields == 1:
        return OneFieldEnsemble(shestakov_ensemble(nb, n, lo=lo, hi=hi), dt=dt, a_turb=0.3, a_prof=a_prof, tol=1e-11, fixed_count=True,
                                keep_err=True, max_iter=rows, alloc=alloc)
    return ThreeFieldEnsemble(threefield_ensemble(nb, n, lo=lo, hi=hi), dt=dt, a_turb=0.3, a_prof=a_prof, tol=1e-9, fixed_count=True,
                              keep_err=True, max_iter=rows, alloc=alloc)


def timed(parts, streams, k, steps, main):
    def go(nsteps):
        fork = torch.cuda.Event()
        fork.record(main)
        for s in streams:
            s.wait_event(fork)
        for _ in range(nsteps // k):
            for E, s in zip(parts, streams):
                E.step(s.cuda_stream, iters=k)
        for s in streams:
            j = torch.cuda.Event()
            j.record(s)
            main.wait_event(j)
    go(max(k, 256 - 256 % k))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    go(steps)
    e1.record(main)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (steps // k * k)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='c4')
    ap.add_argument('--k', type=int, default=0)
    ap.add_argument('--steps', type=int, default=192)
    ap.add_argument('--parts', type=int, nargs='+', default=[1, 2, 4])
    a = ap.parse_args()
    fields, nb, n, a_prof, dt, bpp, kdef = CONFIGS[a.config]
    k = a.k or kdef
    if fields == 3:
        k = 1
    dev = torch.device('cuda', 0)
    main_stream = torch.cuda.current_stream()
    rows = 4096
    for S in a.parts:
        cuts = [nb * i // S for i in range(S + 1)]
        parts = [build(fields, nb, n, a_prof, dt, cuts[i], cuts[i + 1], rows, dev) for i in range(S)]
        streams = [main_stream] if S == 1 else [torch.cuda.Stream() for _ in range(S)]
        ms = timed(parts, streams, k, a.steps, main_stream)
        print(f'{a.config} k={k} parts={S}: {ms:.4f} ms per iteration, {nb * n * bpp / ms / 1e9 * 1e3 / 8000:.3f} of HBM peak', flush=True)
        del parts


if __name__ == '__main__':
    main()
