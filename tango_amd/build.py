"""Build libtango_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python tango_amd/build.py [--report] [--asan-host] [--host-check]   (run as a script: importing the package needs the built library)

--asan-host builds lib/libtango_hip_asan.so with AddressSanitizer on the host code (the device code is left alone);
--host-check builds that and runs csrc/host_check.cpp against it (every entry point's argument checks, no GPU needed).

The library is written in-tree to tango_amd/lib/ so that it travels with the source tree.  The translation units
are compiled side by side and linked afterwards.

Floating-point contraction.  hipcc's default for device code is -ffp-contract=fast-honor-pragmas and it is given
here explicitly: the compiler may fuse a*b+c into one fma anywhere except where a `#pragma clang fp contract(off)`
says otherwise.  Where the rounding of a single operation matters the kernels say so themselves: the division
sequences and the slack-carrying elimination call __builtin_fma where they need one rounding, and the
batch-fastest Thomas kernel, which reproduces LAPACK dgtsv operation by operation, switches contraction off for
its body.  `python tango_amd/build.py --contract-off` builds lib/libtango_hip_nocontract.so with -ffp-contract=off
for comparison (TANGO_HIP_LIB selects the library a process loads); measured on MI355X with bench.py, the two
builds run within 1 % of each other (5227 against 5178 iterations/s at 64 iterations per launch, 2631 against 2628 at one).
"""
import concurrent.futures
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, 'csrc', f) for f in ('tango_hip.hip', 'tango_block.hip', 'tango_seg.hip')]
OUT = os.path.join(HERE, 'lib', 'libtango_hip.so')
OBJ = os.path.join(HERE, 'lib', 'obj')
FLAGS = ['-O3', '--offload-arch=gfx950', '-std=c++17', '-fPIC', '-Wall', '-Wno-format-security',
         '-Wno-unused-function', '-ffp-contract=fast-honor-pragmas']
# host-side AddressSanitizer (the device code is left alone: GPU sanitizers are not available on this pool)
ASAN = ['-Xarch_host', '-fsanitize=address', '-Xarch_host', '-fno-omit-frame-pointer', '-g']


def stale(out=OUT):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = [os.path.join(HERE, 'csrc', f) for f in os.listdir(os.path.join(HERE, 'csrc'))]
    deps.append(os.path.join(HERE, '..', 'include', 'tango_hip.h'))
    deps.append(os.path.abspath(__file__))
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
    return r.stdout


def build(report=False, force=False, asan_host=False, extra_flags=(), out=None):
    out = out or (OUT.replace('.so', '_asan.so') if asan_host else OUT)
    if not (force or report or stale(out)):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    objdir = OBJ + ('_asan' if asan_host else '')
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    flags = FLAGS + (ASAN if asan_host else []) + list(extra_flags)
    if report:
        flags = flags + ['-Rpass-analysis=kernel-resource-usage']
    srcs = [s for s in SRC if os.path.exists(s)]
    objs = [os.path.join(objdir, os.path.basename(s).replace('.hip', '.o')) for s in srcs]
    with concurrent.futures.ThreadPoolExecutor(max_workers=len(srcs)) as ex:
        logs = list(ex.map(lambda so: _run([hipcc] + flags + ['-c', so[0], '-o', so[1]]), zip(srcs, objs)))
    link = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs
    if asan_host:
        link += ['-fsanitize=address']
    _run(link)
    if report:
        name = None
        row = {}
        text = '\n'.join(logs)
        for line in text.splitlines():
            m = re.search(r'Function Name: (\S+)', line)
            if m:
                name = subprocess.run(['c++filt', m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()
                name = re.sub(r'\(anonymous namespace\)::', '', name).split('(')[0]
                row = {}
            for key in ('VGPRs', 'VGPRs Spill', 'SGPRs', 'Occupancy [waves/SIMD]', 'LDS Size [bytes/block]',
                        'ScratchSize [bytes/lane]'):
                m = re.search(r'remark:\s+' + re.escape(key) + r': (\d+)', line)
                if m:
                    row[key] = int(m.group(1))
            if 'LDS Size' in line and name:
                print(f"{name:70s} vgpr={row.get('VGPRs')} spill={row.get('VGPRs Spill')} "
                      f"scratch={row.get('ScratchSize [bytes/lane]')} sgpr={row.get('SGPRs')} "
                      f"occ={row.get('Occupancy [waves/SIMD]')}")
        for line in text.splitlines():
            if 'warning' in line or 'error' in line:
                print(line)
    return out


def host_check(lib):
    """Build and run csrc/host_check.cpp against a (sanitizer) build of the library: every entry point's argument
    validation, no GPU needed.  Returns the program's output; raises if it fails."""
    exe = os.path.join(os.path.dirname(lib), 'host_check')
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    _run([hipcc, '-x', 'c++', '-std=c++17', '-O1', '-g', '-fsanitize=address', '-fno-omit-frame-pointer',
          os.path.join(HERE, 'csrc', 'host_check.cpp'), '-x', 'none', lib, '-Wl,-rpath,' + os.path.dirname(lib), '-o', exe])
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=0:protect_shadow_gap=0:verify_asan_link_order=0')
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
    if r.returncode != 0:
        raise RuntimeError('host_check failed:\n' + r.stdout)
    return r.stdout


if __name__ == '__main__':
    if '--host-check' in sys.argv:
        print(host_check(build(asan_host=True)))
        sys.exit(0)
    print(build(report='--report' in sys.argv, force=True, asan_host='--asan-host' in sys.argv,
                extra_flags=['-ffp-contract=off'] if '--contract-off' in sys.argv else (),
                out=OUT.replace('.so', '_nocontract.so') if '--contract-off' in sys.argv else None))
