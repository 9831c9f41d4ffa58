"""Build libtango_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python tango_amd/build.py [--report]      (run as a script: importing the package needs the built library)

The library is written in-tree to tango_amd/lib/ so that it travels with the source tree."""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, 'csrc', f) for f in ('tango_hip.hip', 'tango_block.hip', 'tango_seg.hip')]
OUT = os.path.join(HERE, 'lib', 'libtango_hip.so')
FLAGS = ['-O3', '--offload-arch=gfx950', '-std=c++17', '-fPIC', '-shared', '-Wall', '-Wno-format-security',
         '-Wno-unused-function']


def stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(HERE, 'csrc', f) for f in os.listdir(os.path.join(HERE, 'csrc'))]
    deps.append(os.path.join(HERE, '..', 'include', 'tango_hip.h'))
    return any(os.path.getmtime(d) > t for d in deps)


def build(report=False, force=False):
    if not (force or report or stale()):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc] + FLAGS + ['-o', OUT] + [s for s in SRC if os.path.exists(s)]
    if report:
        cmd.append('-Rpass-analysis=kernel-resource-usage')
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError('hipcc failed')
    if report:
        name = None
        row = {}
        for line in r.stdout.splitlines():
            m = re.search(r'Function Name: (\S+)', line)
            if m:
                name = subprocess.run(['c++filt', m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()
                name = re.sub(r'\(anonymous namespace\)::', '', name).split('(')[0]
                row = {}
            for key in ('VGPRs', 'VGPRs Spill', 'SGPRs', 'Occupancy [waves/SIMD]', 'LDS Size [bytes/block]'):
                m = re.search(r'remark:\s+' + re.escape(key) + r': (\d+)', line)
                if m:
                    row[key] = int(m.group(1))
            if 'LDS Size' in line and name:
                print(f"{name:70s} vgpr={row.get('VGPRs')} spill={row.get('VGPRs Spill')} "
                      f"sgpr={row.get('SGPRs')} occ={row.get('Occupancy [waves/SIMD]')}")
        for line in r.stdout.splitlines():
            if 'warning' in line or 'error' in line:
                print(line)
    return OUT


if __name__ == '__main__':
    print(build(report='--report' in sys.argv, force=True))
