"""The C-ABI library loads on a machine without a GPU and exports every symbol include/tango_hip.h declares;
compute entry points fail loudly (no CPU fallback) when no GPU is present."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, 'include', 'tango_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(tg_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from tango_amd import _hip as H
    names = header_functions()
    assert len(names) >= 25
    for name in names:
        assert hasattr(H.lib, name), f'{name} declared in include/tango_hip.h but not exported'
        assert name in H.PROTOTYPES, f'{name} has no ctypes prototype'
    assert sorted(H.PROTOTYPES) == names
    assert H.lib.tg_version() >= 100


def test_struct_sizes_match_header():
    """ctypes mirrors of the parameter structs have the size the C compiler gives them."""
    import subprocess
    import tempfile
    from tango_amd import _hip as H
    src = '#include <stdio.h>\n#include "tango_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu", sizeof(tg_hcoeffs), ' \
          'sizeof(tg_state_1f), sizeof(tg_params_1f), sizeof(tg_state_3f), sizeof(tg_params_3f));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, 's.c')
        open(c, 'w').write(src)
        exe = os.path.join(d, 's')
        subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), c, '-o', exe])
        sizes = [int(v) for v in subprocess.check_output([exe]).split()]
    import ctypes as C
    assert sizes == [C.sizeof(H.HCoeffsC), C.sizeof(H.State1F), C.sizeof(H.Params1F), C.sizeof(H.State3F),
                     C.sizeof(H.Params3F)]


def test_no_cpu_fallback():
    import tango_amd as tango
    if tango.device_count() > 0:
        pytest.skip('a GPU is present')
    with pytest.raises(tango.HipError):
        tango.tridiag.solve(np.zeros(4), np.ones(4), np.zeros(4), np.ones(4))
    with pytest.raises(tango.HipError):
        tango.derivatives.dx_centered_difference(np.arange(5.0), 0.1)


def test_product_package_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under tango_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'tango_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                text = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in text.replace('oracle is test', ''), f'{f} mentions the oracle'
