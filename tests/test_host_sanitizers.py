"""Host code under sanitizers (no GPU needed): the C restatement stepped under AddressSanitizer + UBSan, and every entry
point of a host-ASan build of libtango_hip.so called with arguments its validation must refuse."""
import importlib.util
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_oracle_under_address_and_ub_sanitizers():
    r = subprocess.run(['make', '-s', '-C', os.path.join(ROOT, 'oracle'), 'asan'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=280)
    assert r.returncode == 0 and 'oracle sanitizer check ok' in r.stdout, r.stdout[-3000:]


def test_library_argument_checks_under_address_sanitizer():
    spec = importlib.util.spec_from_file_location('tango_amd_build', os.path.join(ROOT, 'tango_amd', 'build.py'))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    out = b.host_check(b.build(asan_host=True))        # rebuilt only when a source is newer than the library
    assert 'host_check ok' in out, out[-3000:]
