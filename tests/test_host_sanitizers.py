"""The library's host side under AddressSanitizer + UBSan (sanitizers for host-side tests).  The
device code is cross-compiled as usual and never runs here; csrc/host_check.cpp walks every entry point that returns
before its first launch.  Set NIF_SKIP_ASAN=1 to leave the 30-second build out of a quick run."""
import os
import shutil
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                    'neuron_interpolation_model_fusion_amd', 'csrc')


@pytest.mark.skipif(os.environ.get('NIF_SKIP_ASAN') == '1', reason='NIF_SKIP_ASAN=1')
@pytest.mark.skipif(shutil.which('make') is None or not os.path.exists('/opt/rocm/bin/hipcc'), reason='no toolchain')
def test_host_side_is_clean_under_asan_and_ubsan():
    r = subprocess.run(['make', '-C', CSRC, '-j8', 'asan-check'], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'host_check: ok' in r.stdout
    assert 'ERROR: AddressSanitizer' not in r.stderr and 'runtime error' not in r.stderr
