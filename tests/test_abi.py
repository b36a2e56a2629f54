"""not-gpu: libnif.so loads, exports every symbol include/nif.h declares, and rejects bad arguments
without touching a device (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'nif.h')


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(nif_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_entry_points():
    names = declared_symbols()
    for must in ['nif_xty', 'nif_cost_dist', 'nif_cost_hf', 'nif_lap_solve_f32', 'nif_refit_gather', 'nif_refit_pairs',
                 'nif_kmeans_lloyd_step', 'nif_kmeans_farthest_init', 'nif_kmeans_repair_kf', 'nif_kmeans_targets']:
        assert must in names


def test_library_exports_every_declared_symbol():
    from neuron_interpolation_model_fusion_amd import _lib
    raw = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in declared_symbols() if not hasattr(raw, n)]
    assert not missing, f'declared in include/nif.h but not exported by libnif.so: {missing}'
    bound = set(_lib.SIGNATURES)
    assert set(declared_symbols()) == bound, f'ctypes table out of sync with the header: {set(declared_symbols()) ^ bound}'


def test_version_errors_and_argument_checks():
    from neuron_interpolation_model_fusion_amd import _lib
    lib = _lib.lib
    assert lib.nif_version() >= 100
    assert lib.nif_error_string(0) == b'ok'
    assert b'workspace' in lib.nif_error_string(-2)
    # sizes / null pointers are rejected before any launch
    assert lib.nif_xty(None, 4, None, 4, 8, 4, 4, 0, 0, 0, None, 4, None, None, None, 0, None) == -1
    assert lib.nif_lap_solve_f32(None, 4, 4, 0, None, None, None, None, None, 0, None) == -1
    assert lib.nif_xty_workspace_bytes(325000, 384, 384, 0, 0) > 0
    assert lib.nif_lap_workspace_bytes(384) > 0
    assert lib.nif_lap_workspace_bytes(100000) == 0
    assert lib.nif_kmeans_workspace_bytes(768, 384) > 768 * 384 * 4


def test_ops_refuse_cpu_tensors():
    import torch
    from neuron_interpolation_model_fusion_amd import ops, _lib
    with pytest.raises(_lib.NifError):
        ops.xty(torch.randn(8, 4), torch.randn(8, 4))
    with pytest.raises(_lib.NifError):
        ops.lap_solve(torch.rand(4, 4))
