"""-m gpu: nif_lap_solve_f32_static -- the entry point that allocates nothing (static device scratch inside
libnif.so, include/nif.h) -- against nif_lap_solve_f32 with a caller workspace, straight through the C ABI.
Two calls in a row on one stream, the second on a smaller problem (the scratch keeps the first call's contents),
both solver modes, optional outputs present and absent, and the size limit."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _solve(lib, C, flags, static, want_duals):
    n = C.shape[0]
    dev = C.device
    col = torch.full((n,), -7, dtype=torch.int32, device=dev)
    row = torch.full((n,), -7, dtype=torch.int32, device=dev) if want_duals else None
    u = torch.zeros(n, dtype=torch.float64, device=dev) if want_duals else None
    v = torch.zeros(n, dtype=torch.float64, device=dev) if want_duals else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    if static:
        rc = lib.nif_lap_solve_f32_static(_p(C), C.stride(0), n, flags, _p(col), _p(row), _p(u), _p(v), st)
    else:
        nbytes = lib.nif_lap_workspace_bytes(n)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        rc = lib.nif_lap_solve_f32(_p(C), C.stride(0), n, flags, _p(col), _p(row), _p(u), _p(v), _p(ws), nbytes, st)
    assert rc == 0, rc
    torch.cuda.synchronize()
    return col, row, u, v


def _cost(n, seed, dev, ties=False):
    g = torch.Generator().manual_seed(seed)
    C = torch.rand(n, n, generator=g)
    if ties:  # duplicated columns: the scipy-order replay has to break exact ties the same way in both entry points
        C[:, n // 2:] = C[:, : n - n // 2]
    return C.to(dev).contiguous()


def test_static_scratch_entry_point_equals_the_workspace_entry_point():
    from neuron_interpolation_model_fusion_amd import _lib
    lib = _lib.lib
    dev = torch.device('cuda', 0)
    with torch.cuda.device(dev):
        for flags in (0, _lib.NIF_LAP_SCIPY_ORDER):
            # first a large problem, then a smaller one right after it on the same stream
            for n, seed, ties in ((1500, 1, False), (333, 2, True), (2048, 3, False), (64, 4, False), (1, 5, False)):
                C = _cost(n, seed, dev, ties)
                for want_duals in (True, False):
                    ref = _solve(lib, C, flags, False, want_duals)
                    got = _solve(lib, C, flags, True, want_duals)
                    assert torch.equal(got[0], ref[0]), (flags, n, want_duals)
                    assert sorted(got[0].tolist()) == list(range(n))
                    if want_duals:
                        assert torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2]) and torch.equal(got[3], ref[3])
        # two back-to-back launches without a host sync in between (stream order protects the shared scratch)
        C1, C2 = _cost(900, 6, dev), _cost(200, 7, dev)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        c1 = torch.empty(900, dtype=torch.int32, device=dev)
        c2 = torch.empty(200, dtype=torch.int32, device=dev)
        assert lib.nif_lap_solve_f32_static(_p(C1), 900, 900, 0, _p(c1), None, None, None, st) == 0
        assert lib.nif_lap_solve_f32_static(_p(C2), 200, 200, 0, _p(c2), None, None, None, st) == 0
        torch.cuda.synchronize()
        assert torch.equal(c1, _solve(lib, C1, 0, False, False)[0])
        assert torch.equal(c2, _solve(lib, C2, 0, False, False)[0])


def test_static_scratch_entry_point_allocates_no_device_memory():
    """What the entry point is for: after a warm-up call (code objects loaded, the scratch symbol resolved) further
    calls leave the device's free memory and torch's allocation counter exactly where they were, in both solver modes
    and with a smaller problem after a larger one."""
    from neuron_interpolation_model_fusion_amd import _lib
    lib = _lib.lib
    dev = torch.device('cuda', 0)
    with torch.cuda.device(dev):
        problems = [(_cost(n, 10 + n, dev), torch.empty(n, dtype=torch.int32, device=dev)) for n in (2048, 700, 64)]
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

        def run_all():
            for flags in (0, _lib.NIF_LAP_SCIPY_ORDER):
                for C, col in problems:
                    n = C.shape[0]
                    assert lib.nif_lap_solve_f32_static(_p(C), n, n, flags, _p(col), None, None, None, st) == 0
            torch.cuda.synchronize()

        run_all()  # warm-up
        free0 = torch.cuda.mem_get_info(dev)[0]
        count0 = torch.cuda.memory_stats(dev).get('allocation.all.allocated', 0)
        for _ in range(3):
            run_all()
        assert torch.cuda.mem_get_info(dev)[0] == free0
        assert torch.cuda.memory_stats(dev).get('allocation.all.allocated', 0) == count0
        for C, col in problems:
            assert sorted(col.tolist()) == list(range(C.shape[0]))


def test_static_scratch_entry_point_rejects_sizes_beyond_its_maximum():
    from neuron_interpolation_model_fusion_amd import _lib
    lib = _lib.lib
    dev = torch.device('cuda', 0)
    n = _lib.NIF_LAP_STATIC_MAX_N + 1
    C = torch.zeros(n, n, device=dev)
    col = torch.empty(n, dtype=torch.int32, device=dev)
    rc = lib.nif_lap_solve_f32_static(_p(C), n, n, 0, _p(col), None, None, None, None)
    assert rc == -3  # NIF_ERR_UNSUPPORTED
    assert lib.nif_lap_solve_f32_static(None, n, n, 0, _p(col), None, None, None, None) == -1
