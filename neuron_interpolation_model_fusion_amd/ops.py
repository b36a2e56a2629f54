"""Torch-tensor front end of the C ABI (include/nif.h).

Tensors supply device memory and the current HIP stream; every computation below
runs in libnif.so.  Inputs must live on a HIP device: there is no CPU path.
"""
import ctypes

import os
import torch

from . import _lib
from ._lib import lib, check

_DT = {torch.float32: _lib.NIF_F32, torch.float64: _lib.NIF_F64}


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.NifError('libnif ops need tensors on a HIP device (no CPU fallback)')


def _mat(t, name):
    if t.dim() != 2:
        raise ValueError(f'{name} must be 2-D, got shape {tuple(t.shape)}')
    if t.shape[1] > 1 and t.stride(1) != 1:
        t = t.contiguous()
    if t.shape[0] > 1 and t.stride(0) < t.shape[1]:
        t = t.contiguous()
    return t


def _ld(t):
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)


def xty(A, B=None, *, acc='f32', relu_a=False, relu_b=False, trans_a=False, trans_b=False, norms=True,
        rows_sharded=False, split_bf16=False, out=None):
    """G = A^T B over the rows (f64, n0 x n1) and the squared column norms (f64).

    rows_sharded=True marks a contraction over the fusion samples: inside a
    parallel.SampleSharding context the outputs are summed over the ranks (one all-reduce).

    A: m x n0, B: m x n1 (or n x m when trans_*).  B=None means B is A (symmetric mode).
    acc: 'f32' (exact-f32 MFMA per row block, f64 across blocks) or 'f64' (f64 MFMA).
    split_bf16=True (f32 inputs, acc 'f32', no transposes): the products run as six bf16 matrix instructions
    on the exact three-way bf16 split of every operand -- f32-level accuracy, different last bits.
    out=(G, na, nb): contiguous f64 device tensors to write into (e.g. views of one packed buffer that goes
    straight into an all-reduce) instead of fresh allocations.
    Returns (G, na, nb); na/nb are None when norms=False.
    """
    symmetric = B is None
    A = _mat(A, 'A')
    B = A if symmetric else _mat(B, 'B')
    _need_gpu(A, B)
    if A.dtype != B.dtype or A.dtype not in _DT:
        raise TypeError(f'xty needs matching float32/float64 inputs, got {A.dtype} / {B.dtype}')
    if symmetric:
        trans_b, relu_b = trans_a, relu_a
    m = A.shape[1] if trans_a else A.shape[0]
    mb = B.shape[1] if trans_b else B.shape[0]
    if m != mb:
        raise ValueError(f'row counts differ: {m} vs {mb}')
    n0 = A.shape[0] if trans_a else A.shape[1]
    n1 = B.shape[0] if trans_b else B.shape[1]
    accv = _lib.NIF_F64 if (acc == 'f64' or A.dtype == torch.float64) else _lib.NIF_F32
    flags = ((_lib.NIF_RELU_A if relu_a else 0) | (_lib.NIF_RELU_B if relu_b else 0) |
             (_lib.NIF_SYMMETRIC if symmetric else 0) | (_lib.NIF_TRANS_A if trans_a else 0) |
             (_lib.NIF_TRANS_B if trans_b else 0) | (_lib.NIF_XTY_SPLIT_BF16 if split_bf16 else 0))
    if rows_sharded and symmetric and out is None:
        from . import parallel
        if parallel.SampleSharding.wants_packed(n0):
            # one collective of the block upper triangle, written by the reduction kernel and expanded by one kernel
            buf, plen = xty_sym_packed(A, acc=acc, relu=relu_a, trans=trans_a, norms=norms)
            parallel.all_reduce_sum(buf, parallel.SampleSharding._active.group)
            G = unpack_upper(buf, n0)
            na = buf[plen:plen + n0] if norms else None
            return G, na, na
    with torch.cuda.device(A.device):
        if out is not None:
            G, na, nb = out
            ok = (G.dtype == torch.float64 and G.shape == (n0, n1) and G.is_contiguous() and G.device == A.device and
                  all(t is None or (t.dtype == torch.float64 and t.is_contiguous() and t.device == A.device and
                                    t.numel() == n) for t, n in ((na, n0), (nb, n1))))
            if not ok or (norms and (na is None or nb is None)):
                raise ValueError('xty: out must be contiguous float64 (n0 x n1, n0, n1) tensors on the operands\' device')
            if not norms:
                na = nb = None
        else:
            G = torch.empty((n0, n1), dtype=torch.float64, device=A.device)
            na = torch.empty(n0, dtype=torch.float64, device=A.device) if norms else None
            nb = torch.empty(n1, dtype=torch.float64, device=A.device) if norms else None
        nbytes = lib.nif_xty_workspace_bytes(m, n0, n1, accv, flags)
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=A.device)
        check(lib.nif_xty(_ptr(A), _ld(A), _ptr(B), _ld(B), m, n0, n1, _DT[A.dtype], accv, flags,
                          _ptr(G), n1, _ptr(na), _ptr(nb), _ptr(ws), nbytes, _stream()), 'nif_xty')
    if rows_sharded:
        from .parallel import SampleSharding
        Gr, nar, nbr = SampleSharding.reduce(G, na, nb, symmetric=symmetric)
        if out is not None and Gr is not G:  # the caller's buffers must hold what is returned: the reduced sums
            G.copy_(Gr)
            if na is not None:
                na.copy_(nar)
            if nb is not None:
                nb.copy_(nbr)
        else:
            G, na, nb = Gr, nar, nbr
    return G, na, nb


def xty_sym_packed(A, *, acc='f32', relu=False, trans=False, norms=False, block=None, tail=0):
    """This rank's partial A^T A as the packed block upper triangle (parallel.pack_gram_upper's layout), written by
    the split-K reduction itself -- no full matrix, no packing copies.  Returns (buf, plen): one f64 buffer
    [packed | na (n, with norms=True) | `tail` spare entries] ready for the all-reduce, and the length of the packed
    part.  unpack_upper() expands the reduced buffer."""
    from . import parallel
    A = _mat(A, 'A')
    _need_gpu(A)
    if A.dtype not in _DT:
        raise TypeError(f'xty_sym_packed needs float32/float64 input, got {A.dtype}')
    block = int(block or parallel.TRI_BLOCK)
    m = A.shape[1] if trans else A.shape[0]
    n = A.shape[0] if trans else A.shape[1]
    accv = _lib.NIF_F64 if (acc == 'f64' or A.dtype == torch.float64) else _lib.NIF_F32
    flags = (_lib.NIF_RELU_A | _lib.NIF_RELU_B if relu else 0) | _lib.NIF_SYMMETRIC | \
        (_lib.NIF_TRANS_A | _lib.NIF_TRANS_B if trans else 0)
    plen = lib.nif_packed_upper_len(n, block)
    with torch.cuda.device(A.device):
        buf = torch.empty(plen + (n if norms else 0) + tail, dtype=torch.float64, device=A.device)
        na = buf[plen:plen + n] if norms else None
        nbytes = lib.nif_xty_workspace_bytes(m, n, n, accv, flags)
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=A.device)
        check(lib.nif_xty_sym_packed(_ptr(A), _ld(A), m, n, _DT[A.dtype], accv, flags, block, _ptr(buf), _ptr(na), _ptr(ws),
                                     nbytes, _stream()), 'nif_xty_sym_packed')
    return buf, plen


def unpack_upper(buf, n, *, block=None, dtype=torch.float64, gdiag=False):
    """The full symmetric n x n matrix from a (reduced) packed block upper triangle, as f64 or as the f32 k-means
    storage; gdiag=True also returns the unrounded f64 diagonal."""
    from . import parallel
    _need_gpu(buf)
    block = int(block or parallel.TRI_BLOCK)
    if buf.dtype != torch.float64 or not buf.is_contiguous() or buf.numel() < lib.nif_packed_upper_len(n, block):
        raise ValueError('unpack_upper: a contiguous float64 buffer of at least nif_packed_upper_len(n, block) entries')
    with torch.cuda.device(buf.device):
        G = torch.empty((n, n), dtype=dtype, device=buf.device)
        gd = torch.empty(n, dtype=torch.float64, device=buf.device) if gdiag else None
        check(lib.nif_unpack_upper(_ptr(buf), n, block, _ptr(G), n, _DT[dtype], _ptr(gd), _stream()), 'nif_unpack_upper')
    return (G, gd) if gdiag else G


def cost_dist(G, na, nb, s0=None, s1=None, colscale=None):
    """C = scale * sqrt(max(na_i + nb_j - 2 G_ij, 0)) as f32 (see nif_cost_dist)."""
    _need_gpu(G, na, nb, s0, s1, colscale)
    n0, n1 = G.shape
    f32 = lambda t: None if t is None else t.to(device=G.device, dtype=torch.float32).contiguous()
    s0, s1, colscale = f32(s0), f32(s1), f32(colscale)
    with torch.cuda.device(G.device):
        C = torch.empty((n0, n1), dtype=torch.float32, device=G.device)
        check(lib.nif_cost_dist(_ptr(G), G.stride(0), _ptr(na), _ptr(nb), _ptr(s0), _ptr(s1), _ptr(colscale),
                                n0, n1, _ptr(C), n1, _stream()), 'nif_cost_dist')
    return C


def cost_hf(G01, pp0, pp1, r0, r1, s0, s1):
    """HF cost matrix from projected-activation Gram quantities (see nif_cost_hf)."""
    _need_gpu(G01, pp0, pp1, r0, r1)
    n0, n1 = G01.shape
    s0 = s0.to(device=G01.device, dtype=torch.float32).contiguous()
    s1 = s1.to(device=G01.device, dtype=torch.float32).contiguous()
    with torch.cuda.device(G01.device):
        C = torch.empty((n0, n1), dtype=torch.float32, device=G01.device)
        check(lib.nif_cost_hf(_ptr(G01), G01.stride(0), _ptr(pp0), _ptr(pp1), _ptr(r0), _ptr(r1), _ptr(s0), _ptr(s1),
                              n0, n1, _ptr(C), n1, _stream()), 'nif_cost_hf')
    return C


LAP_CERT_TOL = 1e-9  # relative reduced-cost margin below which a non-matching edge counts as a possible tie


def _has_duplicate_lines(C):
    """0-d bool tensor: C seems to have two identical rows or two identical columns (what dead units of a
    random-initialised network produce: exact structural ties, hence an ambiguous optimum).  Compared through the f64
    sum and the maximum of every line: identical lines give identical bits.  Only a filter that spares a fast solve
    whose certificate would fail -- the certificate itself decides whether the fast answer is used."""
    found = None
    for dim in (0, 1):
        s1 = C.sum(dim, dtype=torch.float64)
        s2 = C.amax(dim)
        idx = torch.argsort(s1)
        a, b = s1[idx], s2[idx]
        d = ((a[1:] == a[:-1]) & (b[1:] == b[:-1])).any()
        found = d if found is None else (found | d)
    return found


def lap_solve(C, duals=False, scipy_order=False):
    """Min-cost perfect matching of a square f32 cost matrix on the device.

    Returns col4row (int32, device) -- scipy's ``col_ind`` -- and, with duals=True, also (row4col, u, v).

    scipy_order=True: the matching scipy.optimize.linear_sum_assignment returns, also when the optimum is not unique
    because of exact structural ties.  Where the parallel dual initialisation already matches most rows (trained or
    planted models) the fast solver finishes the problem and a certificate pass checks that no edge outside the
    matching closes an alternating cycle of near-tight edges (``nif_lap_certify_unique_f32``): the optimum is then unique and
    scipy's answer is this one.
    Otherwise -- random-initialised networks with dead units are ambiguous almost everywhere -- the kernel that
    replays scipy's augmentation order and tie rule (NIF_LAP_SCIPY_ORDER) runs.  scipy_order='replay' forces that
    kernel.  ``lap_solve.last_path`` names the path taken ('fast', 'certified', 'replay').
    """
    _need_gpu(C)
    C = _mat(C, 'C')
    if C.dtype != torch.float32 or C.shape[0] != C.shape[1]:
        raise TypeError('lap_solve needs a square float32 matrix')
    n = C.shape[0]
    if n > _lib.NIF_LAP_MAX_N:
        raise _lib.NifError(f'lap_solve: n={n} exceeds NIF_LAP_MAX_N={_lib.NIF_LAP_MAX_N}')
    with torch.cuda.device(C.device):
        col = torch.empty(n, dtype=torch.int32, device=C.device)
        row = torch.empty(n, dtype=torch.int32, device=C.device)
        u = torch.empty(n, dtype=torch.float64, device=C.device)
        v = torch.empty(n, dtype=torch.float64, device=C.device)
        nbytes = lib.nif_lap_workspace_bytes(n)
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=C.device)
        lap_solve.last_workspace = ws  # tools read the kernel's step counter out of it

        def run(flags):
            check(lib.nif_lap_solve_f32(_ptr(C), _ld(C), n, flags, _ptr(col), _ptr(row), _ptr(u), _ptr(v), _ptr(ws),
                                        nbytes, _stream()), 'nif_lap_solve_f32')

        path = 'fast'
        if scipy_order and bool((torch.isnan(C) | (C == float('-inf'))).any()):
            # scipy.optimize.linear_sum_assignment refuses these before it starts
            raise ValueError('matrix contains invalid numeric entries')
        if scipy_order is True and n >= 32:
            path = 'replay'
            run(_lib.NIF_LAP_INIT_ONLY)
            free = int((col < 0).sum().item())
            # the duplicate-line filter only where it decides something (four reductions over C: 0.5 ms at n = 8192)
            if free * 8 <= n or not bool(_has_duplicate_lines(C).item()):
                # nearly solved by the dual initialisation: a budget of Dijkstra steps keeps a hard instance from being
                # solved twice (what the budget leaves unmatched fails the certificate).  Not nearly solved, but no two
                # rows or columns alike (unrelated trained models): the fast solver -- augmenting row reduction + the
                # remaining searches, several times quicker than the replay -- and the same certificate; a failed
                # certificate (an ambiguous optimum after all) costs that solve and falls through to the replay.
                budget = ((16 * free + 64) << _lib.NIF_LAP_BUDGET_SHIFT) if free * 8 <= n else 0
                run(_lib.NIF_LAP_RESUME | budget)
                cnt = torch.empty(1, dtype=torch.int32, device=C.device)
                cbytes = lib.nif_lap_certify_workspace_bytes(n)
                cws = torch.empty(cbytes, dtype=torch.uint8, device=C.device)
                check(lib.nif_lap_certify_unique_f32(_ptr(C), _ld(C), n, _ptr(col), _ptr(row), _ptr(u), _ptr(v), LAP_CERT_TOL,
                                                     _ptr(cnt), _ptr(cws), cbytes, _stream()), 'nif_lap_certify_unique_f32')
                if int(cnt.item()) == 0:
                    path = 'certified'
            if path == 'replay':
                run(_lib.NIF_LAP_SCIPY_ORDER)
        elif scipy_order:
            path = 'replay'
            run(_lib.NIF_LAP_SCIPY_ORDER)
        else:
            run(0)
        lap_solve.last_path = path
        if scipy_order and path == 'replay' and bool((col < 0).any()):
            # rows whose finite entries cannot all be matched stay free; scipy raises here too
            raise ValueError('cost matrix is infeasible')
    if duals:
        return col, row, u, v
    return col


LAP_BATCH_STACK_MAX_N = 2048  # lap_solve_many stacks same-size cost matrices (a copy) up to this size


def lap_solve_batch(C, duals=False, flags=0, stream=None):
    """The fast assignment solver (``lap_solve(C)``) on a stack of independent problems of one size, C [B, n, n] f32
    (row stride and problem stride taken from the tensor): ONE launch per kernel for the whole stack, the
    single-workgroup phases of the B problems side by side on B compute units (nif_lap_solve_batch_f32).  Returns
    col4row [B, n] int32 -- row b is bit-identical to ``lap_solve(C[b])`` -- and with duals=True (col4row, row4col, u, v).
    ``flags``: NIF_LAP_* of nif_lap_solve_f32, the same for every problem (0 = the fast path).  ``stream``: a
    torch.cuda.Stream to launch on instead of the current one (inputs and outputs are allocated on the current stream; the
    caller orders the two streams)."""
    _need_gpu(C)
    if C.dtype != torch.float32 or C.dim() != 3 or C.shape[1] != C.shape[2]:
        raise TypeError('lap_solve_batch needs a [B, n, n] float32 tensor')
    B, n = C.shape[0], C.shape[1]
    if n > _lib.NIF_LAP_MAX_N:
        raise _lib.NifError(f'lap_solve_batch: n={n} exceeds NIF_LAP_MAX_N={_lib.NIF_LAP_MAX_N}')
    if B == 0:
        raise ValueError('lap_solve_batch: empty batch')
    if C.stride(2) != 1 or C.stride(1) < n or (B > 1 and C.stride(0) < (n - 1) * C.stride(1) + n):
        C = C.contiguous()
    with torch.cuda.device(C.device):
        col = torch.empty((B, n), dtype=torch.int32, device=C.device)
        row = torch.empty((B, n), dtype=torch.int32, device=C.device)
        u = torch.empty((B, n), dtype=torch.float64, device=C.device)
        v = torch.empty((B, n), dtype=torch.float64, device=C.device)
        nbytes = lib.nif_lap_batch_workspace_bytes(n, B)
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=C.device)
        lap_solve_batch.last_workspace = ws
        check(lib.nif_lap_solve_batch_f32(_ptr(C), C.stride(1), C.stride(0), n, B, flags, _ptr(col), _ptr(row), _ptr(u),
                                          _ptr(v), _ptr(ws), nbytes,
                                          _stream() if stream is None else ctypes.c_void_p(stream.cuda_stream)),
              'nif_lap_solve_batch_f32')
    if duals:
        return col, row, u, v
    return col


def lap_solve_many(costs, scipy_order=True, rows=False):
    """``lap_solve(C, scipy_order=scipy_order)`` for several independent square f32 cost matrices at once (the layers of
    Git Re-Basin / OTFusion in activation mode: their costs come from activations captured once, so nothing orders the
    assignments).  The augmenting kernel is ONE workgroup per problem, so the problems run side by side on their own
    HIP streams, and the host decisions of the scipy-order path (nearly solved? certificate clean?) are read back once
    per stage for all problems instead of once per problem.  Returns the list of col4row vectors (with rows=True:
    (col4row list, row4col list)); ``last_paths`` names the path each problem took."""
    for C in costs:
        _need_gpu(C)
        if C.dtype != torch.float32 or C.dim() != 2 or C.shape[0] != C.shape[1]:
            raise TypeError('lap_solve_many needs square float32 matrices')
        if C.shape[0] > _lib.NIF_LAP_MAX_N:
            raise _lib.NifError(f'lap_solve_many: n={C.shape[0]} exceeds NIF_LAP_MAX_N={_lib.NIF_LAP_MAX_N}')
    if not costs:
        return ([], []) if rows else []
    costs = [_mat(C, 'C') for C in costs]
    dev = costs[0].device
    with torch.cuda.device(dev):
        if scipy_order:
            bad = torch.stack([(torch.isnan(C) | (C == float('-inf'))).any() for C in costs])
            if bool(bad.any()):
                raise ValueError('matrix contains invalid numeric entries')
        main = torch.cuda.current_stream(dev)
        # fast mode: problems of one size go through ONE batched call (their single-workgroup phases side by side on as
        # many compute units, one launch per kernel); they need no stream or buffers of their own
        sizes = {}
        if not scipy_order:
            for C in costs:
                sizes[C.shape[0]] = sizes.get(C.shape[0], 0) + 1
        P = []
        for C in costs:
            n = C.shape[0]
            if sizes.get(n, 0) >= 2 and n <= LAP_BATCH_STACK_MAX_N:
                P.append(dict(C=C, n=n, batched=True, path='fast'))
                continue
            nbytes = lib.nif_lap_workspace_bytes(n)
            st = torch.cuda.Stream(dev)
            st.wait_stream(main)
            P.append(dict(C=C, n=n, nbytes=nbytes, stream=st, batched=False,
                          col=torch.empty(n, dtype=torch.int32, device=dev), row=torch.empty(n, dtype=torch.int32, device=dev),
                          u=torch.empty(n, dtype=torch.float64, device=dev), v=torch.empty(n, dtype=torch.float64, device=dev),
                          ws=torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev), path='fast'))

        def run(p, flags):
            with torch.cuda.stream(p['stream']):
                check(lib.nif_lap_solve_f32(_ptr(p['C']), _ld(p['C']), p['n'], flags, _ptr(p['col']), _ptr(p['row']),
                                            _ptr(p['u']), _ptr(p['v']), _ptr(p['ws']), p['nbytes'],
                                            ctypes.c_void_p(p['stream'].cuda_stream)), 'nif_lap_solve_f32')

        side = []  # (stream, tensors kept alive) of the batched calls

        def join():
            for p in P:
                if not p['batched']:
                    main.wait_stream(p['stream'])
            for st, *_ in side:
                main.wait_stream(st)

        def replay_all(ps):
            """The kernel that replays scipy's order for `ps` (main has joined their streams): problems of one size as ONE
            batched call on a stream of its own -- their single-workgroup searches side by side however many they are --
            the rest on their own streams (four hardware queues: four at a time)."""
            by_size = {}
            for p in ps:
                by_size.setdefault(p['n'], []).append(p)
            for n_, g in by_size.items():
                if len(g) >= 2 and n_ <= LAP_BATCH_STACK_MAX_N:
                    stacked = torch.stack([p['C'] for p in g])
                    st = torch.cuda.Stream(dev)
                    st.wait_stream(main)
                    col, row, u, v = lap_solve_batch(stacked, duals=True, flags=_lib.NIF_LAP_SCIPY_ORDER, stream=st)
                    side.append((st, stacked, lap_solve_batch.last_workspace))
                    for b, p in enumerate(g):
                        p['col'], p['row'], p['u'], p['v'] = col[b], row[b], u[b], v[b]
                else:
                    for p in g:
                        run(p, _lib.NIF_LAP_SCIPY_ORDER)

        if not scipy_order:
            groups = {}
            for p in P:
                if p['batched']:
                    groups.setdefault(p['n'], []).append(p)
                else:
                    run(p, 0)  # on its own stream (a process has 4 hardware queues: streams overlap four problems at most)
            for ps in groups.values():  # one batched call per size, the sizes side by side on streams of their own
                stacked = torch.stack([p['C'] for p in ps])
                st = torch.cuda.Stream(dev)
                st.wait_stream(main)
                col, row, u, v = lap_solve_batch(stacked, duals=True, stream=st)
                side.append((st, stacked, lap_solve_batch.last_workspace))
                for b, p in enumerate(ps):
                    p['col'], p['row'], p['u'], p['v'] = col[b], row[b], u[b], v[b]
        else:
            for p in P:
                p['staged'] = scipy_order is True and p['n'] >= 32
            staged = [p for p in P if p['staged']]
            replays = []
            for p in P:
                p['path'] = 'replay'
                if p['staged']:
                    run(p, _lib.NIF_LAP_INIT_ONLY)
                else:
                    replays.append(p)
            if not staged:
                replay_all(replays)
            else:
                join()
                free = torch.stack([(p['col'] < 0).sum() for p in staged]).tolist()   # one read-back for all
                # the duplicate-line filter only for the problems the initialisation did not nearly solve (one more
                # read-back for those: four reductions over C are 0.5 ms at n = 8192, a planted C3 level is 11 ms)
                hard = [p for p, f in zip(staged, free) if f * 8 > p['n']]
                for p in staged:
                    p['dupv'] = 0
                if hard:
                    for p in hard:
                        with torch.cuda.stream(p['stream']):
                            p['dup'] = _has_duplicate_lines(p['C']).long()
                    join()
                    for p, d in zip(hard, torch.stack([p['dup'] for p in hard]).tolist()):
                        p['dupv'] = d
                certs = []
                for p, f in zip(staged, free):
                    dup = p['dupv']
                    if f * 8 <= p['n'] or not dup:
                        # nearly solved: finish under a step budget; no two lines alike: full fast solve (see lap_solve)
                        budget = ((16 * f + 64) << _lib.NIF_LAP_BUDGET_SHIFT) if f * 8 <= p['n'] else 0
                        run(p, _lib.NIF_LAP_RESUME | budget)
                        with torch.cuda.stream(p['stream']):
                            p['cnt'] = torch.empty(1, dtype=torch.int32, device=dev)
                            cbytes = lib.nif_lap_certify_workspace_bytes(p['n'])
                            p['cws'] = torch.empty(cbytes, dtype=torch.uint8, device=dev)
                            check(lib.nif_lap_certify_unique_f32(_ptr(p['C']), _ld(p['C']), p['n'], _ptr(p['col']),
                                                                 _ptr(p['row']), _ptr(p['u']), _ptr(p['v']), LAP_CERT_TOL,
                                                                 _ptr(p['cnt']), _ptr(p['cws']), cbytes,
                                                                 ctypes.c_void_p(p['stream'].cuda_stream)),
                                  'nif_lap_certify_unique_f32')
                        certs.append(p)
                    else:
                        replays.append(p)
                replay_all(replays)
                if certs:
                    join()
                    failed = []
                    for p, c in zip(certs, torch.cat([p['cnt'] for p in certs]).tolist()):
                        if c == 0:
                            p['path'] = 'certified'
                        else:
                            failed.append(p)
                    replay_all(failed)
        join()
        for p in P:  # the per-problem buffers were used on side streams
            if p['batched']:
                continue
            for t in (p['col'], p['row'], p['u'], p['v'], p['ws'], p.get('cws'), p.get('cnt'), p.get('dup')):
                if t is not None:
                    t.record_stream(p['stream'])
        if scipy_order:
            infeasible = torch.stack([(p['col'] < 0).any() for p in P if p['path'] == 'replay'] or
                                     [torch.zeros((), dtype=torch.bool, device=dev)])
            if bool(infeasible.any()):
                raise ValueError('cost matrix is infeasible')
    lap_solve_many.last_paths = [p['path'] for p in P]
    if rows:
        return [p['col'] for p in P], [p['row'] for p in P]
    return [p['col'] for p in P]


def refit_gather(W0, W1, rowsrc=None, colsrc=None, alpha=1.0, beta=1.0, sp=1):
    """out[i, c*sp+e] = a_i W0[i, c*sp+e] + b_i W1[rowsrc[i], colsrc[c]*sp+e] (see nif_refit_gather).

    alpha / beta are floats or per-row f32 tensors; W0 may be None; index tensors are int32.
    W1 is (n_out, n_in*sp) (any trailing shape is flattened); the result has W1's shape.
    """
    _need_gpu(W0, W1, rowsrc, colsrc)
    shape = W1.shape
    n_out = shape[0]
    W1f = W1.reshape(n_out, -1).contiguous().float()
    width = W1f.shape[1]
    if width % sp:
        raise ValueError('row length is not a multiple of sp')
    W0f = None if W0 is None else W0.reshape(n_out, -1).contiguous().float()
    dev = W1f.device
    i32 = lambda t: None if t is None else t.to(device=dev, dtype=torch.int32).contiguous()
    rowsrc, colsrc = i32(rowsrc), i32(colsrc)
    av = alpha.to(device=dev, dtype=torch.float32).contiguous() if torch.is_tensor(alpha) else None
    bv = beta.to(device=dev, dtype=torch.float32).contiguous() if torch.is_tensor(beta) else None
    with torch.cuda.device(dev):
        out = torch.empty_like(W1f)
        check(lib.nif_refit_gather(_ptr(W0f), _ptr(W1f), _ptr(rowsrc), _ptr(colsrc), _ptr(av), _ptr(bv),
                                   0.0 if av is not None else float(alpha), 0.0 if bv is not None else float(beta),
                                   n_out, width // sp, sp, _ptr(out), _stream()), 'nif_refit_gather')
    return out.reshape(shape)


PROF_XTY, PROF_LAP, PROF_KMEANS, PROF_EIGH_TRD, PROF_SINKHORN = 0, 1, 2, 3, 4


def profile_enable(on=True, slots=None):
    """Record HIP-event spans around the library's dominant kernels; slots: the PROF_* slots wanted (default all).
    Every recorded span costs two event packets on the stream (~5 us of dispatch gap each)."""
    mask = -1 if slots is None else sum(1 << s for s in slots)
    lib.nif_profile_enable(mask if on else 0)


def profile_collect(slot):
    """(total milliseconds, launches) of the events recorded for a slot since the last collect."""
    ms, cnt = ctypes.c_double(0.0), ctypes.c_int64(0)
    check(lib.nif_profile_collect(slot, ctypes.byref(ms), ctypes.byref(cnt)), 'nif_profile_collect')
    return ms.value, cnt.value


def profile_collect_spans(slot, cap=1 << 16):
    """Durations (ms, record order) of the spans recorded for a slot since the last collect, as a list (at most cap)."""
    buf = (ctypes.c_float * cap)()
    cnt = ctypes.c_int64(0)
    check(lib.nif_profile_collect_spans(slot, buf, cap, ctypes.byref(cnt)), 'nif_profile_collect_spans')
    return [float(buf[i]) for i in range(min(cap, cnt.value))]


def profile_work(slot):
    """Algorithmic work booked for a slot's spans since the last call (bytes for PROF_EIGH_TRD)."""
    w = ctypes.c_double(0.0)
    check(lib.nif_profile_work(slot, ctypes.byref(w)), 'nif_profile_work')
    return w.value


# ---- least-squares refit (hf_kf.py:213,233-234,377,397) ---------------------------------------------

_EPS32 = 1.1920928955078125e-07


def padded_rows(m, d, device, dtype=torch.float32):
    """(m, d) view of a zero buffer whose row stride is a multiple of 4 elements, so the Gram kernel can
    use 16-byte loads on design matrices of odd width (d = fan_in + 1)."""
    ld = (d + 3) // 4 * 4
    return torch.zeros((m, ld), dtype=dtype, device=device)[:, :d]


def design_matrix(x, kernel_size=None, stride=1, padding=0, ones=True):
    """(m, d) f32 design matrix of a level for the LS refit (hf_kf.py:196-208, 224-225), row stride padded to a
    multiple of 4 elements: for a conv level (x [B, C, H, W], kernel_size given) the im2col rows ordered
    (sample, out_y, out_x) with columns (in_channel, ky, kx); for a linear level (kernel_size None) x flattened
    per sample.  ones=True appends the bias column.  One launch, no intermediate copies."""
    _need_gpu(x)
    x = x.contiguous().float()
    B = x.shape[0]
    if kernel_size is None:
        C, H, W = x[0].numel(), 1, 1
        kh = kw = sh = sw = 1
        ph = pw = 0
    else:
        if x.dim() != 4:
            raise ValueError(f'design_matrix expects [B, C, H, W] for a conv level, got {tuple(x.shape)}')
        pair = lambda v: (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))
        (kh, kw), (sh, sw), (ph, pw) = pair(kernel_size), pair(stride), pair(padding)
        C, H, W = x.shape[1:]
    OH, OW = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
    d = C * kh * kw + (1 if ones else 0)
    ld = (d + 3) // 4 * 4
    with torch.cuda.device(x.device):
        out = torch.empty((B * OH * OW, ld), dtype=torch.float32, device=x.device)
        check(lib.nif_im2row_f32(_ptr(x), B, C, H, W, kh, kw, sh, sw, ph, pw, 1 if ones else 0, _ptr(out), ld,
                                 _stream()), 'nif_im2row_f32')
    return out[:, :d]


def eigh_kept(G, cutoff_rel):
    """Eigenpairs of the symmetric f64 matrix G above ``cutoff_rel * lambda_max`` (see nif_eigh_kept_f64): returns
    (lam, V) with lam ascending (K,) and the orthonormal eigenvectors in the columns of V (n x K, a view of an
    n x n buffer).  Hand-written tridiagonalisation / multisection / inverse iteration / back-transformation kernels;
    the call blocks the host once (K is data-dependent)."""
    _need_gpu(G)
    if G.dim() != 2 or G.shape[0] != G.shape[1] or G.dtype != torch.float64:
        raise TypeError('eigh_kept needs a square float64 matrix')
    n = G.shape[0]
    if n > _lib.NIF_EIGH_MAX_N:
        raise _lib.NifError(f'eigh_kept: n={n} exceeds NIF_EIGH_MAX_N={_lib.NIF_EIGH_MAX_N}')
    G = _mat(G, 'G')
    with torch.cuda.device(G.device):
        lam = torch.empty(n, dtype=torch.float64, device=G.device)
        V = torch.empty((n, n), dtype=torch.float64, device=G.device)
        nbytes = lib.nif_eigh_kept_workspace_bytes(n)
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=G.device)
        kept = ctypes.c_int32(0)
        check(lib.nif_eigh_kept_f64(_ptr(G), _ld(G), n, float(cutoff_rel), _ptr(lam), _ptr(V), n, ctypes.byref(kept),
                                    _ptr(ws), nbytes, _stream()), 'nif_eigh_kept_f64')
    K = kept.value
    return lam[:K], V[:, :K]


def eigh_trd_launch_counts():
    """{(mode, cpw, two): launches} of every tridiagonalisation step kernel since the process started (diagnostic for
    the parity tests: which trd_step_kernel<CPW, TWO, MODE> instantiations a problem size actually ran)."""
    arr = (ctypes.c_uint64 * 30)()
    check(lib.nif_eigh_trd_launch_counts(arr), 'nif_eigh_trd_launch_counts')
    cpws = (1, 2, 3, 4, 6)
    return {(m, cpws[c], w): int(arr[(m * 5 + c) * 2 + w]) for m in range(3) for c in range(5) for w in range(2)}


def ls_solve(X, A, normalize=False):
    """Minimum-norm least squares W = pinv(X) @ A with torch.linalg.pinv's cutoff for f32 inputs
    (singular values <= max(m, d) * eps_f32 * sigma_max are dropped), from normal equations:

      m >= d:  G = X^T X, R = X^T A (f64 MFMA);  G = V L V^T;  W = V L_kept^-1 V^T R
      m <  d:  K = X X^T;  K = U L U^T;  W = X^T (U L_kept^-1 U^T A)          (dual form)

    One C call (nif_lsq_refit_f32): Gram builds and products on nif_xty, factorisation by nif_eigh_kept_f64 -- no
    vendor solver on the path.  Inside a parallel.SampleSharding context the normal equations are all-reduced over
    the ranks first (primal form, GLOBAL row count in the cutoff) and nif_lsq_solve_f64 finishes.
    normalize=True scales A's columns to unit norm first (fusion_utils.py:38-40).
    Returns dict(W (d x N f64), R = X^T A (d x N f64), na = squared column norms of A (N f64), rank).
    """
    _need_gpu(X, A)
    m, d = X.shape
    N = A.shape[1]
    from .parallel import SampleSharding
    sharded = SampleSharding._active is not None
    rank = ctypes.c_int32(0)
    if not sharded:
        X, A = _mat(X.float(), 'X'), _mat(A.float(), 'A')
        flags = _lib.NIF_LSQ_NORMALIZE if normalize else 0
        with torch.cuda.device(X.device):
            W = torch.empty((d, N), dtype=torch.float64, device=X.device)
            R = torch.empty((d, N), dtype=torch.float64, device=X.device)
            na = torch.empty(N, dtype=torch.float64, device=X.device)
            nbytes = lib.nif_lsq_refit_workspace_bytes(m, d, N, flags)
            if nbytes == 0:
                raise _lib.NifError(f'ls_solve: min(m, d) = {min(m, d)} exceeds NIF_EIGH_MAX_N = {_lib.NIF_EIGH_MAX_N}')
            ws = torch.empty(nbytes, dtype=torch.uint8, device=X.device)
            check(lib.nif_lsq_refit_f32(_ptr(X), _ld(X), m, d, _ptr(A), _ld(A), N, flags, _ptr(W), N, _ptr(R), N,
                                        _ptr(na), ctypes.byref(rank), _ptr(ws), nbytes, _stream()), 'nif_lsq_refit_f32')
        return dict(W=W, R=R, na=na, rank=rank.value)
    # sample-sharded: m is the GLOBAL row count for pinv's cutoff, and the dual form would need every row locally,
    # so the d x d normal equations are reduced and solved on every rank (identical inputs -> identical results).
    # ONE collective per layer: [G (block upper triangle when large) | R | na | m] travel as one f64 buffer.
    from . import parallel
    group = SampleSharding._active.group
    tri = d > 2 * parallel.TRI_BLOCK
    if tri:  # the normal matrix as a packed triangle written by its reduction kernel, the rest behind it in the buffer
        buf, ng = xty_sym_packed(X, acc='f64', tail=d * N + N + 1)
    else:
        ng = d * d
        buf = torch.empty(ng + d * N + N + 1, dtype=torch.float64, device=X.device)
        xty(X, None, acc='f64', norms=False, out=(buf[:ng].view(d, d), None, None))
    # this rank's partial products, written in place (no rows_sharded: everything is reduced together below)
    xty(X, A, acc='f64', out=(buf[ng:ng + d * N].view(d, N), torch.empty(d, dtype=torch.float64, device=X.device),
                              buf[ng + d * N:ng + d * N + N]))
    buf[-1] = float(m)
    parallel.all_reduce_sum(buf, group)
    G = unpack_upper(buf, d) if tri else buf[:ng].view(d, d)
    R = buf[ng:ng + d * N].view(d, N)
    na = buf[ng + d * N:ng + d * N + N]
    m = int(round(float(buf[-1].item())))
    if normalize:
        R = R / na.sqrt()[None, :]
        na = torch.ones_like(na)
    G, R = G.contiguous(), R.contiguous()
    rtol = max(m, d) * _EPS32
    with torch.cuda.device(X.device):
        W = torch.empty((d, N), dtype=torch.float64, device=X.device)
        nbytes = lib.nif_lsq_solve_workspace_bytes(d, N)
        if nbytes == 0:
            raise _lib.NifError(f'ls_solve: d = {d} exceeds NIF_EIGH_MAX_N = {_lib.NIF_EIGH_MAX_N}')
        ws = torch.empty(nbytes, dtype=torch.uint8, device=X.device)
        check(lib.nif_lsq_solve_f64(_ptr(G), d, _ptr(R), N, d, N, rtol * rtol, _ptr(W), N, ctypes.byref(rank), _ptr(ws),
                                    nbytes, _stream()), 'nif_lsq_solve_f64')
    return dict(W=W, R=R, na=na, rank=rank.value)


def refit_pairs(W, col, s0, s1, n0):
    """HF fused neurons: out[i] = a_i W[:, i] + b_i W[:, n0 + col[i]] as rows (n0 x d f32)."""
    _need_gpu(W, col)
    d = W.shape[0]
    dev = W.device
    col = col.to(device=dev, dtype=torch.int32).contiguous()
    s0 = s0.to(device=dev, dtype=torch.float32).contiguous()
    s1 = s1.to(device=dev, dtype=torch.float32).contiguous()
    with torch.cuda.device(dev):
        out = torch.empty((n0, d), dtype=torch.float32, device=dev)
        check(lib.nif_refit_pairs(_ptr(W), W.stride(0), _ptr(col), _ptr(s0), _ptr(s1), n0, d, _ptr(out), _stream()),
              'nif_refit_pairs')
    return out


# ---- Gram-form k-means --------------------------------------------------------------------------------

def kmeans_gram_storage(G):
    """(G as the k-means kernels stream it, its f64 diagonal): G rounded once to f32 (half the bytes of every Lloyd
    step and every farthest-point pick; products and sums stay f64, the diagonal stays exact), or left in f64 with
    NIF_KMEANS_G=f64.  The reference measures these distances in f32 itself (torch.cdist); all its goldens give the
    same labels either way."""
    gdiag = torch.diagonal(G).to(torch.float64).contiguous()
    if G.dtype == torch.float64 and os.environ.get('NIF_KMEANS_G', 'f32') != 'f64':
        return G.float(), gdiag
    return G, gdiag


def kmeans_gram(A, *, rows_sharded=False):
    """``kmeans_gram_storage(xty(A, None, acc='f32', norms=False)[0])`` -- the symmetric Gram of the clustering vectors in
    the storage the k-means kernels stream, (G f32, diagonal f64) -- without the f64 matrix in between: the split-K
    reduction rounds each entry as it writes it and emits the diagonal unrounded (nif_xty_sym_gf32; same values bit
    for bit, a third of the bytes and no conversion pass: at N = 24 576 the f64 matrix alone is 4.8 GB).  Inside a
    SampleSharding context (the partial Grams are summed over the ranks in f64), for f64 inputs or with
    NIF_KMEANS_G=f64 it takes the two steps."""
    from .parallel import SampleSharding
    A = _mat(A, 'A')
    _need_gpu(A)
    if (rows_sharded and A.dtype == torch.float32 and os.environ.get('NIF_KMEANS_G', 'f32') != 'f64' and
            SampleSharding.wants_packed(A.shape[1])):
        # sharded, large N: the ranks' partial Grams travel as packed f64 triangles and the reduced triangle is expanded
        # straight into the f32 storage + f64 diagonal -- the f64 N x N matrix (4.8 GB at N = 24 576) is never formed
        from . import parallel
        buf, _ = xty_sym_packed(A, acc='f32')
        parallel.all_reduce_sum(buf, SampleSharding._active.group)
        return unpack_upper(buf, A.shape[1], dtype=torch.float32, gdiag=True)
    two_steps = (A.dtype != torch.float32 or os.environ.get('NIF_KMEANS_G', 'f32') == 'f64' or
                 (rows_sharded and SampleSharding._active is not None))
    if two_steps:
        G, _, _ = xty(A, None, acc='f32', norms=False, rows_sharded=rows_sharded)
        return kmeans_gram_storage(G)
    m, n = A.shape
    with torch.cuda.device(A.device):
        G = torch.empty((n, n), dtype=torch.float32, device=A.device)
        gdiag = torch.empty(n, dtype=torch.float64, device=A.device)
        nbytes = lib.nif_xty_workspace_bytes(m, n, n, _lib.NIF_F32, _lib.NIF_SYMMETRIC)
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=A.device)
        check(lib.nif_xty_sym_gf32(_ptr(A), _ld(A), m, n, _DT[A.dtype], _lib.NIF_F32, 0, _ptr(G), n, _ptr(gdiag),
                                   _ptr(ws), nbytes, _stream()), 'nif_xty_sym_gf32')
    return G, gdiag


class KMeans:
    """Device state of one Gram-form k-means problem (see nif_kmeans_* in include/nif.h).  G: N x N f64, or f32 with
    its unrounded diagonal in `gdiag` (taken from G itself when not given)."""

    def __init__(self, G, nn, s, k, gdiag=None, tries=1):
        _need_gpu(G, nn)
        self.tries = int(tries)  # problems that share G, nn, s and advance together (the *_many methods)
        if not 1 <= self.tries <= 16:
            raise ValueError('KMeans: 1 <= tries <= 16')
        self.G = None if G is None else G.contiguous()  # None: only centroid_weights() / targets() may be used
        if self.G is not None and self.G.dtype not in (torch.float32, torch.float64):
            raise TypeError(f'KMeans: G must be float32 or float64, got {self.G.dtype}')
        self._f32 = self.G is not None and self.G.dtype == torch.float32
        self._sfx = '_gf32' if self._f32 else ''
        self.dev = nn.device
        self.N = nn.shape[0]
        self.k = int(k)
        self.nn = nn.to(device=self.dev, dtype=torch.float64).contiguous()
        self.s = s.to(device=self.dev, dtype=torch.float32).contiguous()
        self._g = ()
        if self._f32:
            gd = torch.diagonal(self.G) if gdiag is None else gdiag
            self.gdiag = gd.to(device=self.dev, dtype=torch.float64).contiguous()
            self._g = (_ptr(self.gdiag),)
        with torch.cuda.device(self.dev):
            self.nbytes = lib.nif_kmeans_workspace_bytes(self.N, self.k)
            self.ws = torch.empty(max(self.nbytes * self.tries, 1), dtype=torch.uint8, device=self.dev)
            self.mind = torch.empty(self.N, dtype=torch.float64, device=self.dev)

    def _i32(self, n):
        return torch.empty(n, dtype=torch.int32, device=self.dev)

    def farthest_init(self, first):
        """(picks, assign) of the KF farthest-point initialisation started at neuron `first`."""
        with torch.cuda.device(self.dev):
            picks, assign = self._i32(self.k), self._i32(self.N)
            fn = getattr(lib, 'nif_kmeans_farthest_init' + self._sfx)
            check(fn(_ptr(self.G), self.G.stride(0), *self._g, _ptr(self.nn), self.N, self.k, int(first), _ptr(picks),
                     _ptr(assign), _ptr(self.mind), _stream()), 'nif_kmeans_farthest_init' + self._sfx)
        return picks, assign

    def farthest_init_many(self, firsts):
        """farthest_init for several first neurons (KF's tries) in one launch: (picks [T, k], assign [T, N])."""
        firsts = [int(f) for f in firsts]
        T = len(firsts)
        arr = (ctypes.c_int32 * T)(*firsts)
        with torch.cuda.device(self.dev):
            picks = torch.empty((T, self.k), dtype=torch.int32, device=self.dev)
            assign = torch.empty((T, self.N), dtype=torch.int32, device=self.dev)
            mind = torch.empty((T, self.N), dtype=torch.float64, device=self.dev)
            fn = getattr(lib, 'nif_kmeans_farthest_init_batch' + self._sfx)
            check(fn(_ptr(self.G), self.G.stride(0), *self._g, _ptr(self.nn), self.N, self.k, arr, T, _ptr(picks),
                     _ptr(assign), _ptr(mind), _stream()), 'nif_kmeans_farthest_init_batch' + self._sfx)
        return picks, assign

    def farthest_init_redo_count(self):
        """Tries of the multi-workgroup initialisation (N > 4096) that timed out in their exchange and were done again
        by the one-workgroup kernel on this device since the process started (diagnostic; blocks)."""
        n = ctypes.c_uint32(0)
        with torch.cuda.device(self.dev):
            check(lib.nif_kmeans_farthest_init_redo_count(ctypes.byref(n), _stream()), 'nif_kmeans_farthest_init_redo_count')
        return n.value

    def assign_to_neurons(self, c0):
        with torch.cuda.device(self.dev):
            assign = self._i32(self.N)
            fn = getattr(lib, 'nif_kmeans_assign_to_neurons' + self._sfx)
            check(fn(_ptr(self.G), self.G.stride(0), *self._g, _ptr(self.nn), self.N, self.k, int(c0), _ptr(assign),
                     _stream()), 'nif_kmeans_assign_to_neurons' + self._sfx)
        return assign

    def _lloyd_step_sharded(self, assign, arr, L, share):
        """The step with its pass over G split by neuron over the ranks of a parallel.SampleSharding(shard_lloyd=True)
        context (every rank holds the same G, assign, s): this rank's 64-neuron blocks, then one all-reduce each for
        the labels and for distances + cost shares (every entry has one writer: exact), then the cost in the
        unsharded step's order.  Rows of tries outside `arr` stay zero."""
        from . import parallel
        lo, hi, group = share
        T = self.tries if arr is not None else 1
        nb = (self.N + 63) // 64
        with torch.cuda.device(self.dev):
            out = torch.zeros((T, self.N), dtype=torch.int32, device=self.dev)
            f = torch.zeros(T * (self.N + nb), dtype=torch.float64, device=self.dev)  # distances | cost shares
            nearest, partial = f[:T * self.N].view(T, self.N), f[T * self.N:].view(T, nb)
            cost = torch.zeros(T, dtype=torch.float64, device=self.dev)
            kprime = torch.zeros(T, dtype=torch.int32, device=self.dev)
            fn = getattr(lib, 'nif_kmeans_lloyd_step_range' + self._sfx)
            check(fn(_ptr(self.G), self.G.stride(0), *self._g, _ptr(self.nn), _ptr(self.s), _ptr(assign), self.N, self.k,
                     arr, L, lo, hi, _ptr(out), _ptr(nearest), _ptr(partial), _ptr(kprime), _ptr(self.ws),
                     self.nbytes * T, _stream()), 'nif_kmeans_lloyd_step_range' + self._sfx)
            parallel.all_reduce_sum(out, group)
            parallel.all_reduce_sum(f, group)
            check(lib.nif_kmeans_cost_from_partials(_ptr(partial), nb, arr, L, _ptr(cost), _stream()),
                  'nif_kmeans_cost_from_partials')
        return out, nearest, cost, kprime

    def lloyd_step(self, assign):
        """(assign_out, nearest, cost, kprime): cost / kprime are 0-d device tensors."""
        from .parallel import SampleSharding
        share = SampleSharding.lloyd_blocks((self.N + 63) // 64)
        if share is not None:
            out, nearest, cost, kprime = self._lloyd_step_sharded(assign.contiguous(), None, 1, share)
            return out[0], nearest[0], cost[0], kprime[0]
        with torch.cuda.device(self.dev):
            out = self._i32(self.N)
            nearest = torch.empty(self.N, dtype=torch.float64, device=self.dev)
            cost = torch.empty((), dtype=torch.float64, device=self.dev)
            kprime = torch.empty((), dtype=torch.int32, device=self.dev)
            fn = getattr(lib, 'nif_kmeans_lloyd_step' + self._sfx)
            check(fn(_ptr(self.G), self.G.stride(0), *self._g, _ptr(self.nn), _ptr(self.s), _ptr(assign), self.N, self.k,
                     _ptr(out), _ptr(nearest), _ptr(cost), _ptr(kprime), _ptr(self.ws), self.nbytes, _stream()),
                  'nif_kmeans_lloyd_step' + self._sfx)
        return out, nearest, cost, kprime

    def _live(self, live):
        live = [int(t) for t in (range(self.tries) if live is None else live)]
        if not live or min(live) < 0 or max(live) >= self.tries:
            raise ValueError(f'tries {live} outside 0..{self.tries - 1}')
        return (ctypes.c_int32 * len(live))(*live), len(live)

    def lloyd_step_many(self, assign, live=None):
        """lloyd_step for the tries in `live` (default: all) of assign [tries, N]: one launch per kernel for the batch.
        Returns (assign_out [tries, N], nearest [tries, N], cost [tries], kprime [tries]); rows of other tries are
        left unwritten."""
        arr, L = self._live(live)
        assign = assign.contiguous()
        from .parallel import SampleSharding
        share = SampleSharding.lloyd_blocks((self.N + 63) // 64)
        if share is not None:
            return self._lloyd_step_sharded(assign, arr, L, share)
        with torch.cuda.device(self.dev):
            out = torch.empty((self.tries, self.N), dtype=torch.int32, device=self.dev)
            nearest = torch.empty((self.tries, self.N), dtype=torch.float64, device=self.dev)
            cost = torch.zeros(self.tries, dtype=torch.float64, device=self.dev)
            kprime = torch.zeros(self.tries, dtype=torch.int32, device=self.dev)
            fn = getattr(lib, 'nif_kmeans_lloyd_step_batch' + self._sfx)
            check(fn(_ptr(self.G), self.G.stride(0), *self._g, _ptr(self.nn), _ptr(self.s), _ptr(assign), self.N, self.k,
                     arr, L, _ptr(out), _ptr(nearest), _ptr(cost), _ptr(kprime), _ptr(self.ws), self.nbytes * self.tries,
                     _stream()), 'nif_kmeans_lloyd_step_batch' + self._sfx)
        return out, nearest, cost, kprime

    def repair_kf_many(self, assign, nearest, prev, live=None):
        """In-place KF repair of the rows `live` of assign / nearest [tries, N]; returns changed [tries] (int32)."""
        arr, L = self._live(live)
        with torch.cuda.device(self.dev):
            changed = torch.zeros(self.tries, dtype=torch.int32, device=self.dev)
            check(lib.nif_kmeans_repair_kf_batch(_ptr(assign), _ptr(nearest), _ptr(self.s), _ptr(prev), self.N, self.k,
                                                 arr, L, _ptr(changed), _ptr(self.ws), self.nbytes * self.tries,
                                                 _stream()), 'nif_kmeans_repair_kf_batch')
        return changed

    def repair_kf(self, assign, nearest, prev):
        """In-place KF empty-cluster repair; returns the 0-d device flag (assign != prev).any()."""
        with torch.cuda.device(self.dev):
            changed = torch.empty((), dtype=torch.int32, device=self.dev)
            check(lib.nif_kmeans_repair_kf(_ptr(assign), _ptr(nearest), _ptr(self.s), _ptr(prev), self.N, self.k,
                                           _ptr(changed), _ptr(self.ws), self.nbytes, _stream()),
                  'nif_kmeans_repair_kf')
        return changed

    def centroid_weights(self, W, assign):
        """k x d f32 rows: importance-weighted mean of the member columns of W (d x N f64) per
        non-empty cluster in label order; trailing rows zero.  Returns (rows, kprime)."""
        d = W.shape[0]
        with torch.cuda.device(self.dev):
            out = torch.empty((self.k, d), dtype=torch.float32, device=self.dev)
            kprime = torch.empty((), dtype=torch.int32, device=self.dev)
            check(lib.nif_kmeans_centroid_weights(_ptr(W), W.stride(0), _ptr(self.s), _ptr(assign), self.N, self.k, d,
                                                  _ptr(out), _ptr(kprime), _ptr(self.ws), self.nbytes, _stream()),
                  'nif_kmeans_centroid_weights')
        return out, kprime

    def targets(self, A, assign):
        """ALF target matrix (m x k f32): weighted centroid of every cluster's member columns of A
        (m x N f32), non-empty clusters in label order.  Returns (T, kprime)."""
        A = _mat(A, 'A')
        m = A.shape[0]
        with torch.cuda.device(self.dev):
            T = torch.empty((m, self.k), dtype=torch.float32, device=self.dev)
            kprime = torch.empty((), dtype=torch.int32, device=self.dev)
            check(lib.nif_kmeans_targets(_ptr(A), _ld(A), m, _ptr(self.s), _ptr(assign), self.N, self.k, _ptr(T),
                                         self.k, _ptr(kprime), _ptr(self.ws), self.nbytes, _stream()),
                  'nif_kmeans_targets')
        return T, kprime


# ---- fused loss of the ALF level fit -----------------------------------------------------------------

def _loss_dims(t):
    if t.dim() < 2:
        raise ValueError(f'weighted_mse expects [batch, channels, ...], got shape {tuple(t.shape)}')
    inner = 1
    for s in t.shape[2:]:
        inner *= s
    return t.shape[0], t.shape[1], inner


class _WeightedMSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, truth, w):
        outer, C, inner = _loss_dims(pred)
        dev = pred.device
        with torch.cuda.device(dev):
            loss = torch.empty((), dtype=torch.float32, device=dev)
            wsz = lib.nif_wmse_workspace_bytes()
            ws = torch.empty(wsz, dtype=torch.uint8, device=dev)
            check(lib.nif_wmse_forward(_ptr(pred), _ptr(truth), _ptr(w), outer, C, inner, _ptr(loss), _ptr(ws), wsz,
                                       _stream()), 'nif_wmse_forward')
        ctx.save_for_backward(pred, truth, w)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        pred, truth, w = ctx.saved_tensors
        outer, C, inner = _loss_dims(pred)
        grad_out = grad_out.to(torch.float32).contiguous()
        with torch.cuda.device(pred.device):
            grad = torch.empty_like(pred)
            check(lib.nif_wmse_backward(_ptr(pred), _ptr(truth), _ptr(w), _ptr(grad_out), outer, C, inner, _ptr(grad),
                                        _stream()), 'nif_wmse_backward')
        return grad, None, None


def weighted_mse(pred, truth, w=None):
    """mean(w[c] * (pred - truth)^2) over every element, c = index along dim 1 (alf.py:414-422); plain
    MSE when w is None.  pred / truth: [batch, channels, ...] f32; differentiable w.r.t. pred."""
    _need_gpu(pred, truth, w)
    if pred.shape != truth.shape:
        raise ValueError(f'shape mismatch: {tuple(pred.shape)} vs {tuple(truth.shape)}')
    pred = pred.float().contiguous()
    truth = truth.detach().float().contiguous()
    if w is not None:
        w = w.detach().to(device=pred.device, dtype=torch.float32).contiguous()
        if w.numel() != pred.shape[1]:
            raise ValueError(f'w has {w.numel()} entries for {pred.shape[1]} channels')
    return _WeightedMSE.apply(pred, truth, w)


# ---- exact transport with general marginals and the sparse refit --------------------------------------

class SparsePlan:
    """A transport plan T (n0 x n1) by column: colptr[n1 + 1], rowidx / val[nnz] (val f64), on the device."""

    def __init__(self, n0, n1, colptr, rowidx, val):
        self.n0, self.n1, self.colptr, self.rowidx, self.val = n0, n1, colptr, rowidx, val

    @property
    def nnz(self):
        return self.rowidx.shape[0]

    def colidx(self):
        counts = (self.colptr[1:] - self.colptr[:-1]).long()
        return torch.repeat_interleave(torch.arange(self.n1, device=self.colptr.device), counts)

    def dense(self, dtype=torch.float32):
        T = torch.zeros(self.n0, self.n1, dtype=torch.float64, device=self.val.device)
        T[self.rowidx.long(), self.colidx()] = self.val
        return T.to(dtype)

    @staticmethod
    def from_permutation(row4col, value):
        """T[row4col[j], j] = value."""
        n = row4col.shape[0]
        dev = row4col.device
        return SparsePlan(n, n, torch.arange(n + 1, device=dev, dtype=torch.int32),
                          row4col.to(torch.int32).contiguous(),
                          torch.full((n,), float(value), dtype=torch.float64, device=dev))

    def scaled(self, row_scale=None, col_scale=None):
        """CSC triplet (colptr, rowidx, f32 values) of diag(row_scale) T diag(col_scale) for nif_refit_sparse."""
        v = self.val
        if row_scale is not None:
            v = v * row_scale.to(v)[self.rowidx.long()]
        if col_scale is not None:
            v = v * col_scale.to(v)[self.colidx()]
        return self.colptr, self.rowidx, v.float().contiguous()


def _emd_problem(C, a, b):
    """Checked inputs and output buffers of one transport problem."""
    _need_gpu(C, a, b)
    C = _mat(C.float(), 'C')
    n0, n1 = C.shape
    dev = C.device
    a = a.to(device=dev, dtype=torch.float64).contiguous()
    b = b.to(device=dev, dtype=torch.float64)
    b = (b * (a.sum() / b.sum())).contiguous()
    if a.numel() != n0 or b.numel() != n1:
        raise ValueError(f'marginals {a.numel()}, {b.numel()} do not match the cost {n0} x {n1}')
    with torch.cuda.device(dev):
        cap = n0 + n1 + 64
        wsz = lib.nif_emd_workspace_bytes(n0, n1)
        if wsz == 0:
            raise _lib.NifError(f'nif_emd_solve_f32: sizes {n0} x {n1} beyond NIF_EMD_MAX_N')
        return dict(C=C, a=a, b=b, n0=n0, n1=n1, dev=dev, cap=cap, wsz=wsz, tol=a.sum() * 1e-13,
                    colptr=torch.empty(n1 + 1, dtype=torch.int32, device=dev),
                    rowidx=torch.empty(cap, dtype=torch.int32, device=dev),
                    val=torch.empty(cap, dtype=torch.float64, device=dev),
                    status=torch.zeros(16, dtype=torch.int32, device=dev),  # [4..] only a -DNIF_EMD_TIMING build writes
                    u=torch.empty(n0, dtype=torch.float64, device=dev),
                    v=torch.empty(n1, dtype=torch.float64, device=dev),
                    ws=torch.empty(wsz, dtype=torch.uint8, device=dev))


def _emd_launch(p, flags, stream):
    check(lib.nif_emd_solve_f32(_ptr(p['C']), _ld(p['C']), p['n0'], p['n1'], _ptr(p['a']), _ptr(p['b']), p['tol'], flags,
                                _ptr(p['colptr']), _ptr(p['rowidx']), _ptr(p['val']), p['cap'], _ptr(p['status']),
                                _ptr(p['u']), _ptr(p['v']), _ptr(p['ws']), p['wsz'], stream), 'nif_emd_solve_f32')


def _emd_result(p, work, duals):
    nnz = work[0]
    if nnz < 0:
        check(nnz, 'nif_emd_solve_f32 (device status)')
    plan = SparsePlan(p['n0'], p['n1'], p['colptr'], p['rowidx'][:nnz].contiguous(), p['val'][:nnz].contiguous())
    return (plan, p['u'], p['v']) if duals else plan


def emd_solve(C, a, b, duals=False, flags=0):
    """Exact optimal transport plan for cost C (n0 x n1 f32, >= 0) and marginals a, b (``ot.emd(a, b, C)``,
    otfusion.py:32).  b is rescaled to a's total like POT does.  Returns a SparsePlan (and f64 duals u, v).
    flags: NIF_EMD_* tuning flags (cold start, the arc-pool kernel)."""
    p = _emd_problem(C, a, b)
    p['tol'] = float(p['tol'])
    with torch.cuda.device(p['dev']):
        for more in (0, _lib.NIF_EMD_EDGES_GLOBAL):
            _emd_launch(p, flags | more, _stream())
            work = p['status'].tolist()
            if work[0] != _lib.NIF_ERR_WORKSPACE:  # a degenerate problem outgrew the LDS arc pool: retry once
                break
    emd_solve.last_work = dict(augmentations=work[1], steps=work[2], rows_expanded=work[3])
    emd_solve.last_status = work
    return _emd_result(p, work, duals)


EMD_BATCH_MAX_N = 2048  # square problems up to this size can share one batched call (the forest kernel's range)


def _emd_launch_batch(group, stream):
    """One nif_emd_solve_batch_f32 call (on `stream`; everything is allocated on the current stream, which `stream` has
    to wait for and which waits for `stream` before the results are read) for square problems of one size: inputs
    stacked, every problem's output tensors replaced by its rows of the stacked outputs.  False when the library
    declines (sizes / flags outside the batched path): the caller then solves the group problem by problem."""
    n, B, dev = group[0]['n0'], len(group), group[0]['dev']
    cap = group[0]['cap']
    C = torch.stack([p['C'] for p in group])
    a = torch.stack([p['a'] for p in group])
    b = torch.stack([p['b'] for p in group])
    tol = torch.stack([p['tol'] for p in group]).to(torch.float64).contiguous()
    colptr = torch.empty((B, n + 1), dtype=torch.int32, device=dev)
    rowidx = torch.empty((B, cap), dtype=torch.int32, device=dev)
    val = torch.empty((B, cap), dtype=torch.float64, device=dev)
    status = torch.zeros((B, 16), dtype=torch.int32, device=dev)
    u = torch.empty((B, n), dtype=torch.float64, device=dev)
    v = torch.empty((B, n), dtype=torch.float64, device=dev)
    wsz = lib.nif_emd_batch_workspace_bytes(n, B)
    ws = torch.empty(max(wsz, 1), dtype=torch.uint8, device=dev)
    stream.wait_stream(torch.cuda.current_stream(dev))
    rc = lib.nif_emd_solve_batch_f32(_ptr(C), n, n * n, n, B, _ptr(a), _ptr(b), _ptr(tol), 0, _ptr(colptr), _ptr(rowidx),
                                     _ptr(val), cap, _ptr(status), _ptr(u), _ptr(v), _ptr(ws), wsz,
                                     ctypes.c_void_p(stream.cuda_stream))
    if rc == _lib.NIF_ERR_UNSUPPORTED:
        return False
    check(rc, 'nif_emd_solve_batch_f32')
    group[0]['batch_keep'] = (C, a, b, tol, ws)  # alive until the caller has joined the stream
    for i, p in enumerate(group):
        p['colptr'], p['rowidx'], p['val'], p['status'], p['u'], p['v'] = colptr[i], rowidx[i], val[i], status[i], u[i], v[i]
    return True


def emd_solve_many(problems, duals=False):
    """emd_solve for independent problems [(C, a, b), ...], side by side.  Square problems of one size go through ONE
    batched call (nif_emd_solve_batch_f32: the assignment start and the forest kernel with the problems along
    gridDim.z, each on a compute unit of its own); the rest run on a stream each (which overlaps four at a time: a
    process has four hardware queues).  The statuses are read back once.  Same plans as one by one, bit for bit."""
    if not problems:
        return []
    ps = [_emd_problem(C, a, b) for C, a, b in problems]
    dev = ps[0]['dev']
    main = torch.cuda.current_stream(dev)
    with torch.cuda.device(dev):
        groups, rest = {}, []
        for p in ps:
            if p['n0'] == p['n1'] and 2 <= p['n0'] <= EMD_BATCH_MAX_N:
                groups.setdefault(p['n0'], []).append(p)
            else:
                rest.append(p)
        side = []
        for n_, group in groups.items():  # one batched call per size, the sizes side by side on streams of their own
            if len(group) >= 2:
                st = torch.cuda.Stream(dev)
                if _emd_launch_batch(group, st):
                    side.append(st)
                    continue
            rest.extend(group)
        if rest:
            tols = torch.stack([p['tol'] for p in rest]).tolist()  # one read-back for every tolerance
            streams = [torch.cuda.Stream(dev) for _ in rest]
            for p, tol, st in zip(rest, tols, streams):
                p['tol'] = tol
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    _emd_launch(p, 0, st.cuda_stream)
            for st in streams:
                main.wait_stream(st)
        for st in side:
            main.wait_stream(st)
        works = torch.stack([p['status'] for p in ps]).tolist()
        for i, (p, work) in enumerate(zip(ps, works)):
            if work[0] == _lib.NIF_ERR_WORKSPACE:  # outgrew the LDS arc pool: once more with the pool in global memory
                if torch.is_tensor(p['tol']):  # a problem of a batched call: its outputs are rows of the stacked tensors
                    p['tol'] = float(p['tol'])
                    p['status'] = torch.zeros(16, dtype=torch.int32, device=dev)
                    for key in ('colptr', 'rowidx', 'val', 'u', 'v'):
                        p[key] = p[key].clone()
                _emd_launch(p, _lib.NIF_EMD_EDGES_GLOBAL, _stream())
                works[i] = p['status'].tolist()
    emd_solve_many.last_work = [dict(augmentations=w[1], steps=w[2], rows_expanded=w[3]) for w in works]
    return [_emd_result(p, w, duals) for p, w in zip(ps, works)]


def sinkhorn(C, a, b, reg, max_iter=1000, tol=1e-9):
    """Entropic transport plan diag(u) exp(-C / reg) diag(v) by Sinkhorn-Knopp iterations (POT's
    ``ot.sinkhorn(a, b, C, reg)``; NOT what the reference calls -- it uses the exact ot.emd -- an additional approximate
    mode).  Returns dict(T (n0 x n1 f32), u, v (f64), state (0 cap reached / 1 converged / 2 breakdown), iterations,
    err)."""
    _need_gpu(C, a, b)
    C = _mat(C.float(), 'C')
    n0, n1 = C.shape
    dev = C.device
    a = a.to(device=dev, dtype=torch.float64).contiguous()
    b = b.to(device=dev, dtype=torch.float64).contiguous()
    if a.numel() != n0 or b.numel() != n1:
        raise ValueError(f'marginals {a.numel()}, {b.numel()} do not match the cost {n0} x {n1}')
    with torch.cuda.device(dev):
        T = torch.empty((n0, n1), dtype=torch.float32, device=dev)
        u = torch.empty(n0, dtype=torch.float64, device=dev)
        v = torch.empty(n1, dtype=torch.float64, device=dev)
        info = torch.zeros(2, dtype=torch.int32, device=dev)
        err = torch.zeros((), dtype=torch.float64, device=dev)
        wsz = lib.nif_sinkhorn_workspace_bytes(n0, n1)
        ws = torch.empty(wsz, dtype=torch.uint8, device=dev)
        check(lib.nif_sinkhorn_f32(_ptr(C), _ld(C), n0, n1, _ptr(a), _ptr(b), float(reg), int(max_iter), float(tol), _ptr(T),
                                   n1, _ptr(u), _ptr(v), _ptr(info), _ptr(err), _ptr(ws), wsz, _stream()),
              'nif_sinkhorn_f32')
        state, iters = info.tolist()
    return dict(T=T, u=u, v=v, state=state, iterations=iters, err=float(err))


def refit_sparse(W0, W1, rows=None, cols=None, alpha=1.0, beta=1.0, sp=1):
    """out = beta * W0 + alpha * R W1 P with sparse R (n_out x n_k, given as the CSC triplet of its transpose:
    column j lists the model-k rows feeding fused row j) and sparse P acting on the input units (column q lists
    the model-k input units feeding fused input q; each unit spans ``sp`` consecutive positions).  None = identity."""
    _need_gpu(W1, W0)
    W1 = _mat(W1.float(), 'W1')
    dev = W1.device
    n_out = (rows[0].shape[0] - 1) if rows is not None else W1.shape[0]
    n_in = (cols[0].shape[0] - 1) if cols is not None else W1.shape[1] // sp
    if W0 is not None:
        W0 = W0.float().contiguous()
        if tuple(W0.shape) != (n_out, n_in * sp):
            raise ValueError(f'W0 is {tuple(W0.shape)}, expected {(n_out, n_in * sp)}')
    r = rows if rows is not None else (None, None, None)
    c = cols if cols is not None else (None, None, None)
    with torch.cuda.device(dev):
        out = torch.empty((n_out, n_in * sp), dtype=torch.float32, device=dev)
        check(lib.nif_refit_sparse(_ptr(W0), _ptr(W1), _ld(W1), _ptr(r[0]), _ptr(r[1]), _ptr(r[2]), _ptr(c[0]),
                                   _ptr(c[1]), _ptr(c[2]), float(alpha), float(beta), n_out, n_in, sp, _ptr(out),
                                   _stream()), 'nif_refit_sparse')
    return out
