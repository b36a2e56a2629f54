"""-m gpu: the HIP kernels, called through the C ABI, against f64 torch / scipy on the same inputs."""
import numpy as np
import pytest
import torch
from scipy.optimize import linear_sum_assignment

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'these tests need the GPU box'
    from neuron_interpolation_model_fusion_amd import ops as _ops
    return _ops


def _ref_gram(A, B, relu_a=False, relu_b=False):
    A = A.double().cpu()
    B = B.double().cpu()
    if relu_a:
        A = A.clamp_min(0)
    if relu_b:
        B = B.clamp_min(0)
    return A.T @ B, (A * A).sum(0), (B * B).sum(0)


XTY_SHAPES = [
    (48, 32, 32), (400, 256, 256), (1000, 10, 10), (777, 130, 70), (5000, 384, 384),
    (6144, 8, 8), (33, 257, 129), (16, 128, 128), (1, 5, 7), (20000, 64, 200),
]


@pytest.mark.parametrize('m,n0,n1', XTY_SHAPES)
@pytest.mark.parametrize('acc', ['f32', 'f64'])
def test_xty_matches_f64(ops, m, n0, n1, acc):
    g = torch.Generator().manual_seed(m * 7 + n0)
    A = torch.randn(m, n0, generator=g)
    B = torch.randn(m, n1, generator=g) + 0.5
    G, na, nb = ops.xty(A.cuda(), B.cuda(), acc=acc)
    Gr, nar, nbr = _ref_gram(A, B)
    scale = (nar.sqrt()[:, None] * nbr.sqrt()[None, :])
    tol = 2e-6 if acc == 'f32' else 1e-12
    assert ((G.cpu() - Gr).abs() / scale).max().item() < tol
    assert ((na.cpu() - nar).abs() / nar).max().item() < tol
    assert ((nb.cpu() - nbr).abs() / nbr).max().item() < tol


def test_xty_exact_on_integers(ops):
    """Asymmetric small-integer data: every product and sum is exact, so any layout slip shows."""
    m, n0, n1 = 300, 200, 136
    r = torch.arange(m)[:, None]
    A = ((r * 3 + torch.arange(n0)[None, :] * 5) % 7 - 3).float()
    B = ((r * 2 + torch.arange(n1)[None, :] * 11) % 5 - 1).float()
    for acc in ('f32', 'f64'):
        G, na, nb = ops.xty(A.cuda(), B.cuda(), acc=acc)
        Gr, nar, nbr = _ref_gram(A, B)
        assert torch.equal(G.cpu(), Gr)
        assert torch.equal(na.cpu(), nar) and torch.equal(nb.cpu(), nbr)


@pytest.mark.parametrize('acc', ['f32', 'f64'])
def test_xty_symmetric_relu_unaligned(ops, acc):
    g = torch.Generator().manual_seed(5)
    A = torch.randn(901, 300, generator=g)
    G, na, nb = ops.xty(A.cuda(), None, acc=acc, relu_a=True)
    Gr, nar, _ = _ref_gram(A, A, True, True)
    tol = 2e-6 if acc == 'f32' else 1e-12
    assert ((G.cpu() - Gr).abs().max() / Gr.abs().max()).item() < tol
    assert torch.allclose(G.cpu(), G.cpu().T, rtol=0, atol=0)
    assert torch.allclose(na.cpu(), nar, rtol=tol * 10)
    assert torch.allclose(nb.cpu(), nar, rtol=tol * 10)
    # odd leading dimension (no 16-byte vector loads) and a column slice
    Ab = torch.randn(500, 77, generator=g).cuda()
    view = Ab[:, 3:60]
    G2, _, _ = ops.xty(view, Ab, acc=acc)
    Gr2, _, _ = _ref_gram(view, Ab)
    assert ((G2.cpu() - Gr2).abs().max() / Gr2.abs().max()).item() < tol


@pytest.mark.parametrize('acc', ['f32', 'f64'])
@pytest.mark.parametrize('n,m', [(2881, 37), (3200, 20)])
def test_xty_symmetric_large_n_goes_through_the_tile_reduction(ops, acc, n, m):
    """From 45 tiles of 64 per side on, the symmetric reduction sums 64 x 64 tiles of the upper triangle and writes each
    twice (as it stands and transposed): values against f64, exact symmetry, ragged last tile, norms on the side."""
    g = torch.Generator().manual_seed(n)
    A = torch.randn(m, n, generator=g)
    G, na, _ = ops.xty(A.cuda(), None, acc=acc)
    Gr = A.double().T @ A.double()
    tol = 2e-6 if acc == 'f32' else 1e-12
    Gc = G.cpu()
    assert ((Gc - Gr).abs().max() / Gr.abs().max()).item() < tol
    assert torch.equal(Gc, Gc.T)
    assert torch.allclose(na.cpu(), torch.diagonal(Gr), rtol=tol * 10)
    G2, _, _ = ops.xty(A.cuda(), None, acc=acc, norms=False)
    assert torch.equal(G2.cpu(), Gc)


@pytest.mark.parametrize('n,m', [(50, 33), (771, 100), (2950, 24)])
def test_kmeans_gram_equals_the_rounded_f64_gram_bit_for_bit(ops, n, m):
    """ops.kmeans_gram (nif_xty_sym_gf32: f32 storage + f64 diagonal straight out of the split-K reduction) against the
    two steps it replaces, both reduction kernels (per-entry below 45 tiles of 64, tile-transposing above)."""
    g = torch.Generator().manual_seed(n + m)
    A = torch.randn(m, n, generator=g).cuda()
    G, _, _ = ops.xty(A, None, acc='f32', norms=False)
    want, want_d = ops.kmeans_gram_storage(G)
    got, got_d = ops.kmeans_gram(A)
    assert got.dtype == torch.float32 and got_d.dtype == torch.float64
    assert torch.equal(got, want) and torch.equal(got_d, want_d)
    # a column slice (row stride larger than the width)
    B = torch.randn(m, n + 7, generator=g).cuda()
    G2, _, _ = ops.xty(B[:, 3:n + 3], None, acc='f32', norms=False)
    got2, got2_d = ops.kmeans_gram(B[:, 3:n + 3])
    assert torch.equal(got2, G2.float()) and torch.equal(got2_d, torch.diagonal(G2))


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
def test_xty_transposed_operands(ops, dtype):
    g = torch.Generator().manual_seed(9)
    At = torch.randn(150, 420, generator=g, dtype=dtype)  # n0 x m
    B = torch.randn(420, 90, generator=g, dtype=dtype)
    G, na, nb = ops.xty(At.cuda(), B.cuda(), acc='f64', trans_a=True)
    Gr = At.double() @ B.double()
    assert ((G.cpu() - Gr).abs().max() / Gr.abs().max()).item() < 1e-12
    assert torch.allclose(na.cpu(), (At.double() ** 2).sum(1), rtol=1e-12)
    # X X^T (dual normal equations): both operands transposed, symmetric
    G2, _, _ = ops.xty(At.cuda(), None, acc='f64', trans_a=True)
    Gr2 = At.double() @ At.double().T
    assert ((G2.cpu() - Gr2).abs().max() / Gr2.abs().max()).item() < 1e-12
    Bt = B.T.contiguous()
    G3, _, _ = ops.xty(At.cuda(), Bt.cuda(), acc='f64', trans_a=True, trans_b=True)
    assert ((G3.cpu() - Gr).abs().max() / Gr.abs().max()).item() < 1e-12


def test_cost_dist_and_hf(ops):
    g = torch.Generator().manual_seed(11)
    m, n0, n1 = 600, 96, 80
    A = torch.randn(m, n0, generator=g)
    B = torch.randn(m, n1, generator=g)
    s0 = torch.rand(n0, generator=g) + 0.1
    s1 = torch.rand(n1, generator=g) + 0.1
    G, na, nb = ops.xty(A.cuda(), B.cuda())
    C = ops.cost_dist(G, na, nb).cpu()
    ref = torch.cdist(A.double().T, B.double().T)
    assert torch.allclose(C.double(), ref, rtol=2e-6)
    Cg = ops.cost_dist(G, na, nb, s0.cuda(), s1.cuda()).cpu()
    w = 2 * s0[:, None].double() * s1[None, :].double() / (s0[:, None].double() + s1[None, :].double())
    assert torch.allclose(Cg.double(), ref * w, rtol=2e-6)
    # HF closed form against the brute-force definition (hf_kf.py:109-119) on a rank-deficient X
    d = 40
    X = torch.randn(m, d, generator=g, dtype=torch.float64)
    X[:, 5] = X[:, 4]
    P = X @ torch.linalg.pinv(X)
    A64, B64 = A.double(), B.double()
    P0, P1 = P @ A64, P @ B64
    r0 = ((A64 - P0) ** 2).sum(0)
    r1 = ((B64 - P1) ** 2).sum(0)
    Ch = ops.cost_hf((P0.T @ P1).cuda(), (P0 ** 2).sum(0).cuda(), (P1 ** 2).sum(0).cuda(), r0.cuda(), r1.cuda(),
                     s0.cuda(), s1.cuda()).cpu()
    ref_h = torch.zeros(n0, n1, dtype=torch.float64)
    for i in range(n0):
        pred = (s0[i] * P0[:, i:i + 1] + s1[None, :] * P1) / (s0[i] + s1[None, :])
        ref_h[i] = s0[i] * (pred - A64[:, i:i + 1]).norm(dim=0) + s1 * (pred - B64).norm(dim=0)
    assert torch.allclose(Ch.double(), ref_h, rtol=1e-5)


@pytest.mark.parametrize('m,d,dead,dup', [(900, 70, 0, 0), (900, 70, 12, 3), (60, 150, 0, 0), (60, 150, 40, 2), (500, 33, 33 - 4, 0)])
def test_ls_solve_matches_pinv(ops, m, d, dead, dup):
    """ops.ls_solve against torch.linalg.pinv(X) @ A in f32 semantics (hf_kf.py:213,233-234): full rank, exactly
    zero columns (dead channels -- the deflated path), duplicated columns (truncation without a zero column), and
    the dual form (m < d)."""
    g = torch.Generator().manual_seed(1000 + m + d + dead)
    X = torch.randn(m, d, generator=g)
    A = torch.randn(m, 24, generator=g)
    zero = torch.randperm(d, generator=g)[:dead]
    X[:, zero] = 0
    live = [j for j in range(d) if j not in set(zero.tolist())]
    for t in range(dup):
        X[:, live[2 * t + 1]] = X[:, live[2 * t]]
    out = ops.ls_solve(X.cuda(), A.cuda())
    W = out['W'].cpu()
    # reference in f64 with the f32 cutoff (pinv on f32 input uses rtol = max(m, d) * eps_f32)
    X64 = X.double()
    ref = torch.linalg.pinv(X64, rtol=max(m, d) * 1.1920928955078125e-07) @ A.double()
    assert out['rank'] == min(m, d - dead - dup)
    assert (W[zero] == 0).all()
    assert torch.allclose(W, ref, rtol=1e-6, atol=1e-8 * ref.abs().max().item())
    # and against the f32 call the reference actually makes, to its own accuracy
    ref32 = (torch.linalg.pinv(X) @ A).double()
    assert torch.allclose(W, ref32, rtol=1e-3, atol=1e-4 * ref32.abs().max().item())


def _check_lap(ops, C, scipy_order=False):
    n = C.shape[0]
    col, row, u, v = ops.lap_solve(C.cuda(), duals=True, scipy_order=scipy_order)
    col = col.cpu().numpy().astype(np.int64)
    row = row.cpu().numpy()
    assert sorted(col.tolist()) == list(range(n)), 'not a permutation'
    assert (row[col] == np.arange(n)).all()
    Cn = C.double().numpy()
    r, c = linear_sum_assignment(Cn)
    opt = Cn[r, c].sum()
    got = Cn[np.arange(n), col].sum()
    assert abs(got - opt) <= 1e-9 * max(1.0, abs(opt)), (got, opt)
    # dual certificate: feasibility and complementary slackness
    u = u.cpu().numpy()
    v = v.cpu().numpy()
    slack = Cn - u[:, None] - v[None, :]
    assert slack.min() > -1e-9 * max(1.0, np.abs(Cn).max())
    assert np.abs(slack[np.arange(n), col]).max() < 1e-9 * max(1.0, np.abs(Cn).max())
    return col, c


@pytest.mark.parametrize('n', [1, 2, 7, 10, 64, 100, 256, 384, 513, 1024, 1500, 2048, 3000, 4100])
def test_lap_random(ops, n):
    g = torch.Generator().manual_seed(n)
    C = torch.rand(n, n, generator=g) * 10
    col, ref = _check_lap(ops, C)
    assert (col == ref).all()  # continuous random costs: the optimum is unique


def test_lap_planted_and_ties(ops):
    g = torch.Generator().manual_seed(3)
    n = 384
    A = torch.randn(2000, n, generator=g)
    perm = torch.randperm(n, generator=g)
    B = A[:, perm] + 0.1 * torch.randn(2000, n, generator=g)
    C = torch.cdist(A.T, B.T)
    col, ref = _check_lap(ops, C)
    assert (col == ref).all()
    assert (perm[col] == torch.arange(n)).all()
    # heavy ties: integer costs; only optimality is required
    Ci = torch.randint(0, 4, (200, 200), generator=g).float()
    _check_lap(ops, Ci)
    _check_lap(ops, torch.zeros(50, 50))
    # negative costs
    _check_lap(ops, torch.randn(300, 300, generator=g))


@pytest.mark.parametrize('n', [1, 2, 5, 33, 64, 200, 384, 1025, 2100, 4100])
def test_lap_scipy_order_reproduces_scipy_under_ties(ops, n):
    """Structural ties (duplicated rows / columns, constant blocks, small integers): the scipy-order
    mode must return scipy's own col_ind, not just an optimal one."""
    g = torch.Generator().manual_seed(100 + n)
    cases = [torch.zeros(n, n), torch.randint(0, 3, (n, n), generator=g).float()]
    C = torch.rand(n, n, generator=g)
    if n >= 5:
        dead_c = torch.randperm(n, generator=g)[: max(2, n // 4)]
        dead_r = torch.randperm(n, generator=g)[: max(2, n // 5)]
        C2 = C.clone()
        C2[:, dead_c] = C2[:, dead_c[:1]]   # identical columns (dead units of model 1)
        C2[dead_r, :] = C2[dead_r[:1], :]   # identical rows (dead units of model 0)
        cases.append(C2)
    cases.append(C)
    for Cx in cases:
        for mode in (True, 'replay'):  # adaptive (certificate or replay) and the replay kernel itself
            col, ref = _check_lap(ops, Cx, scipy_order=mode)
            assert (col == ref).all()


@pytest.mark.parametrize('n', [40, 384, 1500])
def test_lap_scipy_order_takes_the_certified_fast_path_on_planted_instances(ops, n):
    """A planted matching with noise (trained / permuted models): the dual initialisation matches nearly every row,
    the certificate finds no tight edge outside the matching, and the answer -- scipy's own -- comes from the fast
    solver.  Duplicating a column afterwards makes the optimum ambiguous: the certificate must see it and the
    replay kernel must run."""
    g = torch.Generator().manual_seed(7 + n)
    A = torch.randn(600, n, generator=g)
    perm = torch.randperm(n, generator=g)
    B = A[:, perm] + 0.2 * torch.randn(600, n, generator=g)
    C = torch.cdist(A.T, B.T)
    col, ref = _check_lap(ops, C, scipy_order=True)
    assert ops.lap_solve.last_path == 'certified'
    assert (col == ref).all()
    colr, _ = _check_lap(ops, C, scipy_order='replay')
    assert (colr == col).all()
    C2 = C.clone()
    C2[:, 1] = C2[:, 0]
    C2[2, :] = C2[3, :]
    col2, ref2 = _check_lap(ops, C2, scipy_order=True)
    assert ops.lap_solve.last_path == 'replay'
    assert (col2 == ref2).all()
    # a unique optimum the initialisation does not nearly solve, no two rows / columns alike (unrelated trained models):
    # the fast solver (augmenting row reduction + searches) and the certificate, not the replay kernel
    Cr = torch.rand(n, n, generator=g)
    col3, ref3 = _check_lap(ops, Cr, scipy_order=True)
    assert (col3 == ref3).all()
    assert ops.lap_solve.last_path == 'certified'
    # the same with a block of small integers mixed in: ties without duplicated lines fail the certificate -> replay
    Ct = Cr.clone()
    Ct[: n // 2, : n // 2] = torch.randint(0, 3, (n // 2, n // 2), generator=g).float()
    col4, ref4 = _check_lap(ops, Ct, scipy_order=True)
    assert (col4 == ref4).all()


def test_lap_cycle_certificate_accepts_ties_without_a_cycle_and_rejects_an_alternating_cycle(ops):
    """nif_lap_certify_unique_f32: the augmenting row reduction leaves tight edges outside the matching by construction
    (the strict certificate counts them), yet a random instance has a unique optimum and the cycle test says so; an
    exactly ambiguous 2 x 2 block (c00 + c11 == c01 + c10) with no duplicated line must be refused."""
    from neuron_interpolation_model_fusion_amd import _lib
    lib = _lib.lib
    g = torch.Generator().manual_seed(5)
    n = 200
    st = torch.cuda.current_stream().cuda_stream

    def verdicts(C):
        col, row, u, v = ops.lap_solve(C, duals=True)
        res = torch.full((1,), -1, dtype=torch.int32, device='cuda')
        strict = torch.empty(1, dtype=torch.int32, device='cuda')
        nb = lib.nif_lap_certify_workspace_bytes(n)
        ws = torch.empty(nb, dtype=torch.uint8, device='cuda')
        _lib.check(lib.nif_lap_certify_unique_f32(C.data_ptr(), C.stride(0), n, col.data_ptr(), row.data_ptr(), u.data_ptr(),
                                                  v.data_ptr(), 1e-9, res.data_ptr(), ws.data_ptr(), nb, st), 'certify')
        _lib.check(lib.nif_lap_count_tight_f32(C.data_ptr(), C.stride(0), n, col.data_ptr(), u.data_ptr(), v.data_ptr(), 1e-9,
                                               strict.data_ptr(), st), 'count_tight')
        return col, int(res.item()), int(strict.item())

    C = torch.rand(n, n, generator=g).cuda()
    col, res, strict = verdicts(C)
    ref = linear_sum_assignment(C.cpu().numpy())[1]
    assert (col.cpu().numpy() == ref).all()
    assert strict > 0, 'the solver leaves tight edges behind (a bidder\'s best and second-best columns tie)'
    assert res == 0, 'but they close no cycle: the optimum is unique'
    Ca = (torch.rand(n, n, generator=g) + 10.0)
    Ca[:2, :] = 100.0
    Ca[:, :2] = 100.0
    Ca[:2, :2] = torch.tensor([[1.0, 2.0], [3.0, 4.0]])   # both matchings of the block cost 5
    Ca = Ca.cuda()
    col, res, _ = verdicts(Ca)
    assert sorted(col[:2].tolist()) == [0, 1]
    assert res != 0


def test_lap_certificate_counts_tight_edges(ops):
    from neuron_interpolation_model_fusion_amd import _lib
    g = torch.Generator().manual_seed(11)
    n = 96
    C = torch.rand(n, n, generator=g).cuda()
    col, row, u, v = ops.lap_solve(C, duals=True)
    cnt = torch.empty(1, dtype=torch.int32, device='cuda')

    def count(Cm, colm):
        _lib.check(_lib.lib.nif_lap_count_tight_f32(Cm.data_ptr(), Cm.stride(0), n, colm.data_ptr(), u.data_ptr(), v.data_ptr(),
                                                    1e-9, cnt.data_ptr(), torch.cuda.current_stream().cuda_stream),
                   'nif_lap_count_tight_f32')
        return int(cnt.item())

    slack = (C.double() - u[:, None] - v[None, :])
    assert slack[torch.arange(n), col.long()].abs().max() < 1e-14
    slack[torch.arange(n), col.long()] = 1.0
    expected = int((slack <= 1e-9 * (C.double().abs() + u.abs()[:, None] + v.abs()[None, :])).sum())
    assert count(C, col) == expected
    # a matching that is not on tight edges is flagged
    wrong = col.roll(1)
    assert count(C, wrong) > expected


@pytest.mark.parametrize('shape', [(37, 10), (5, 7, 3, 3), (4, 16, 17), (64, 384, 65), (256, 64, 8, 8), (1, 1)])
@pytest.mark.parametrize('weighted', [False, True])
def test_weighted_mse_forward_and_gradient(ops, shape, weighted):
    """Fused loss of the ALF fit against the reference's expression (alf.py:420-421) under autograd, in f64."""
    g = torch.Generator().manual_seed(sum(shape))
    pred = torch.randn(*shape, generator=g)
    truth = torch.randn(*shape, generator=g)
    w = (torch.rand(shape[1], generator=g) + 0.1) if weighted else None

    p64 = pred.double().requires_grad_(True)
    wb = torch.ones(shape[1], dtype=torch.float64) if w is None else w.double()
    wb = wb.reshape([1, -1] + [1] * (len(shape) - 2))
    ref = (((p64 - truth.double()) ** 2) * wb).mean()
    (ref * 3.0).backward()

    pg = pred.cuda().requires_grad_(True)
    loss = ops.weighted_mse(pg, truth.cuda(), None if w is None else w.cuda())
    (loss * 3.0).backward()
    assert loss.dtype == torch.float32 and loss.dim() == 0
    np.testing.assert_allclose(loss.item(), ref.item(), rtol=2e-6)
    np.testing.assert_allclose(pg.grad.cpu().numpy(), p64.grad.float().numpy(), rtol=2e-6, atol=1e-9)
    # the flattened (rows, channels) view the reference feeds its criterion gives the same value
    if len(shape) > 2:
        flat = pred.movedim(1, -1).reshape(-1, shape[1]).contiguous().cuda()
        flat_t = truth.movedim(1, -1).reshape(-1, shape[1]).contiguous().cuda()
        loss_flat = ops.weighted_mse(flat, flat_t, None if w is None else w.cuda())
        np.testing.assert_allclose(loss_flat.item(), ref.item(), rtol=2e-6)


def _check_transport_certificate(C, a, b, plan, u, v):
    """Primal feasibility + dual feasibility + complementary slackness = optimality, no reference solver needed."""
    T = plan.dense(torch.float64).cpu()
    C = C.double().cpu()
    a, b, u, v = a.double().cpu(), b.double().cpu(), u.cpu(), v.cpu()
    b = b * a.sum() / b.sum()
    assert (T >= 0).all()
    assert (T.sum(1) - a).abs().max() <= 1e-11 * a.sum() + 1e-15
    assert (T.sum(0) - b).abs().max() <= 1e-11 * a.sum() + 1e-15
    slack = C - u[:, None] - v[None, :]
    scale = float(C.abs().max()) + 1e-30
    assert slack.min() >= -1e-9 * scale, 'dual infeasible'
    assert (slack[T > 0].abs().max() if (T > 0).any() else 0.0) <= 1e-9 * scale, 'support not tight'
    return float((T * C).sum())


@pytest.mark.parametrize('n0,n1', [(1, 1), (1, 7), (9, 1), (5, 5), (17, 30), (64, 64), (130, 90), (384, 384), (700, 1100),
                                   (2100, 1500)])
def test_emd_optimality_certificate(ops, n0, n1):
    g = torch.Generator().manual_seed(n0 * 31 + n1)
    C = torch.cdist(torch.randn(n0, 12, generator=g), torch.randn(n1, 12, generator=g))
    a = torch.rand(n0, generator=g) + 0.05
    b = torch.rand(n1, generator=g) + 0.05
    a, b = a / a.sum(), b / b.sum()
    plan, u, v = ops.emd_solve(C.cuda(), a.cuda(), b.cuda(), duals=True)
    _check_transport_certificate(C, a, b, plan, u, v)
    assert plan.nnz <= n0 + n1 - 1


@pytest.mark.parametrize('n0,n1', [(1, 1), (3, 70), (64, 64), (130, 90), (257, 129), (384, 384), (640, 1000), (900, 2048),
                                   (300, 2500), (2300, 2000)])   # the last: the forest kernel with its per-tree distances and demands in the workspace
def test_emd_kernels_agree(ops, n0, n1):
    """The rooted-forest kernel (default where it fits: n1 <= 2048), the same with every augmenting path walked by its
    general one-lane code, and the first-generation arc-pool kernel, from the cold and from the assignment start: each
    passes the optimality certificate and all reach the same optimal value."""
    from neuron_interpolation_model_fusion_amd import _lib
    g = torch.Generator().manual_seed(n0 * 17 + n1)
    C = torch.cdist(torch.randn(n0, 6, generator=g), torch.randn(n1, 6, generator=g))
    a = torch.rand(n0, generator=g) ** 3 + 0.01   # skewed masses: many arcs run empty, trees are cut all the time
    b = torch.rand(n1, generator=g) ** 3 + 0.01
    a, b = a / a.sum(), b / b.sum()
    vals = []
    for flags in (0, _lib.NIF_EMD_ONE_LANE_PATHS, _lib.NIF_EMD_ARC_POOL, _lib.NIF_EMD_COLD_START,
                  _lib.NIF_EMD_COLD_START | _lib.NIF_EMD_ONE_LANE_PATHS):
        plan, u, v = ops.emd_solve(C.cuda(), a.cuda(), b.cuda(), duals=True, flags=flags)
        vals.append(_check_transport_certificate(C, a, b, plan, u, v))
        assert plan.nnz <= n0 + n1 - 1
        # CSC with ascending rows inside a column
        cp, ri = plan.colptr.cpu().numpy(), plan.rowidx.cpu().numpy()
        assert cp[0] == 0 and cp[-1] == plan.nnz and (np.diff(cp) >= 0).all()
    np.testing.assert_allclose(vals, vals[0], rtol=1e-12)


def test_emd_many_small_random_problems_reach_the_lp_value(ops):
    """Forty small problems of every flavour the kernels branch on -- square / wide / tall, continuous and integer
    (massively tied) costs, uniform, skewed and partly zero marginals -- through the forest kernel, its one-lane path
    code and the arc-pool kernel: certificate each time, and the optimal value of the LP oracle (HiGHS)."""
    from oracle import otfusion as o
    from neuron_interpolation_model_fusion_amd import _lib
    g = torch.Generator().manual_seed(2024)
    for case in range(40):
        n0 = int(torch.randint(1, 41, (1,), generator=g))
        n1 = n0 if case % 3 == 0 else int(torch.randint(1, 41, (1,), generator=g))
        if case % 4 == 1:
            C = torch.randint(0, 5, (n0, n1), generator=g).float()
        else:
            C = torch.cdist(torch.randn(n0, 3, generator=g), torch.randn(n1, 3, generator=g))
        if case % 5 == 2:
            a, b = torch.full((n0,), 1.0 / n0), torch.full((n1,), 1.0 / n1)
        else:
            a = torch.rand(n0, generator=g) ** 4 + (0.0 if case % 5 == 3 else 0.01)
            b = torch.rand(n1, generator=g) ** 4 + (0.0 if case % 5 == 3 else 0.01)
            if case % 5 == 3 and n0 > 2 and n1 > 2:
                a[::3] = 0.0
                b[1::4] = 0.0
            a, b = a / a.sum(), b / b.sum()
        ref = o.emd(a.double().numpy(), b.double().numpy(), C.numpy())
        ref_val = float((ref.double() * C.double()).sum())
        for flags in (0, _lib.NIF_EMD_ONE_LANE_PATHS, _lib.NIF_EMD_ARC_POOL, _lib.NIF_EMD_COLD_START):
            plan, u, v = ops.emd_solve(C.cuda(), a.cuda(), b.cuda(), duals=True, flags=flags)
            val = _check_transport_certificate(C, a, b, plan, u, v)
            np.testing.assert_allclose(val, ref_val, rtol=1e-6, atol=1e-9, err_msg=f'case {case} ({n0} x {n1}) flags {flags}')


def test_emd_forest_kernel_is_deterministic(ops):
    g = torch.Generator().manual_seed(11)
    C = torch.randint(0, 6, (200, 200), generator=g).float()   # integer costs: ties everywhere
    a = torch.rand(200, generator=g) + 0.05
    b = torch.rand(200, generator=g) + 0.05
    runs = [ops.emd_solve(C.cuda(), (a / a.sum()).cuda(), (b / b.sum()).cuda(), duals=True) for _ in range(3)]
    for plan, u, v in runs[1:]:
        assert torch.equal(plan.colptr, runs[0][0].colptr) and torch.equal(plan.rowidx, runs[0][0].rowidx)
        assert torch.equal(plan.val, runs[0][0].val) and torch.equal(u, runs[0][1]) and torch.equal(v, runs[0][2])
    _check_transport_certificate(C, a / a.sum(), b / b.sum(), *runs[0])


@pytest.mark.parametrize('n0,n1', [(6, 6), (12, 20), (60, 45), (150, 200)])
def test_emd_plan_matches_the_lp_oracle(ops, n0, n1):
    from oracle import otfusion as o
    g = torch.Generator().manual_seed(n0 + 1000 * n1)
    C = torch.cdist(torch.randn(n0, 8, generator=g), torch.randn(n1, 8, generator=g))
    a = torch.rand(n0, generator=g) + 0.1
    b = torch.rand(n1, generator=g) + 0.1
    a, b = (a / a.sum()).double(), (b / b.sum()).double()
    ref = o.emd(a.numpy(), b.numpy(), C.numpy())
    plan = ops.emd_solve(C.cuda(), a.cuda(), b.cuda())
    got = plan.dense().cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=3e-7, rtol=1e-4)  # HiGHS' own feasibility tolerance is 1e-7


def test_emd_solve_many_equals_one_by_one(ops):
    """Independent transport problems solved side by side (one stream each) give the plans of emd_solve."""
    g = torch.Generator().manual_seed(5)
    probs = []
    for n0, n1 in [(17, 17), (40, 33), (64, 64), (25, 90), (128, 128)]:
        C = torch.rand(n0, n1, generator=g).cuda()
        a = (torch.rand(n0, generator=g) + 0.1).double()
        b = (torch.rand(n1, generator=g) + 0.1).double()
        probs.append((C, (a / a.sum()).cuda(), (b / b.sum()).cuda()))
    many = ops.emd_solve_many(probs, duals=True)
    for (C, a, b), (plan, u, v) in zip(probs, many):
        one, u1, v1 = ops.emd_solve(C, a, b, duals=True)
        assert torch.equal(plan.colptr, one.colptr) and torch.equal(plan.rowidx, one.rowidx)
        assert torch.equal(plan.val, one.val) and torch.equal(u, u1) and torch.equal(v, v1)


def test_emd_solve_many_batches_square_problems_of_one_size(ops):
    """Square problems of one size take nif_emd_solve_batch_f32 (assignment start and forest kernel with the problems
    along gridDim.z); plans, duals and work counters are those of one nif_emd_solve_f32 call per problem, bit for bit.
    Mixed in: sizes that occur once and a rectangular problem (their own streams), planted costs, a zero-mass row."""
    g = torch.Generator().manual_seed(11)
    probs = []
    for n0, n1 in [(64, 64), (130, 130), (64, 64), (40, 33), (384, 384), (64, 64), (130, 130), (384, 384), (2, 2), (2, 2),
                   (17, 17), (600, 600), (600, 600)]:
        C = torch.rand(n0, n1, generator=g)
        if len(probs) % 3 == 1 and n0 == n1:   # a planted matching: most of the mass stays on the assignment start
            C = C + 1.0
            C[torch.arange(n0), torch.randperm(n0, generator=g)] = 0.05
        a = (torch.rand(n0, generator=g) + 0.1).double()
        b = (torch.rand(n1, generator=g) + 0.1).double()
        if len(probs) == 2:
            a[5] = 0.0
        probs.append((C.cuda(), (a / a.sum()).cuda(), (b / b.sum()).cuda()))
    many = ops.emd_solve_many(probs, duals=True)
    works = list(ops.emd_solve_many.last_work)
    for k, ((C, a, b), (plan, u, v)) in enumerate(zip(probs, many)):
        one, u1, v1 = ops.emd_solve(C, a, b, duals=True)
        assert ops.emd_solve.last_work == works[k], k
        assert torch.equal(plan.colptr, one.colptr) and torch.equal(plan.rowidx, one.rowidx), k
        assert torch.equal(plan.val, one.val) and torch.equal(u, u1) and torch.equal(v, v1), k
        _check_transport_certificate(C, a, b, plan, u, v)
    # the C entry by itself: argument checks and what it declines
    lib, _lib = ops.lib, ops._lib
    n, B = 64, 2
    z = torch.zeros(8, device='cuda', dtype=torch.float64)
    assert lib.nif_emd_batch_workspace_bytes(n, B) > B * lib.nif_emd_workspace_bytes(n, n)
    assert lib.nif_emd_batch_workspace_bytes(0, B) == 0
    args = lambda **kw: [kw.get('C', z.data_ptr()), n, kw.get('stride', n * n), kw.get('n', n), B, z.data_ptr(), z.data_ptr(),
                         kw.get('tol', z.data_ptr()), kw.get('flags', 0), z.data_ptr(), z.data_ptr(), z.data_ptr(), 200,
                         z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), kw.get('wsz', 1 << 30), None]
    assert lib.nif_emd_solve_batch_f32(*args(tol=None)) == -1
    assert lib.nif_emd_solve_batch_f32(*args(stride=n * n - 1)) == -1
    assert lib.nif_emd_solve_batch_f32(*args(flags=_lib.NIF_EMD_ARC_POOL)) == _lib.NIF_ERR_UNSUPPORTED
    assert lib.nif_emd_solve_batch_f32(*args(wsz=1024)) == _lib.NIF_ERR_WORKSPACE


@pytest.mark.parametrize('n0,n1,reg', [(17, 17, 0.1), (64, 48, 0.05), (150, 190, 0.06), (192, 192, 0.05), (384, 384, 0.05),
                                        (300, 771, 0.08), (1024, 1024, 0.02)])   # up to 192 x 192: the one-workgroup kernel
def test_sinkhorn_matches_the_f64_restatement(ops, n0, n1, reg):
    """The additional Sinkhorn mode against oracle.otfusion.sinkhorn_knopp (POT's algorithm restated, parity unpinned):
    same iteration count (the stopping rule is evaluated on the same iterations), plan and marginals."""
    from oracle.otfusion import sinkhorn_knopp
    g = torch.Generator().manual_seed(n0 * 7 + n1)
    C = torch.rand(n0, n1, generator=g)
    a = torch.rand(n0, generator=g).double() + 0.1
    b = torch.rand(n1, generator=g).double() + 0.1
    a, b = a / a.sum(), b / b.sum()
    res = ops.sinkhorn(C.cuda(), a.cuda(), b.cuda(), reg)
    T_ref, iters = sinkhorn_knopp(a.numpy(), b.numpy(), C.numpy(), reg)
    assert res['state'] == 1 and res['iterations'] == iters
    T = res['T'].cpu()
    np.testing.assert_allclose(T.numpy(), T_ref.numpy(), rtol=2e-5, atol=1e-12)
    np.testing.assert_allclose(T.double().sum(0).numpy(), b.numpy(), rtol=1e-5)
    np.testing.assert_allclose(T.double().sum(1).numpy(), a.numpy(), rtol=1e-4)


def test_sinkhorn_iteration_cap_and_breakdown(ops):
    g = torch.Generator().manual_seed(3)
    C = torch.rand(40, 40, generator=g).cuda()
    a = torch.full((40,), 1 / 40, dtype=torch.float64).cuda()
    res = ops.sinkhorn(C, a, a, 0.01, max_iter=7)
    assert res['state'] == 0 and res['iterations'] == 7
    # a regularisation so small that exp(-C / reg) underflows to zero everywhere: POT stops with a warning and
    # returns the initial scalings; so does the kernel (state 2, no iteration completed, finite plan)
    res = ops.sinkhorn(C + 1.0, a, a, 1e-4)
    assert res['state'] == 2 and res['iterations'] == 0 and bool(torch.isfinite(res['T']).all())


def test_emd_degenerate_and_uniform_cases(ops):
    """Ties / degenerate vertices: the plan need not be unique, the optimal value is."""
    g = torch.Generator().manual_seed(5)
    # uniform square = assignment: same value as the LAP kernel
    n = 96
    C = torch.rand(n, n, generator=g)
    a = torch.full((n,), 1.0 / n)
    plan, u, v = ops.emd_solve(C.cuda(), a.cuda(), a.cuda(), duals=True)
    val = _check_transport_certificate(C, a, a, plan, u, v)
    r, c = linear_sum_assignment(C.double().numpy())
    np.testing.assert_allclose(val, C.double().numpy()[r, c].sum() / n, rtol=1e-6)  # a = f32(1/n)
    # uniform rectangular with integer costs: massively degenerate
    C = torch.randint(0, 4, (8, 12), generator=g).float()
    a, b = torch.full((8,), 1 / 8), torch.full((12,), 1 / 12)
    plan, u, v = ops.emd_solve(C.cuda(), a.cuda(), b.cuda(), duals=True)
    _check_transport_certificate(C, a, b, plan, u, v)
    # importance scores at the reference's floor (1e-8 before normalisation) on some neurons
    a = torch.rand(40, generator=g)
    a[::5] = 1e-8
    b = torch.rand(40, generator=g)
    b[3::7] = 1e-8
    a, b = a / a.sum(), b / b.sum()
    C = torch.cdist(torch.randn(40, 5, generator=g), torch.randn(40, 5, generator=g))
    plan, u, v = ops.emd_solve(C.cuda(), a.cuda(), b.cuda(), duals=True)
    _check_transport_certificate(C, a, b, plan, u, v)
    # exactly zero mass on some neurons (rescale_min_importance = 0 maps the least important neuron to 0)
    a = torch.rand(30, generator=g)
    b = torch.rand(45, generator=g)
    a[::6] = 0.0
    b[2::9] = 0.0
    a, b = a / a.sum(), b / b.sum()
    C = torch.cdist(torch.randn(30, 4, generator=g), torch.randn(45, 4, generator=g))
    plan, u, v = ops.emd_solve(C.cuda(), a.cuda(), b.cuda(), duals=True)
    _check_transport_certificate(C, a, b, plan, u, v)
    T = plan.dense(torch.float64).cpu()
    assert float(T[::6].abs().max()) == 0.0 and float(T[:, 2::9].abs().max()) == 0.0


def test_refit_sparse_matches_dense_products(ops):
    g = torch.Generator().manual_seed(9)
    n_k, n_f, in_k, in_f, sp = 13, 11, 7, 9, 3
    T = (torch.rand(n_k, n_f, generator=g) < 0.3).double() * torch.rand(n_k, n_f, generator=g).double()
    Tp = (torch.rand(in_k, in_f, generator=g) < 0.4).double() * torch.rand(in_k, in_f, generator=g).double()

    def csc(M):
        rows, cols = [], []
        ptr = [0]
        vals = []
        for j in range(M.shape[1]):
            nz = torch.nonzero(M[:, j]).reshape(-1)
            rows += nz.tolist()
            vals += M[nz, j].tolist()
            ptr.append(len(rows))
        return ops.SparsePlan(M.shape[0], M.shape[1], torch.tensor(ptr, dtype=torch.int32).cuda(),
                              torch.tensor(rows, dtype=torch.int32).cuda(), torch.tensor(vals, dtype=torch.float64).cuda())
    W1 = torch.randn(n_k, in_k * sp, generator=g)
    W0 = torch.randn(n_f, in_f * sp, generator=g)
    mu = torch.rand(n_k, generator=g).double() + 0.5
    nu_prev = torch.rand(in_f, generator=g).double() + 0.5
    # dense reference: T^T diag(1/mu) W1 [per position] T_prev diag(1/nu_prev)
    Wk = W1.double().reshape(n_k, in_k, sp)
    edges = torch.einsum('ips,pq->iqs', Wk, Tp / nu_prev[None, :])
    ref = 0.25 * W0.double() + 0.5 * torch.einsum('ij,iqs->jqs', T / mu[:, None], edges).reshape(n_f, -1)
    got = ops.refit_sparse(W0.cuda(), W1.cuda(), rows=csc(T).scaled(row_scale=(1 / mu).cuda()),
                           cols=csc(Tp).scaled(col_scale=(1 / nu_prev).cuda()), alpha=0.5, beta=0.25, sp=sp)
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=2e-5, atol=2e-6)
    # identity on either side
    got = ops.refit_sparse(None, W1.cuda(), rows=None, cols=csc(Tp).scaled(), alpha=1.0, sp=sp)
    np.testing.assert_allclose(got.cpu().numpy(), torch.einsum('ips,pq->iqs', Wk, Tp).reshape(n_k, -1).numpy(),
                               rtol=2e-5, atol=2e-6)
    got = ops.refit_sparse(None, W1.cuda(), rows=csc(T).scaled(), cols=None, alpha=1.0, sp=sp)
    np.testing.assert_allclose(got.cpu().numpy(), (T.T @ W1.double()).numpy(), rtol=2e-5, atol=2e-6)


def _kmeans_problem(N, k, m, seed, empty=()):
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(m, N, generator=g)
    s = torch.rand(N, generator=g) + 0.2
    assign = torch.randint(0, k, (N,), generator=g)
    for c in empty:  # leave these labels without members
        assign[assign == c] = (c + 1) % k
    return A, s, assign.int()


def _ref_lloyd(A, s, assign, k):
    """Gram-form Lloyd step in f64 on the CPU: compacted centroids, nearest centroid (lowest label on ties)."""
    A64, s64 = A.double(), s.double()
    labels = [c for c in range(k) if (assign == c).any()]
    cents = torch.stack([(A64[:, assign == c] * s64[assign == c]).sum(1) / s64[assign == c].sum() for c in labels], 1)
    d = torch.cdist(cents.T.contiguous(), A64.T.contiguous())  # k' x N
    nearest, arg = d.min(0)
    return arg, nearest, float((nearest * s64).sum()), len(labels), cents


@pytest.mark.parametrize('N,k,m,empty', [(50, 7, 40, ()), (300, 64, 64, (3, 10)), (1000, 130, 128, (0,)), (2500, 16, 96, ()),
                                         (5000, 40, 32, (7,))])
def test_kmeans_lloyd_step_matches_f64_reference(ops, N, k, m, empty):
    A, s, assign = _kmeans_problem(N, k, m, N + k, empty)
    G, _, _ = ops.xty(A.double().cuda(), None, acc='f64', norms=False)
    km = ops.KMeans(G, torch.diagonal(G).contiguous(), s.cuda(), k)
    out, nearest, cost, kp = km.lloyd_step(assign.cuda())
    # f32 storage of G (the fusions' default): same labels, distances within the rounding of G
    km32 = ops.KMeans(G.float(), torch.diagonal(G).contiguous(), s.cuda(), k, gdiag=torch.diagonal(G))
    out32, nearest32, cost32, kp32 = km32.lloyd_step(assign.cuda())
    assert torch.equal(out32, out) and int(kp32) == int(kp)
    np.testing.assert_allclose(nearest32.cpu().numpy(), nearest.cpu().numpy(), rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(float(cost32), float(cost), rtol=1e-6)
    arg, near, c, kprime, _ = _ref_lloyd(A, s, assign, k)
    assert int(kp) == kprime
    # ties within rounding aside (none in random data), the labels are exact
    assert (out.cpu().long() == arg).all()
    # Gram form: a neuron that is its own centroid gets sqrt(cancellation residue) ~ 1e-7 * |x| instead of 0
    np.testing.assert_allclose(nearest.cpu().numpy(), near.numpy(), rtol=1e-7, atol=2e-6)
    np.testing.assert_allclose(float(cost), c, rtol=1e-9)


@pytest.mark.parametrize('N,k,cuts', [(1000, 64, (0, 5, 16)), (6000, 40, (0, 1, 50, 50, 94)), (130, 9, (0, 3)), (40, 5, (0, 0, 1))])
def test_kmeans_lloyd_step_in_block_ranges_equals_the_whole_step(ops, N, k, cuts):
    """nif_kmeans_lloyd_step_range over consecutive 64-neuron block ranges (what the ranks of a neuron-sharded step
    each do) + nif_kmeans_cost_from_partials = nif_kmeans_lloyd_step, bit for bit; N = 6000 carries a big cluster."""
    import ctypes
    from neuron_interpolation_model_fusion_amd._lib import lib, check
    A, s, assign = _kmeans_problem(N, k, 24, 3 * N + k)
    if N == 6000:
        assign[torch.randperm(N, generator=torch.Generator().manual_seed(1))[:2500]] = 2
    G, _, _ = ops.xty(A.double().cuda(), None, acc='f64', norms=False)
    gd = torch.diagonal(G).contiguous()
    nb = (N + 63) // 64
    assert cuts[-1] == nb
    for Gs, sfx in ((G, ''), (G.float(), '_gf32')):
        km = ops.KMeans(Gs, gd, s.cuda(), k, gdiag=gd)
        out, nearest, cost, kp = km.lloyd_step(assign.cuda())
        a = assign.cuda()
        out_r = torch.zeros(N, dtype=torch.int32, device='cuda')
        near_r = torch.zeros(N, dtype=torch.float64, device='cuda')
        part = torch.zeros(nb, dtype=torch.float64, device='cuda')
        cost_r = torch.zeros((), dtype=torch.float64, device='cuda')
        kp_r = torch.zeros((), dtype=torch.int32, device='cuda')
        g = (gd.data_ptr(),) if sfx else ()
        fn = getattr(lib, 'nif_kmeans_lloyd_step_range' + sfx)
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            check(fn(Gs.data_ptr(), Gs.stride(0), *g, km.nn.data_ptr(), km.s.data_ptr(), a.data_ptr(), N, k, None, 1, lo, hi,
                     out_r.data_ptr(), near_r.data_ptr(), part.data_ptr(), kp_r.data_ptr(), km.ws.data_ptr(), km.nbytes,
                     torch.cuda.current_stream().cuda_stream), 'range')
        check(lib.nif_kmeans_cost_from_partials(part.data_ptr(), nb, None, 1, cost_r.data_ptr(),
                                                torch.cuda.current_stream().cuda_stream), 'cost')
        assert torch.equal(out_r, out) and torch.equal(near_r, nearest) and int(kp_r) == int(kp)
        assert float(cost_r) == float(cost)
        # a range outside the layer is refused
        rc = fn(Gs.data_ptr(), Gs.stride(0), *g, km.nn.data_ptr(), km.s.data_ptr(), a.data_ptr(), N, k, None, 1, 0, nb + 1,
                out_r.data_ptr(), near_r.data_ptr(), part.data_ptr(), kp_r.data_ptr(), km.ws.data_ptr(), km.nbytes,
                torch.cuda.current_stream().cuda_stream)
        assert rc != 0


@pytest.mark.parametrize('N,k,m,sizes', [(6000, 40, 48, (3000,)), (9000, 64, 32, (1025, 4100, 1024)), (5000, 3, 24, (2500, 2400)),
                                          (3000, 8, 16, (2993,))])
def test_kmeans_lloyd_step_clusters_above_the_chunking_threshold(ops, N, k, m, sizes):
    """Clusters of more than 1024 members take the chunked summation order (csrc/nif_kmeans.hip KM_BIG): same labels
    and distances as the f64 reference, the same step twice gives the same bits, and a batch of tries equals one by one."""
    A, s, assign = _kmeans_problem(N, k, m, N + 11 * k)
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(N))
    at = 0
    assert len(sizes) < k
    for c, n_c in enumerate(sizes):  # label c + 1 takes n_c neurons, scattered over the layer
        assign[perm[at:at + n_c]] = c + 1
        at += n_c
    assert 0 <= int(assign.min()) and int(assign.max()) < k
    G, _, _ = ops.xty(A.double().cuda(), None, acc='f64', norms=False)
    gd = torch.diagonal(G).contiguous()
    arg, near, c_ref, kprime, _ = _ref_lloyd(A, s, assign, k)
    for Gs in (G, G.float()):
        km = ops.KMeans(Gs, gd, s.cuda(), k, gdiag=gd)
        out, nearest, cost, kp = km.lloyd_step(assign.cuda())
        out, nearest, cost = out.clone(), nearest.clone(), float(cost)
        assert int(kp) == kprime
        assert (out.cpu().long() == arg).all()
        tol = dict(rtol=1e-7, atol=2e-6) if Gs.dtype == torch.float64 else dict(rtol=4e-6, atol=4e-5)
        np.testing.assert_allclose(nearest.cpu().numpy(), near.numpy(), **tol)
        np.testing.assert_allclose(cost, c_ref, rtol=1e-9 if Gs.dtype == torch.float64 else 2e-6)
        out2, nearest2, cost2, _ = km.lloyd_step(assign.cuda())
        assert torch.equal(out2, out) and torch.equal(nearest2, nearest) and float(cost2) == cost
        # two tries in one launch: the skewed assignment beside a balanced one
        other = _kmeans_problem(N, k, m, 5)[2]
        kmb = ops.KMeans(Gs, gd, s.cuda(), k, gdiag=gd, tries=2)
        both = torch.stack([assign, other]).cuda()
        res = kmb.lloyd_step_many(both, [0, 1])
        assert torch.equal(res[0][0], out) and torch.equal(res[1][0], nearest) and float(res[2][0]) == cost
        o1 = km.lloyd_step(other.cuda())
        assert torch.equal(res[0][1], o1[0]) and torch.equal(res[1][1], o1[1])


@pytest.mark.parametrize('N,k,m', [(50, 7, 33), (771, 100, 20), (4096, 512, 70), (20000, 96, 9), (40001, 12, 5)])
def test_kmeans_targets_all_staging_variants(ops, N, k, m):
    """LDS-staged rows (several rows per block, one wide row per block, unaligned width) and the gather fallback."""
    A, s, assign = _kmeans_problem(N, k, m, N * 3 + k, empty=(1,) if k > 2 else ())
    km = ops.KMeans(None, torch.zeros(N, dtype=torch.float64).cuda(), s.cuda(), k)
    T, kp = km.targets(A.cuda(), assign.cuda())
    _, _, _, kprime, cents = _ref_lloyd(A, s, assign, k)
    assert int(kp) == kprime and T.shape == (m, k)
    np.testing.assert_allclose(T[:, :kprime].cpu().numpy(), cents.float().numpy(), rtol=1e-5, atol=1e-6)
    assert float(T[:, kprime:].abs().max()) == 0.0 if kprime < k else True


def _targets_sequential(A, s, assign, k):
    """The reference's weighted centroid rows as the library defines their rounding: members in ascending neuron
    order, f64 products (exact) added one by one, one f64 division, one rounding to f32."""
    A64, s64 = A.double(), s.double()
    labels = [c for c in range(k) if (assign == c).any()]
    members = [torch.nonzero(assign == c).flatten() for c in labels]
    acc = torch.zeros(A.shape[0], len(labels), dtype=torch.float64)
    S = torch.zeros(len(labels), dtype=torch.float64)
    for t in range(max(len(mb) for mb in members)):
        cols = [j for j, mb in enumerate(members) if len(mb) > t]
        idx = torch.stack([members[j][t] for j in cols])
        acc[:, cols] = acc[:, cols] + A64[:, idx] * s64[idx]
        S[cols] = S[cols] + s64[idx]
    return (acc / S).float()


@pytest.mark.parametrize('N,k,m', [
    (768, 384, 37),      # C4 shape: 16 rows per tile, one cluster per lane, ragged last tile
    (4096, 2048, 19),    # two clusters per lane
    (8192, 4000, 7),     # four clusters per lane
    (24576, 3072, 5),    # C5 shape: one 96 KB row per tile, weights only partly in LDS
    (20000, 96, 9),      # clusters of ~200 members, some reaching past the LDS-held weights
    (30000, 700, 3),     # 120 KB rows
    (6144, 768, 11),
    (772, 5000, 3),      # more labels than the wide-load kernel takes: the simple kernel
])
def test_kmeans_targets_bits_equal_sequential_f64_sums(ops, N, k, m):
    """Every staging variant of nif_kmeans_targets must give the SAME bits: summation order and arithmetic are fixed."""
    A, s, assign = _kmeans_problem(N, k, m, N + 7 * k, empty=(1, 5) if k > 8 else ())
    if k > N:
        assign = (assign % 60).int()
    km = ops.KMeans(None, torch.zeros(N, dtype=torch.float64).cuda(), s.cuda(), k)
    T, kp = km.targets(A.cuda(), assign.cuda())
    want = _targets_sequential(A, s, assign, k)
    assert int(kp) == want.shape[1]
    assert torch.equal(T[:, :int(kp)].cpu(), want)
    assert float(T[:, int(kp):].abs().max()) == 0.0


@pytest.mark.parametrize('gdtype', [torch.float64, torch.float32])
@pytest.mark.parametrize('N,k', [(64, 5), (300, 20), (3000, 33), (4097, 12), (6000, 40), (24576, 70), (40000, 6), (65536, 5), (70000, 4)])
def test_kmeans_farthest_init_matches_sequential_reference(ops, N, k, gdtype):
    g = torch.Generator().manual_seed(N)
    A = torch.randn(24, N, generator=g).double()
    P = A * 0.9  # "projected" vectors whose Gram feeds the picks; distances go to the raw vectors
    G = (P.T @ P).to(gdtype)
    G_dev = G.cuda()
    G = G.double()  # the kernels widen the stored entries and work in f64
    nn = (A * A).sum(0)
    # the kernel's distance: G[p][p] - 2 G[p][j] + nn[j]
    first = 3
    picks, mind, lab = [first], None, None
    for c in range(k):
        p = picks[c]
        d = (G[p, p] - 2 * G[p] + nn).clamp_min(0).sqrt()
        if c == 0:
            mind, lab = d.clone(), torch.zeros(N, dtype=torch.long)
        else:
            upd = d < mind
            mind[upd], lab[upd] = d[upd], c
        if c + 1 < k:
            picks.append(int(torch.argmax(mind)))  # first index of the maximum
    km = ops.KMeans(G_dev, nn.cuda(), torch.ones(N).cuda(), k)
    got_picks, got_assign = km.farthest_init(first)
    assert got_picks.cpu().tolist() == picks
    assert (got_assign.cpu().long() == lab).all()
    np.testing.assert_allclose(km.mind.cpu().numpy(), mind.numpy(), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize('gdtype', [torch.float64, torch.float32])
@pytest.mark.parametrize('N,k,T', [(300, 20, 10), (3000, 33, 16), (6000, 40, 10), (24576, 12, 3)])
def test_kmeans_farthest_init_many_equals_one_by_one(ops, N, k, T, gdtype):
    """The tries of one KF layer in one launch (one workgroup, or one row of workgroups, per try)."""
    g = torch.Generator().manual_seed(N + T)
    A = torch.randn(24, N, generator=g).double()
    G = ((0.9 * A).T @ (0.9 * A)).to(gdtype).cuda()
    nn = (A * A).sum(0).cuda()
    km = ops.KMeans(G, nn, torch.ones(N).cuda(), k)
    firsts = torch.randint(0, N, (T,), generator=g).tolist()
    picks, assign = km.farthest_init_many(firsts)
    for t, first in enumerate(firsts):
        p1, a1 = km.farthest_init(first)
        assert torch.equal(picks[t], p1) and torch.equal(assign[t], a1), (t, first)


@pytest.mark.parametrize('gdtype', [torch.float64, torch.float32])
def test_kmeans_farthest_init_exchange_timeout_is_redone_not_returned_partial(ops, gdtype, monkeypatch):
    """The multi-workgroup initialisation (N > 4096) with a timeout forced into its exchange
    (NIF_KMEANS_INIT_INJECT_TIMEOUT): every try falls to the one-workgroup kernel queued behind it inside the same
    call, the caller gets the complete picks / labels, and the device counter says what happened."""
    N, k, T = 6000, 40, 3
    g = torch.Generator().manual_seed(5)
    A = torch.randn(24, N, generator=g).double()
    G = ((0.9 * A).T @ (0.9 * A)).to(gdtype).cuda()
    nn = (A * A).sum(0).cuda()
    km = ops.KMeans(G, nn, torch.ones(N).cuda(), k)
    firsts = [3, 4100, 5999]
    picks, assign = km.farthest_init_many(firsts)
    assert int(picks.min()) >= 0
    before = km.farthest_init_redo_count()
    monkeypatch.setenv('NIF_KMEANS_INIT_INJECT_TIMEOUT', '1')
    picks2, assign2 = km.farthest_init_many(firsts)
    p1, a1 = km.farthest_init(firsts[1])
    monkeypatch.delenv('NIF_KMEANS_INIT_INJECT_TIMEOUT')
    assert km.farthest_init_redo_count() - before == T + 1
    assert torch.equal(picks, picks2) and torch.equal(assign, assign2)
    assert torch.equal(picks[1], p1) and torch.equal(assign[1], a1)
    picks3, _ = km.farthest_init_many(firsts)          # and the next call is undisturbed
    assert torch.equal(picks, picks3)
    assert km.farthest_init_redo_count() - before == T + 1


def test_kmeans_farthest_init_calls_in_flight_on_many_streams(ops):
    """More multi-workgroup initialisations in flight (one per stream) than the library has mailbox slots: a slot is
    handed out again only behind its last user's event, so every call gets what it gets alone."""
    N, k = 5000, 24
    g = torch.Generator().manual_seed(6)
    A = torch.randn(16, N, generator=g).double()
    G = ((0.9 * A).T @ (0.9 * A)).float().cuda()
    nn = (A * A).sum(0).cuda()
    km = ops.KMeans(G, nn, torch.ones(N).cuda(), k)
    firsts = list(range(0, 4000, 200))     # 20 calls
    alone = [km.farthest_init(f) for f in firsts]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in firsts]
    got = []
    kms = [ops.KMeans(G, nn, km.s, k) for _ in firsts]   # `mind` scratch of its own per call in flight
    torch.cuda.synchronize()
    for f, st, kmi in zip(firsts, streams, kms):
        with torch.cuda.stream(st):
            got.append(kmi.farthest_init(f))
    torch.cuda.synchronize()
    for (p0, a0), (p1, a1) in zip(alone, got):
        assert torch.equal(p0, p1) and torch.equal(a0, a1)


@pytest.mark.parametrize('gdtype', [torch.float64, torch.float32])
def test_kmeans_batched_tries_equal_one_by_one(ops, gdtype):
    """lloyd_step_many / repair_kf_many over a subset of tries: the rows of the single-problem calls, bit for bit."""
    N, k, m, T = 700, 90, 48, 5
    g = torch.Generator().manual_seed(11)
    A = torch.randn(m, N, generator=g)
    s = (torch.rand(N, generator=g) + 0.2).cuda()
    G64, _, _ = ops.xty(A.double().cuda(), None, acc='f64', norms=False)
    G, nn = G64.to(gdtype), torch.diagonal(G64).contiguous()
    assign = torch.randint(0, k - 3, (T, N), generator=g).int().cuda()  # the top labels are empty: repair has work
    many = ops.KMeans(G, nn, s, k, gdiag=nn, tries=T)
    live = [0, 2, 3]
    out, nearest, cost, kp = many.lloyd_step_many(assign, live)
    pre = out.clone()
    changed = many.repair_kf_many(out, nearest, assign, live)
    one = ops.KMeans(G, nn, s, k, gdiag=nn)
    for t in live:
        o1, n1, c1, k1 = one.lloyd_step(assign[t].contiguous())
        assert torch.equal(pre[t], o1) and torch.equal(cost[t], c1) and int(kp[t]) == int(k1)
        ch1 = one.repair_kf(o1, n1, assign[t].contiguous())
        assert torch.equal(out[t], o1) and torch.equal(nearest[t], n1) and int(changed[t]) == int(ch1)


def test_kmeans_centroid_weights_and_repair(ops):
    N, k, d = 40, 9, 17
    g = torch.Generator().manual_seed(2)
    W = torch.randn(d, N, generator=g).double()
    s = torch.rand(N, generator=g) + 0.1
    assign = torch.randint(0, k - 2, (N,), generator=g).int()  # the two top labels are empty
    km = ops.KMeans(None, torch.zeros(N, dtype=torch.float64).cuda(), s.cuda(), k)
    rows, kp = km.centroid_weights(W.cuda(), assign.cuda())
    labels = [c for c in range(k) if (assign == c).any()]
    assert int(kp) == len(labels)
    for r, c in enumerate(labels):
        mask = assign == c
        ref = (W[:, mask] * s[mask].double()).sum(1) / s[mask].double().sum()
        np.testing.assert_allclose(rows[r].cpu().numpy(), ref.float().numpy(), rtol=1e-6, atol=1e-7)
    assert float(rows[len(labels):].abs().max()) == 0.0
    # repair (hf_kf.py:466-471): every empty label takes the neuron with the largest s * distance, in label order
    nearest = torch.rand(N, generator=g).double()
    a_dev, n_dev = assign.clone().cuda(), nearest.clone().cuda()
    changed = km.repair_kf(a_dev, n_dev, assign.cuda())
    a_ref, n_ref = assign.clone().long(), nearest.clone()
    for c in range(k):
        if not (a_ref == c).any():
            far = int(torch.argmax(s.double() * n_ref))
            a_ref[far], n_ref[far] = c, 0.0
    assert (a_dev.cpu().long() == a_ref).all() and int(changed) == 1
    np.testing.assert_allclose(n_dev.cpu().numpy(), n_ref.numpy())


@pytest.mark.parametrize('shape,k,stride,pad,ones', [((3, 5, 8, 8), 3, 1, 1, True), ((2, 4, 9, 7), (3, 2), (2, 1), (0, 1), False),
                                                      ((5, 3, 32, 32), 3, 1, 1, True), ((2, 70, 6, 6), 3, 1, 1, True),
                                                      ((4, 6, 5, 5), 5, 1, 2, True), ((7, 33), None, 1, 0, True),
                                                      ((7, 300), None, 1, 0, False)])
def test_design_matrix_matches_unfold(ops, shape, k, stride, pad, ones):
    """ops.design_matrix against F.unfold + the reference's row order and ones column (hf_kf.py:196-208, 224-225):
    bit-exact (it only moves data)."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(*shape, generator=g)
    if k is None:
        ref = x.reshape(shape[0], -1)
    else:
        u = torch.nn.functional.unfold(x, kernel_size=k, stride=stride, padding=pad)
        ref = u.permute(0, 2, 1).reshape(-1, u.shape[1])
    if ones:
        ref = torch.cat([ref, torch.ones(ref.shape[0], 1)], dim=1)
    out = ops.design_matrix(x.cuda(), k, stride, pad, ones=ones)
    assert out.shape == ref.shape
    assert out.stride(0) % 4 == 0
    assert torch.equal(out.cpu(), ref)
    base = out.as_strided((out.shape[0], out.stride(0)), (out.stride(0), 1))  # pad columns are written as zeros
    assert (base[:, out.shape[1]:] == 0).all()


@pytest.mark.parametrize('m,n0,n1,sym,relu', [(1000, 128, 256, False, False), (777, 132, 260, False, False),
                                              (4096, 384, 384, False, True), (530, 200, 200, True, False),
                                              (16, 128, 128, False, False), (5, 4, 8, False, False),
                                              (20000, 384, 384, False, False), (3000, 520, 520, True, True)])
def test_xty_split_bf16_matches_f64_gram(ops, m, n0, n1, sym, relu):
    """The split-bf16 Gram (six bf16 matrix instructions on the exact three-way split of every f32 operand) is as
    close to the exact Gram as the f32 matrix path: both are checked against the f64-accumulated result on ragged
    tiles (columns not a multiple of 128, rows not a multiple of 16), symmetric mode and ReLU-on-load."""
    g = torch.Generator().manual_seed(m + n0)
    A = torch.randn(m, n0, generator=g).cuda()
    B = None if sym else torch.randn(m, n1, generator=g).cuda()
    kw = dict(relu_a=relu, relu_b=relu and not sym)
    G, na, nb = ops.xty(A, B, split_bf16=True, **kw)
    Ar = A.double().clamp_min(0) if relu else A.double()
    Br = Ar if sym else (B.double().clamp_min(0) if relu else B.double())
    ref = Ar.T @ Br
    scale = ref.abs().max().item()
    G32, _, _ = ops.xty(A, B, **kw)
    err, err32 = (G - ref).abs().max().item() / scale, (G32 - ref).abs().max().item() / scale
    assert err < 3e-6 and err < 4 * err32 + 1e-7, (err, err32)
    assert torch.allclose(na, (Ar ** 2).sum(0), rtol=2e-6)
    assert torch.allclose(nb, (Br ** 2).sum(0), rtol=2e-6)
    if sym:
        assert torch.equal(G, G.T)


def test_xty_split_bf16_falls_back_when_rows_are_not_16_byte_pieces(ops):
    """Operands the split kernel cannot address in 16-byte pieces (odd width) take the f32 matrix path: same bits."""
    g = torch.Generator().manual_seed(3)
    A = torch.randn(300, 131, generator=g).cuda()
    B = torch.randn(300, 64, generator=g).cuda()
    G0, na0, nb0 = ops.xty(A, B)
    G1, na1, nb1 = ops.xty(A, B, split_bf16=True)
    assert torch.equal(G0, G1) and torch.equal(na0, na1) and torch.equal(nb0, nb1)
    with pytest.raises(Exception):
        ops.xty(A, B, acc='f64', split_bf16=True)


@pytest.mark.parametrize('split', [False, True])
def test_xty_empty_contraction_axis(ops, split):
    """No fusion samples at all (an empty shard of a sharded run): the Gram and the norms are exact zeros."""
    A = torch.empty(0, 12, device='cuda')
    B = torch.empty(0, 20, device='cuda')
    G, na, nb = ops.xty(A, B, split_bf16=split)
    assert G.shape == (12, 20) and not G.any() and not na.any() and not nb.any()
    Gs, nas, _ = ops.xty(A, None, acc='f64')
    assert Gs.shape == (12, 12) and not Gs.any() and not nas.any()


def test_lap_scipy_order_rejects_invalid_costs_like_scipy(ops):
    """scipy.optimize.linear_sum_assignment raises ValueError on NaN entries and on infeasible (all-infinite)
    problems; the call sites that stand in for it do the same instead of returning a partial matching."""
    C = torch.rand(40, 40)
    bad = C.clone()
    bad[3, :] = float('inf')
    with pytest.raises(ValueError):
        linear_sum_assignment(bad.numpy())
    with pytest.raises(ValueError):
        ops.lap_solve(bad.cuda(), scipy_order=True)
    nan = C.clone()
    nan[5, 7] = float('nan')
    with pytest.raises(ValueError):
        linear_sum_assignment(nan.numpy())
    with pytest.raises(ValueError):
        ops.lap_solve(nan.cuda(), scipy_order=True)


def test_xty_writes_into_caller_buffers(ops):
    """out=: the reduction writes into views of one packed buffer (what the sharded level hands to the all-reduce)."""
    from neuron_interpolation_model_fusion_amd.parallel import unpack_gram, pack_gram
    g = torch.Generator().manual_seed(4)
    A = torch.randn(500, 40, generator=g).cuda()
    B = torch.randn(500, 24, generator=g).cuda()
    G, na, nb = ops.xty(A, B)
    buf = torch.full((40 * 24 + 40 + 24,), float('nan'), dtype=torch.float64, device='cuda')
    G2, na2, nb2 = ops.xty(A, B, out=unpack_gram(buf, 40, 24))
    assert G2.data_ptr() == buf.data_ptr()
    assert torch.equal(buf, pack_gram(G, na, nb))
    with pytest.raises(ValueError):
        ops.xty(A, B, out=(G.float(), na, nb))


@pytest.mark.parametrize('n,block,acc', [(70, 8, 'f32'), (300, 64, 'f32'), (257, 1024, 'f64'), (3000, 1024, 'f32'),
                                         (2881, 100, 'f32')])
def test_xty_sym_packed_equals_packing_the_full_gram(ops, n, block, acc):
    """The split-K reduction writing the packed block upper triangle (what a sharded run all-reduces) gives the bits
    of the full symmetric Gram packed afterwards, and nif_unpack_upper the bits of the full matrix (f64, or the f32
    k-means storage with its f64 diagonal)."""
    from neuron_interpolation_model_fusion_amd import parallel
    g = torch.Generator().manual_seed(n)
    A = torch.randn(900, n, generator=g).cuda()
    G, na, _ = ops.xty(A, None, acc=acc)
    buf, plen = ops.xty_sym_packed(A, acc=acc, norms=True, block=block, tail=3)
    assert buf.numel() == plen + n + 3
    ref = parallel.pack_gram_upper(G, na, block=block)
    assert ref.numel() == plen + n
    assert torch.equal(buf[:plen + n], ref)
    assert torch.equal(ops.unpack_upper(buf, n, block=block), G)
    Gf, gd = ops.unpack_upper(buf, n, block=block, dtype=torch.float32, gdiag=True)
    assert torch.equal(Gf, G.float()) and torch.equal(gd, torch.diagonal(G))
    G2, na2 = parallel.unpack_gram_upper(buf[:plen + n], n, block=block)
    assert torch.equal(G2, G) and torch.equal(na2, na)
    # ReLU on load and the transposed operand layout go through the same path
    bufr, _ = ops.xty_sym_packed(A, acc=acc, relu=True, block=block)
    Gr, _, _ = ops.xty(A, None, acc=acc, relu_a=True, norms=False)
    assert torch.equal(ops.unpack_upper(bufr, n, block=block), Gr)
    At = A.T.contiguous()
    buft, _ = ops.xty_sym_packed(At, acc=acc, trans=True, block=block)
    Gt, _, _ = ops.xty(At, None, acc=acc, trans_a=True, norms=False)
    assert torch.equal(ops.unpack_upper(buft, n, block=block), Gt)


# ---- symmetric eigendecomposition with a cutoff (the LS refit's factor step) --------------------------------------

def _eig_cases():
    rng = np.random.default_rng(0)
    cases = {}
    for n in (1, 2, 3, 4, 5, 17, 64, 65, 130, 257):
        B = rng.standard_normal((n + 3, n))
        cases[f'full{n}'] = (B.T @ B, 1e-13)
    X = np.maximum(rng.standard_normal((900, 24)) @ rng.standard_normal((24, 200)) + 0.3 * rng.standard_normal((900, 200)), 0)
    X = X.astype(np.float32).astype(np.float64)
    X[:, ::7] = 0.0  # dead units: exactly zero rows / columns of the Gram
    cases['relu200'] = (X.T @ X, (900 * 1.1920929e-07) ** 2)
    L = rng.standard_normal((300, 40))
    cases['lowrank300'] = (L @ L.T, 1e-10)
    Q, _ = np.linalg.qr(rng.standard_normal((150, 150)))
    cases['multiple150'] = (Q @ np.diag(np.r_[np.ones(40), np.full(40, 0.5), np.linspace(0.01, 0.4, 70)]) @ Q.T, 1e-6)
    nw = 101
    cases['wilkinson101'] = (np.diag(np.abs(np.arange(nw) - nw // 2).astype(float) + 3.0) + np.diag(np.ones(nw - 1), 1) +
                             np.diag(np.ones(nw - 1), -1), 1e-9)
    cases['diag50'] = (np.diag(np.linspace(0.0, 5.0, 50)), 1e-3)
    blk = np.zeros((90, 90))
    blk[:45, :45] = cases['full17'][0][:1, :1] * 0 + rng.standard_normal((45, 45))
    blk[:45, :45] = blk[:45, :45] @ blk[:45, :45].T
    blk[45:, 45:] = blk[:45, :45]   # two identical decoupled blocks: every eigenvalue is double, T splits
    cases['twoblocks90'] = (blk, 1e-12)
    return cases


@pytest.mark.parametrize('name', sorted(_eig_cases()))
def test_eigh_kept_matches_numpy(ops, name):
    G, cutoff = _eig_cases()[name]
    n = G.shape[0]
    lam_ref, U = np.linalg.eigh(G)
    keep = lam_ref > cutoff * lam_ref[-1]
    lam, V = ops.eigh_kept(torch.from_numpy(G).cuda(), cutoff)
    lam, V = lam.cpu().numpy(), V.cpu().numpy()
    scale = max(abs(lam_ref[-1]), 1e-300)
    # an eigenvalue within rounding of the threshold may fall on either side: compare away from it
    margin = 1e-10 * scale
    sure = lam_ref > cutoff * lam_ref[-1] + margin
    assert sure.sum() <= lam.shape[0] <= (lam_ref > cutoff * lam_ref[-1] - margin).sum()
    if lam.shape[0] == keep.sum():
        np.testing.assert_allclose(lam, lam_ref[keep], rtol=0, atol=1e-12 * scale * max(n, 10))
    assert V.shape == (n, lam.shape[0])
    K = lam.shape[0]
    assert np.abs(V.T @ V - np.eye(K)).max() < 1e-8
    assert np.abs(G @ V - V * lam).max() < 1e-11 * scale * max(n, 10)
    # the spectral function the refit needs: sum over kept of v v^T / lambda
    if lam.shape[0] == keep.sum() and K > 0:
        P = (V / lam) @ V.T
        Pref = (U[:, keep] / lam_ref[keep]) @ U[:, keep].T
        assert np.abs(P - Pref).max() <= 1e-7 * np.abs(Pref).max()


@pytest.mark.parametrize('m,d', [(700, 577), (300, 1153), (2600, 130)])
def test_ls_solve_with_own_eigensolver_matches_f64_pinv(ops, m, d):
    """W = pinv(X) A through nif_eigh_kept_f64 (primal and dual forms) against numpy's SVD-based pinv in f64 with the
    same cutoff; X has dead columns and a decaying spectrum so that the cutoff is active."""
    rng = np.random.default_rng(m)
    r = min(m, d)
    U, _ = np.linalg.qr(rng.standard_normal((m, r)))
    Vt, _ = np.linalg.qr(rng.standard_normal((d, r)))
    sv = np.logspace(0, -7, r)
    X = (U * sv) @ Vt.T
    X[:, ::11] = 0.0
    X = X.astype(np.float32)
    A = rng.standard_normal((m, 9)).astype(np.float32)
    rtol = max(m, d) * 1.1920928955078125e-07
    ref = np.linalg.pinv(X.astype(np.float64), rcond=rtol) @ A.astype(np.float64)
    sol = ops.ls_solve(torch.from_numpy(X).cuda(), torch.from_numpy(A).cuda())
    s = np.linalg.svd(X.astype(np.float64), compute_uv=False)
    gap = np.abs(s / s[0] - rtol).min()
    assert gap > 1e-6 * rtol, 'test matrix has a singular value on the cutoff'
    assert sol['rank'] == int((s > rtol * s[0]).sum())
    W = sol['W'].cpu().numpy()
    assert np.abs(W - ref).max() <= 2e-6 * np.abs(ref).max()


def test_lap_solve_many_equals_one_by_one(ops):
    """Concurrent multi-stream solves of independent problems: same matchings and paths as solving them in turn, in the
    fast mode and in scipy order (a planted instance takes the certified path, a tie-laden one the replay kernel)."""
    g = torch.Generator().manual_seed(21)
    costs = []
    for n in (40, 384, 129, 16):
        costs.append(torch.rand(n, n, generator=g).cuda())
    A = torch.randn(500, 200, generator=g)
    perm = torch.randperm(200, generator=g)
    B = A[:, perm] + 0.05 * torch.randn(500, 200, generator=g)
    G, na, nb = ops.xty(A.cuda(), B.cuda())
    costs.append(ops.cost_dist(G, na, nb))                      # nearly solved by the dual initialisation
    tied = torch.rand(96, 96, generator=g)
    tied[:, 48:] = tied[:, :48]                                 # duplicated columns: exact ties
    costs.append(tied.cuda())
    for mode in (False, True):
        one = [ops.lap_solve(C, scipy_order=mode) for C in costs]
        paths = []
        for C in costs:
            ops.lap_solve(C, scipy_order=mode)
            paths.append(ops.lap_solve.last_path)
        many, rows = ops.lap_solve_many(costs, scipy_order=mode, rows=True)
        for a, b, r in zip(one, many, rows):
            assert torch.equal(a, b)
            assert torch.equal(r[b.long()].long(), torch.arange(b.shape[0], device='cuda'))
        assert ops.lap_solve_many.last_paths == paths
    if True:
        from scipy.optimize import linear_sum_assignment
        many = ops.lap_solve_many(costs, scipy_order=True)
        for C, col in zip(costs, many):
            assert (linear_sum_assignment(C.cpu().numpy())[1] == col.cpu().numpy()).all()


@pytest.mark.parametrize('n,B', [(1, 3), (37, 5), (64, 1), (130, 4), (384, 9), (900, 2), (1100, 3), (2100, 2)])
def test_lap_solve_batch_equals_one_call_per_problem(ops, n, B):
    """nif_lap_solve_batch_f32 (the problems along gridDim.z of every kernel): matchings AND duals bit-identical to one
    nif_lap_solve_f32 call per problem -- hard instances (unrelated activations), a planted one, a tie-laden one; row and
    problem strides wider than the matrices; the staged flags (INIT_ONLY, then RESUME) and the scipy-order kernel too."""
    g = torch.Generator().manual_seed(1000 + n)
    m = max(2 * n, 64)
    probs = []
    for b in range(B):
        A0 = torch.randn(m, n, generator=g)
        A1 = torch.randn(m, n, generator=g) if b != 1 else A0[:, torch.randperm(n, generator=g)] + 0.1 * torch.randn(m, n, generator=g)
        probs.append(torch.cdist(A0.T.contiguous(), A1.T.contiguous()))
    if B > 2 and n >= 4:
        probs[2][:, n // 2:n // 2 * 2] = probs[2][:, :n // 2]      # duplicated columns: exact ties
    big = torch.zeros(B, n + 3, n + 5)
    for b in range(B):
        big[b, :n, :n] = probs[b]
    big = big.cuda()
    C = big[:, :n, :n]                                               # strides (n + 3) (n + 5), n + 5, 1
    for flags in (0, ops._lib.NIF_LAP_SCIPY_ORDER):
        col, row, u, v = ops.lap_solve_batch(C, duals=True, flags=flags)
        for b in range(B):
            one = ops.lap_solve(C[b], duals=True, scipy_order='replay' if flags else False)
            for got, ref in zip((col[b], row[b], u[b], v[b]), one):
                assert torch.equal(got, ref), (n, b, flags)
        if flags:
            from scipy.optimize import linear_sum_assignment
            for b in range(B):
                assert (linear_sum_assignment(probs[b].numpy())[1] == col[b].cpu().numpy()).all()
    # staged: initialisation only, then the searches from that state (what the scipy-order call sites do)
    cols = torch.empty((B, n), dtype=torch.int32, device='cuda')
    rows = torch.empty_like(cols)
    u = torch.empty((B, n), dtype=torch.float64, device='cuda')
    v = torch.empty_like(u)
    nbytes = ops.lib.nif_lap_batch_workspace_bytes(n, B)
    assert nbytes == B * ops.lib.nif_lap_workspace_bytes(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    Cc = C.contiguous()
    for flags in (ops._lib.NIF_LAP_INIT_ONLY, ops._lib.NIF_LAP_RESUME):
        ops.check(ops.lib.nif_lap_solve_batch_f32(Cc.data_ptr(), n, n * n, n, B, flags, cols.data_ptr(), rows.data_ptr(),
                                                  u.data_ptr(), v.data_ptr(), ws.data_ptr(), nbytes, None), 'batch')
        if flags == ops._lib.NIF_LAP_INIT_ONLY:
            free_after_init = int((cols < 0).sum())
    fast = ops.lap_solve_batch(C)
    assert torch.equal(cols, fast)
    assert int((cols < 0).sum()) == 0 and free_after_init >= 0
    # argument checks: overlapping problems, missing outputs, short workspace
    lib = ops.lib
    assert lib.nif_lap_solve_batch_f32(Cc.data_ptr(), n, n * n - 1 if n > 1 else -1, n, max(B, 2), 0, cols.data_ptr(), rows.data_ptr(),
                                       u.data_ptr(), v.data_ptr(), ws.data_ptr(), nbytes, None) == -1
    assert lib.nif_lap_solve_batch_f32(Cc.data_ptr(), n, n * n, n, B, 0, cols.data_ptr(), None, u.data_ptr(), v.data_ptr(),
                                       ws.data_ptr(), nbytes, None) == -1
    assert lib.nif_lap_solve_batch_f32(Cc.data_ptr(), n, n * n, n, B, 0, cols.data_ptr(), rows.data_ptr(), u.data_ptr(),
                                       v.data_ptr(), ws.data_ptr(), nbytes - 1, None) != 0


def test_lap_solve_many_batches_same_size_problems(ops):
    """Fast mode: problems of one size take the batched call, odd sizes their own streams; same answers as one by one."""
    g = torch.Generator().manual_seed(5)
    costs = [torch.rand(n, n, generator=g).cuda() for n in (200, 64, 200, 200, 33, 64)]
    one = [ops.lap_solve(C) for C in costs]
    many, rows = ops.lap_solve_many(costs, scipy_order=False, rows=True)
    for a, b, r in zip(one, many, rows):
        assert torch.equal(a, b)
        assert torch.equal(r[b.long()].long(), torch.arange(b.shape[0], device='cuda'))
    # scipy order: the replays of same-size problems (tie-laden costs, and sizes below the staged path's 32) are one
    # batched call per size; certified problems stay on their own streams.  Same matchings and paths as one by one, = scipy
    from scipy.optimize import linear_sum_assignment
    tied = []
    for k in range(3):
        t = torch.rand(96, 96, generator=g)
        t[:, 48:] = t[:, :48]                       # duplicated columns: exact ties -> replay
        tied.append(t.cuda())
    small = [torch.randint(0, 3, (16, 16), generator=g).float().cuda() for _ in range(3)]   # n < 32: replay at once
    mixed = [tied[0], costs[0], small[0], tied[1], costs[2], small[1], tied[2], small[2], costs[1]]
    paths = []
    for C in mixed:
        ops.lap_solve(C, scipy_order=True)
        paths.append(ops.lap_solve.last_path)
    many, rows = ops.lap_solve_many(mixed, scipy_order=True, rows=True)
    assert ops.lap_solve_many.last_paths == paths and paths.count('replay') >= 6
    for C, col, r in zip(mixed, many, rows):
        assert (linear_sum_assignment(C.cpu().numpy())[1] == col.cpu().numpy()).all()
        assert torch.equal(r[col.long()].long(), torch.arange(col.shape[0], device='cuda'))
