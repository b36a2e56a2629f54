"""Timing and work counters of the exact transport kernel on random and on planted (fusion-like) problems."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuron_interpolation_model_fusion_amd import ops

from neuron_interpolation_model_fusion_amd import _lib

def check(C, a, b, plan, u, v):
    """marginals met, duals feasible, complementary slackness (relative to the cost scale)"""
    T = plan.dense(torch.float64)
    a = a.double(); b = b.double() * (a.sum() / b.double().sum())
    red = C.double() - u[:, None] - v[None, :]
    scale = float(C.abs().max())
    return (f'marg {float((T.sum(1) - a).abs().max() + (T.sum(0) - b).abs().max()):.1e} '
            f'minred {float(red.min()) / scale:.1e} slack {float((red * (T > 0)).abs().max()) / scale:.1e}')

def run(tag, C, a, b):
    variants = [('forest warm', 0), ('forest cold', _lib.NIF_EMD_COLD_START), ('pool warm', _lib.NIF_EMD_ARC_POOL),
                ('pool cold', _lib.NIF_EMD_ARC_POOL | _lib.NIF_EMD_COLD_START)]
    if C.shape[0] != C.shape[1]:
        variants = [('forest', 0), ('pool', _lib.NIF_EMD_ARC_POOL)]
    for name, flags in variants:
        ops.emd_solve(C, a, b, flags=flags); torch.cuda.synchronize()
        t0 = time.perf_counter(); plan, u, v = ops.emd_solve(C, a, b, duals=True, flags=flags); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        w = ops.emd_solve.last_work
        cost = float((plan.dense(torch.float64) * C.double()).sum())
        print(f'{tag} [{name}]: {C.shape[0]}x{C.shape[1]} {dt*1e3:.1f} ms, nnz {plan.nnz}, {w}, cost {cost:.9f}, '
              f'{check(C, a, b, plan, u, v)}', flush=True)

g = torch.Generator(device='cuda').manual_seed(0)
for n0, n1 in ((384, 384), (512, 512), (1024, 1024), (2100, 1500)):
    a = torch.rand(n0, device='cuda', generator=g) + 0.05; b = torch.rand(n1, device='cuda', generator=g) + 0.05
    C = torch.cdist(torch.randn(n0, 12, device='cuda', generator=g), torch.randn(n1, 12, device='cuda', generator=g))
    run('random 12-d', C, a / a.sum(), b / b.sum())
    if n0 == n1:
        # fusion-like: model-k neurons are noisy permuted copies of the target's, importances correlated
        m = 2000
        A = torch.randn(m, n1, device='cuda', generator=g)
        perm = torch.randperm(n0, device='cuda')
        B = 0.9 * A[:, perm] + 0.3 * torch.randn(m, n0, device='cuda', generator=g)
        G, na, nb = ops.xty(B, A)
        C = ops.cost_dist(G, na, nb)
        s = torch.rand(n1, device='cuda', generator=g) + 0.2
        a2 = s[perm] * (1 + 0.1 * torch.rand(n0, device='cuda', generator=g)); b2 = s
        run('planted', C, a2 / a2.sum(), b2 / b2.sum())
