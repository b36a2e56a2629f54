"""Probes at the C3 / C4 shapes: LAP 8192 (fast and scipy-order), ALF k-means pieces at N=768, m=325000."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuron_interpolation_model_fusion_amd import ops

def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

g = torch.Generator(device='cuda').manual_seed(0)
which = sys.argv[1] if len(sys.argv) > 1 else 'all'
if which in ('all', 'c3'):
    for m in (400, 4096):
        n = 8192
        A0 = torch.randn(m, n, device='cuda', generator=g).clamp_min(0)
        perm = torch.randperm(n, device='cuda')
        A1 = 0.9 * A0[:, perm] + 0.3 * torch.randn(m, n, device='cuda', generator=g)
        G, na, nb = ops.xty(A0, A1)
        C = ops.cost_dist(G, na, nb)
        print(f'C3 m={m}: gram {t(lambda: ops.xty(A0, A1)):.2f} ms, cost {t(lambda: ops.cost_dist(G, na, nb)):.2f} ms, '
              f'lap fast {t(lambda: ops.lap_solve(C), 1):.1f} ms, lap scipy-order {t(lambda: ops.lap_solve(C, scipy_order=True), 1):.1f} ms', flush=True)
        del A0, A1, G, C
if which in ('all', 'c4'):
    m, h = 325000, 384
    acts = [torch.randn(m, h, device='cuda', generator=g) for _ in range(2)]
    acts[1] = acts[0][:, torch.randperm(h, device='cuda')] + 0.5 * acts[1]
    A = torch.cat(acts, dim=1).contiguous()
    print(f'C4 normalise+cat: {t(lambda: torch.cat([a / a.norm(dim=1, keepdim=True) for a in acts], dim=1)):.2f} ms')
    K = torch.cat([a / a.norm(dim=1, keepdim=True) for a in acts], dim=1).contiguous()
    print(f'C4 symmetric gram N=768: {t(lambda: ops.xty(K, None, norms=False)):.2f} ms')
    G, _, _ = ops.xty(K, None, norms=False)
    w = torch.ones(2 * h, device='cuda')
    km = ops.KMeans(G, torch.diagonal(G).contiguous(), w, h)
    a0 = km.assign_to_neurons(0)
    print(f'C4 assign_to_neurons: {t(lambda: km.assign_to_neurons(0)):.3f} ms; lloyd step: {t(lambda: km.lloyd_step(a0)):.3f} ms')
    a1, nearest, cost, kp = km.lloyd_step(a0)
    print('   kprime', int(kp), 'cost', float(cost))
    print(f'C4 targets (m x k): {t(lambda: km.targets(A, a1)):.2f} ms')
