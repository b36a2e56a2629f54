"""LAP kernel on hard instances (uniform random costs; random-init VGG11 layer costs), both modes."""
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
for n in (128, 256, 384, 512, 1024, 2048):
    C = torch.rand(n, n, device='cuda', generator=g)
    A = torch.randn(1600, n, device='cuda', generator=g).clamp_min(0)
    B = torch.randn(1600, n, device='cuda', generator=g).clamp_min(0)
    G, na, nb = ops.xty(A, B)
    Cd = ops.cost_dist(G, na, nb)
    print(f'n={n}: uniform fast {t(lambda: ops.lap_solve(C)):.2f} ms, scipy-order {t(lambda: ops.lap_solve(C, scipy_order=True)):.2f} ms; '
          f'unrelated-activations cdist fast {t(lambda: ops.lap_solve(Cd)):.2f} ms, scipy-order {t(lambda: ops.lap_solve(Cd, scipy_order=True)):.2f} ms', flush=True)
