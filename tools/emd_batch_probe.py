"""B copies of one transport problem (unrelated-activation costs, random marginals) through ops.emd_solve_many: the batched
entry point (nif_emd_solve_batch_f32) against the batch size."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuron_interpolation_model_fusion_amd import ops
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev).manual_seed(7)
for n in (384, 512):
    A0 = torch.randn(4096, n, device=dev, generator=g); A1 = torch.randn(4096, n, device=dev, generator=g)
    G, na, nb = ops.xty(A0, A1); C = ops.cost_dist(G, na, nb)
    a = torch.rand(n, device=dev, generator=g, dtype=torch.float64) + 0.05; a /= a.sum()
    b = torch.rand(n, device=dev, generator=g, dtype=torch.float64) + 0.05; b /= b.sum()
    for B in (1, 2, 4, 12, 24):
        probs = [(C, a, b)] * B
        ops.emd_solve_many(probs); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): ops.emd_solve_many(probs)
        torch.cuda.synchronize()
        print(f'n={n} B={B}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per batch', flush=True)
