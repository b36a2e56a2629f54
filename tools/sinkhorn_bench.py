"""Sinkhorn mode: time per iteration and achieved bytes/s (8 n0 n1 bytes per iteration: K f32 streamed twice)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuron_interpolation_model_fusion_amd import ops

g = torch.Generator(device='cuda').manual_seed(0)
for n in [int(x) for x in (sys.argv[1:] or ['384', '1024', '2048', '4096', '8192'])]:
    C = torch.rand(n, n, device='cuda', generator=g)
    a = torch.full((n,), 1.0 / n, dtype=torch.float64, device='cuda')
    iters = 200
    ops.sinkhorn(C, a, a, 0.05, max_iter=20, tol=0.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = ops.sinkhorn(C, a, a, 0.05, max_iter=iters, tol=0.0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'n={n}: {iters} iterations {dt * 1e3:.2f} ms = {dt / iters * 1e6:.1f} us/iteration, '
          f'{8.0 * n * n * iters / dt / 1e9:.0f} GB/s (state {r["state"]}, err {r["err"]:.2e})', flush=True)
