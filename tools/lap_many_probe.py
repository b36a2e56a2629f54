"""How far independent single-workgroup assignments overlap: ops.lap_solve_many on B copies of one hard instance
(unrelated activations, n = 384 by default), wall-clock per batch and per problem."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuron_interpolation_model_fusion_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 384
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev).manual_seed(4242)
A0 = torch.randn(20000, n, device=dev, generator=g)
A1 = torch.randn(20000, n, device=dev, generator=g)
G, na, nb = ops.xty(A0, A1)
C = ops.cost_dist(G, na, nb)
print('GPU_MAX_HW_QUEUES =', os.environ.get('GPU_MAX_HW_QUEUES'))
for B in (1, 2, 4, 8, 16, 32, 2, 1, 1):
    costs = [C.clone() for _ in range(B)]
    ops.lap_solve_many(costs, scipy_order=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ops.lap_solve_many(costs, scipy_order=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f'n={n} B={B}: {dt * 1e3:.2f} ms per batch, {dt / B * 1e3:.3f} ms per problem', flush=True)
# one problem through the plain call and through the batch entry, 50 solves each
for name, fn in (('lap_solve', lambda: ops.lap_solve(C)), ('lap_solve_batch B=1', lambda: ops.lap_solve_batch(C[None])),
                 ('lap_solve_batch B=2', lambda: ops.lap_solve_batch(torch.stack([C, C])))):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    print(f'{name}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per call', flush=True)
