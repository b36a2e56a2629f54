"""Time per phase of the exact transport kernel (needs libnif.so built with -DNIF_EMD_TIMING; 100 MHz wall clock)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuron_interpolation_model_fusion_amd import ops

g = torch.Generator(device='cuda').manual_seed(0)
for n in (384, 1024):
    a = torch.rand(n, device='cuda', generator=g) + 0.05; b = torch.rand(n, device='cuda', generator=g) + 0.05
    C = torch.cdist(torch.randn(n, 12, device='cuda', generator=g), torch.randn(n, 12, device='cuda', generator=g))
    for _ in range(2):
        ops.emd_solve(C, a / a.sum(), b / b.sum())
    st = ops.emd_solve.last_status
    names = ('settle', 'relax', 'reduce', 'duals', 'aug: slow path + barrier', 'relabel', 'aug: list trees + barrier', 'aug: wave 0')
    t = st[4:12]
    print(n, ops.emd_solve.last_work, 'hops (walk iterations of lane 0 / arcs of lane 0 / evert hops of lane 0)', st[12:15])
    for k, v in zip(names, t):
        print(f'   {k:28s} {v / 100e3:7.3f} ms  {100 * v / sum(t):5.1f}%')
