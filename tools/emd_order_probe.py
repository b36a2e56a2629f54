import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from neuron_interpolation_model_fusion_amd import ops
g = torch.Generator(device='cuda').manual_seed(0)
for n in (384, 1024):
    a = torch.rand(n, device='cuda', generator=g) + 0.05; b = torch.rand(n, device='cuda', generator=g) + 0.05
    C = torch.cdist(torch.randn(n, 12, device='cuda', generator=g), torch.randn(n, 12, device='cuda', generator=g))
    a = a / a.sum(); b = b / b.sum()
    col = ops.lap_solve(C).long()
    surplus = a - b[col]
    orders = {'as is': torch.arange(n, device='cuda'), 'a descending': torch.argsort(-a), 'a ascending': torch.argsort(a),
              'surplus descending': torch.argsort(-surplus), 'surplus ascending': torch.argsort(surplus),
              '|surplus| ascending': torch.argsort(surplus.abs())}
    for name, p in orders.items():
        Cp, ap = C[p].contiguous(), a[p].contiguous()
        ops.emd_solve(Cp, ap, b); torch.cuda.synchronize()
        t0 = time.perf_counter(); plan = ops.emd_solve(Cp, ap, b); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f'n={n} rows {name}: {dt*1e3:.1f} ms {ops.emd_solve.last_work} cost {float((plan.dense(torch.float64) * Cp.double()).sum()):.9f}', flush=True)
