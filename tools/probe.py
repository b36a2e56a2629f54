import torch, time, sys
sys.path.insert(0, '.')
from neuron_interpolation_model_fusion_amd import ops
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
g = torch.Generator(device='cuda').manual_seed(0)
for (m, n, acc) in [(325000, 384, 'f32'), (325000, 1536, 'f32'), (325000, 768, 'f32'), (4096, 8192, 'f32'), (400, 8192, 'f32'), (102400, 577, 'f64'), (6400, 4609, 'f64')]:
    A = torch.randn(m, n, device='cuda', generator=g); B = torch.randn(m, n, device='cuda', generator=g)
    ms = t(lambda: ops.xty(A, B, acc=acc))
    print(f'xty m={m} n={n} {acc}: {ms:.3f} ms  {2*m*n*n/ms/1e9:.1f} TF')
    if n == 768:
        ms = t(lambda: ops.xty(A, None, acc=acc)); print(f'  symmetric: {ms:.3f} ms {m*n*(n+1)/ms/1e9:.1f} TF(alg)')
    del A, B
for n in [384, 1536, 4096, 8192]:
    m = 2000
    A = torch.randn(m, n, device='cuda', generator=g)
    perm = torch.randperm(n, device='cuda')
    B = A[:, perm] + 0.1 * torch.randn(m, n, device='cuda', generator=g)
    G, na, nb = ops.xty(A, B)
    C = ops.cost_dist(G, na, nb)
    ms = t(lambda: ops.lap_solve(C), 3)
    print(f'lap planted n={n}: {ms:.3f} ms')
    R = torch.rand(n, n, device='cuda') if n <= 4096 else None
    if R is not None:
        ms = t(lambda: ops.lap_solve(R), 1)
        print(f'lap uniform-random n={n}: {ms:.3f} ms')
