import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuron_interpolation_model_fusion_amd import ops
N, k, m = 24576, 3072, 4096
g = torch.Generator(device='cuda').manual_seed(0)
A = torch.randn(m, N, device='cuda', generator=g)
G, _, _ = ops.xty(A, None, norms=False)
nn = torch.diagonal(G).contiguous()
km = ops.KMeans(G.float(), nn, torch.ones(N, device='cuda'), k, gdiag=nn)
for frac in (0.0, 0.5, 0.95):
    a = torch.randint(0, k, (N,), device='cuda', generator=g).int()
    big = torch.rand(N, device='cuda', generator=g) < frac
    a[big] = 7
    km.lloyd_step(a); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): km.lloyd_step(a)
    torch.cuda.synchronize()
    print(f'fraction {frac} of the neurons in one cluster: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per Lloyd step', flush=True)
