"""Fused Lloyd pass (kmeans_nearest_fused_kernel) on random labels -- cluster sizes Poisson around N / k, so most member
rows sit in a cluster's remainder -- timed with the library's own HIP events (ops.PROF_KMEANS): min of 5 x 20 launches."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuron_interpolation_model_fusion_amd import ops
g = torch.Generator(device='cuda').manual_seed(0)
for N, k, m in ((768, 384, 4096), (2048, 512, 4096), (6144, 768, 4096), (24576, 3072, 4096)):
    A = torch.randn(m, N, device='cuda', generator=g)
    G, _, _ = ops.xty(A, None, norms=False)
    nn = torch.diagonal(G).contiguous()
    km = ops.KMeans(G.float(), nn, torch.ones(N, device='cuda'), k, gdiag=nn)
    a = torch.randint(0, k, (N,), device='cuda', generator=g).int()
    for _ in range(3):
        km.lloyd_step(a)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        ops.profile_collect(ops.PROF_KMEANS)
        ops.profile_enable(True, slots=[ops.PROF_KMEANS])
        for _ in range(20):
            km.lloyd_step(a)
        torch.cuda.synchronize()
        ops.profile_enable(False)
        ms, n = ops.profile_collect(ops.PROF_KMEANS)
        best = min(best, ms / max(n, 1))
    print(f'N={N} k={k}: fused pass {best * 1e3:.1f} us '
          f'({4.0 * N * N / best / 1e6:.0f} GB/s)', flush=True)
