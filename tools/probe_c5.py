"""C5 shapes on one GPU's shard: 8 models, widths 768 / 3072, 25 000 rows per GPU (200k over 8 GPUs):
symmetric Gram N x N (N = 6144 / 24576) then Gram-form k-means pieces."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuron_interpolation_model_fusion_amd import ops

def t(fn, n=2):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

g = torch.Generator(device='cuda').manual_seed(0)
m = 25000
for h in (768, 3072):
    N, k = 8 * h, h
    base = torch.randn(m, h, device='cuda', generator=g)
    A = torch.cat([base[:, torch.randperm(h, device='cuda')] + 0.4 * torch.randn(m, h, device='cuda', generator=g) for _ in range(8)], dim=1).contiguous()
    tg = t(lambda: ops.xty(A, None, norms=False), 1)
    G, _, _ = ops.xty(A, None, norms=False)
    flops = m * N * (N + 1)
    print(f'C5 h={h}: symmetric Gram N={N}: {tg:.1f} ms = {flops / tg / 1e9:.1f} TF (f32 MFMA)', flush=True)
    w = torch.ones(N, device='cuda')
    km = ops.KMeans(G, torch.diagonal(G).contiguous(), w, k)
    a0 = km.assign_to_neurons(0)
    ts = t(lambda: km.lloyd_step(a0), 2)
    a1, nearest, cost, kp = km.lloyd_step(a0)
    bytes_step = 8.0 * N * N + 16.0 * k * N
    print(f'   lloyd step: {ts:.2f} ms = {bytes_step / ts / 1e6:.0f} GB/s of 8 N^2 + 16 k N bytes; kprime {int(kp)}', flush=True)
    tt = t(lambda: km.targets(A, a1), 2)
    print(f'   targets (m x k): {tt:.2f} ms = {(4.0 * m * N + 4.0 * m * k) / tt / 1e6:.0f} GB/s', flush=True)
    ti = t(lambda: km.farthest_init(0), 1)
    print(f'   farthest-point init (k={k} picks): {ti:.1f} ms', flush=True)
    del A, G, km
    torch.cuda.empty_cache()
