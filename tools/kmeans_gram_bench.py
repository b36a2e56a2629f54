"""Gram of the clustering vectors in k-means storage at the C5b shape: ops.kmeans_gram (f32 + f64 diagonal straight out
of the reduction) against the two steps it replaces (f64 Gram, then rounding)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuron_interpolation_model_fusion_amd import ops

N, m = 24576, 25000
A = torch.randn(m, N, device='cuda')
def two_steps():
    G, _, _ = ops.xty(A, None, acc='f32', norms=False)
    return ops.kmeans_gram_storage(G)
for name, fn in (('xty + kmeans_gram_storage', two_steps), ('kmeans_gram', lambda: ops.kmeans_gram(A))):
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    out = fn(); torch.cuda.synchronize()
    peak = (torch.cuda.max_memory_allocated() - base) / 2 ** 30
    del out
    t0 = time.perf_counter()
    for _ in range(3):
        out = fn(); del out
    torch.cuda.synchronize()
    print(f'{name}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms, peak extra memory {peak:.1f} GiB', flush=True)
