"""Gram kernel + slab reduction time over a set of shapes (wall clock of ops.xty, averaged)."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuron_interpolation_model_fusion_amd import ops
g = torch.Generator(device='cuda').manual_seed(0)
shapes = [(325000, 384, 'f32'), (162500, 384, 'f32'), (81250, 384, 'f32'), (40625, 384, 'f32'), (325000, 768, 'f32'), (25000, 6144, 'f32'),
          (4096, 8192, 'f32'), (1600, 512, 'f32'), (6400, 512, 'f32'), (25600, 256, 'f32'), (102400, 128, 'f32'), (409600, 64, 'f32'),
          (102400, 577, 'f64'), (25600, 1153, 'f64'), (6400, 2305, 'f64'), (6400, 4609, 'f64'), (5000, 1536, 'f32'), (20000, 200, 'f32')]
for m, n, acc in shapes:
    A = torch.randn(m, n, device='cuda', generator=g)
    sym = acc == 'f64' or n >= 6144
    B = None if sym else torch.randn(m, n, device='cuda', generator=g)
    split = acc == 'f32' and os.environ.get('SPLIT') == '1'
    for _ in range(2): out = ops.xty(A, B, acc=acc, split_bf16=split)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    it = 10
    for _ in range(it): ops.xty(A, B, acc=acc, split_bf16=split)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / it
    fl = m * n * (n + 1) if sym else 2.0 * m * n * n
    err = ''
    if split and m * n <= 4e8:  # accuracy against the f64-accumulated Gram
        ref = ops.xty(A, B, acc='f64')
        f32 = ops.xty(A, B, acc='f32')
        scale = ref[0].abs().max()
        err = (f'; max err vs f64 / max|G|: split {((out[0] - ref[0]).abs().max() / scale).item():.2e}, '
               f'f32 path {((f32[0] - ref[0]).abs().max() / scale).item():.2e}; norms {((out[1] - ref[1]).abs().max() / ref[1].abs().max()).item():.1e}')
    print(f'm={m} n={n} {acc} sym={sym}: {dt * 1e3:.3f} ms ({fl / dt / 1e12:.1f} TF incl. reduction){err}', flush=True)
