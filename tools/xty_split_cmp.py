"""Split-bf16 Gram: kernel-level timing of a few shapes (HIP events inside nif_xty), accuracy vs the f64 Gram."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuron_interpolation_model_fusion_amd import ops
g = torch.Generator(device='cuda').manual_seed(0)
for m, n, sym in [(325000, 384, False), (40625, 384, False), (325000, 768, False), (325000, 768, True), (25000, 6144, True), (81250, 1536, False)]:
    A = torch.randn(m, n, device='cuda', generator=g)
    B = None if sym else torch.randn(m, n, device='cuda', generator=g)
    out = []
    for split in (False, True):
        for _ in range(3): G = ops.xty(A, B, split_bf16=split)[0]
        torch.cuda.synchronize()
        ops.profile_collect(ops.PROF_XTY); ops.profile_enable(True)
        for _ in range(10): ops.xty(A, B, split_bf16=split)
        torch.cuda.synchronize(); ops.profile_enable(False)
        ms, cnt = ops.profile_collect(ops.PROF_XTY)
        fl = m * n * (n + 1) if sym else 2.0 * m * n * n
        out.append((ms / cnt, fl / (ms / cnt * 1e-3) / 1e12, G))
    err = ''
    if m * n <= 2e8:
        ref = ops.xty(A, B, acc='f64')[0]
        err = f' err split {((out[1][2] - ref).abs().max() / ref.abs().max()).item():.1e} f32 {((out[0][2] - ref).abs().max() / ref.abs().max()).item():.1e}'
    print(f'm={m} n={n} sym={sym}: f32 kernel {out[0][0]:.3f} ms ({out[0][1]:.0f} TF), split-bf16 {out[1][0]:.3f} ms ({out[1][1]:.0f} TF){err}', flush=True)
