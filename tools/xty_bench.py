"""Micro-benchmark of the cross-Gram kernel at one shape (for rocprofv3 --pmc passes)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuron_interpolation_model_fusion_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--m', type=int, default=325000)
ap.add_argument('--n', type=int, default=384)
ap.add_argument('--acc', default='f32')
ap.add_argument('--iters', type=int, default=5)
ap.add_argument('--sym', action='store_true')
ap.add_argument('--relu', action='store_true')
a = ap.parse_args()
g = torch.Generator(device='cuda').manual_seed(0)
A = torch.randn(a.m, a.n, device='cuda', generator=g)
B = None if a.sym else torch.randn(a.m, a.n, device='cuda', generator=g)
ops.xty(A, B, acc=a.acc, relu_a=a.relu, relu_b=a.relu)
torch.cuda.synchronize()
ops.profile_enable(True)
for _ in range(a.iters):
    ops.xty(A, B, acc=a.acc, relu_a=a.relu, relu_b=a.relu)
torch.cuda.synchronize()
ms, cnt = ops.profile_collect(ops.PROF_XTY)
flops = (a.m * a.n * (a.n + 1)) if a.sym else 2.0 * a.m * a.n * a.n
print(f'xty m={a.m} n={a.n} acc={a.acc} sym={a.sym} relu={a.relu}: {ms / cnt:.4f} ms/launch, {flops / (ms / cnt) / 1e9:.1f} TF')
