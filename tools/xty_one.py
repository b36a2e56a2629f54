"""One Gram shape, a few launches (profiling target): python tools/xty_one.py m n [split] [sym]."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuron_interpolation_model_fusion_amd import ops
m, n = int(sys.argv[1]), int(sys.argv[2])
split = len(sys.argv) > 3 and sys.argv[3] == '1'
sym = len(sys.argv) > 4 and sys.argv[4] == '1'
g = torch.Generator(device='cuda').manual_seed(0)
A = torch.randn(m, n, device='cuda', generator=g)
B = None if sym else torch.randn(m, n, device='cuda', generator=g)
for _ in range(5):
    ops.xty(A, B, split_bf16=split)
torch.cuda.synchronize()
