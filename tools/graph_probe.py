"""Bisect HIP-graph capture: which ingredient of the ALF fit step survives capture (argv[1] = case)."""
import sys, os
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuron_interpolation_model_fusion_amd import ops
case = sys.argv[1]
dev = torch.device('cuda')
net = nn.Sequential(nn.Flatten(), nn.Linear(64, 48)).to(dev)
x = torch.randn(16, 64, device=dev); y = torch.randn(16, 48, device=dev); w = torch.rand(48, device=dev)
tot = torch.zeros((), dtype=torch.float64, device=dev)

def step():
    for p in net.parameters(): p.grad = None
    if case == 'torch':
        loss = (((net(x) - y) ** 2) * w).mean(); loss.backward()
    elif case == 'fwd_only':
        with torch.no_grad(): loss = ops.weighted_mse(net(x), y, w)
    elif case == 'lib_only':
        with torch.no_grad(): loss = ops.weighted_mse(x, x * 0.5, None)
    elif case == 'cost':
        G = torch.rand(48, 48, dtype=torch.float64, device=dev); n = torch.rand(48, dtype=torch.float64, device=dev) + 2
        loss = ops.cost_dist(G, n, n).sum()
    else:
        loss = ops.weighted_mse(net(x), y, w); loss.backward()
    tot.copy_(loss.detach().double())

step(); step(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
print(case, 'captured', flush=True)
g.replay(); torch.cuda.synchronize()
print(case, 'replayed, loss', tot.item(), flush=True)
