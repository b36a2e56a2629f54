"""How often is the optimal matching of a C1 layer ambiguous within tolerance?  For every GR layer of two random-init
VGG11s: solve, then look for an alternating cycle among the arcs whose reduced cost is below tol."""
import sys, os, io, contextlib
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuron_interpolation_model_fusion_amd import ops
from neuron_interpolation_model_fusion_amd.models import VGG
from neuron_interpolation_model_fusion_amd.fusion_algorithms import GitRebasin

def has_cycle(n, edges):
    adj = [[] for _ in range(n)]
    indeg = [0] * n
    for a, b in edges:
        adj[a].append(b); indeg[b] += 1
    stack = [i for i in range(n) if indeg[i] == 0]
    seen = 0
    while stack:
        x = stack.pop(); seen += 1
        for y in adj[x]:
            indeg[y] -= 1
            if indeg[y] == 0: stack.append(y)
    return seen < n

dev = torch.device('cuda')
models = []
for k in range(2):
    torch.manual_seed(100 + k)
    models.append(VGG('VGG11', 10, (3, 32, 32), batch_norm=False, use_bias=True, use_activations=False).eval().to(dev))
torch.manual_seed(0)
X = torch.randn(400, 3, 32, 32, device=dev)
gr = GitRebasin(models[0], models[1])
gr.cost_matrices = []
with contextlib.redirect_stdout(io.StringIO()):
    gr.fuse(X, None, 'uniform')
for l, C in enumerate(gr.cost_matrices):
    C = C.float().contiguous()
    n = C.shape[0]
    col, row, u, v = ops.lap_solve(C, duals=True)
    rc = C.double() - u[:, None] - v[None, :]
    rc[torch.arange(n, device=dev), col.long()] = float('inf')
    scale = float(C.max())
    for tol in (1e-9, 1e-7, 1e-6):
        ii, jj = torch.nonzero(rc < tol * scale, as_tuple=True)
        edges = list(zip(ii.tolist(), row.long()[jj].tolist()))
        print(f'layer {l} n={n} tol={tol:g}: {len(edges)} tight arcs, ambiguous={has_cycle(n, edges)}', end='; ')
    dup = n - torch.unique(C, dim=1).shape[1]
    print(f'duplicate columns {dup}')
