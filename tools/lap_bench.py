"""LAP kernel on hard instances (uniform random costs; random-init VGG11 layer costs), both modes."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuron_interpolation_model_fusion_amd import ops

def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

def steps(n):
    """Dijkstra steps of the last solve: the kernel leaves the count in the first word of the workspace's jbest area."""
    a256 = lambda x: (x + 255) & ~255
    chunks = min((n + 15) // 16, 64)
    rpc = (n + chunks - 1) // chunks
    chunks = (n + rpc - 1) // rpc
    off = a256(chunks * n * 4) + 2 * a256(n * 8)
    return int(ops.lap_solve.last_workspace[off:off + 4].view(torch.int32).item())

def run(C, scipy_order):
    ms = t(lambda: ops.lap_solve(C, scipy_order='replay' if scipy_order else False))
    st = steps(C.shape[0])
    return f'{ms:.2f} ms / {st} steps = {ms * 1e3 / max(st, 1):.2f} us/step'

g = torch.Generator(device='cuda').manual_seed(0)
for n in (128, 256, 384, 512, 1024, 2048):
    C = torch.rand(n, n, device='cuda', generator=g)
    A = torch.randn(1600, n, device='cuda', generator=g).clamp_min(0)
    B = torch.randn(1600, n, device='cuda', generator=g).clamp_min(0)
    G, na, nb = ops.xty(A, B)
    Cd = ops.cost_dist(G, na, nb)
    print(f'n={n}: uniform fast {run(C, False)}, scipy-order {run(C, True)}; '
          f'unrelated-activations cdist fast {run(Cd, False)}, scipy-order {run(Cd, True)}', flush=True)

# planted instances (C3 shape: GR of two MLPs, hidden 8192, B = 4096; and the C2 width)
for n, m in ((384, 4096), (8192, 4096)):
    A = torch.randn(m, n, device='cuda', generator=g).clamp_min(0)
    perm = torch.randperm(n, device='cuda', generator=g)
    B = 0.9 * A[:, perm] + 0.3 * torch.randn(m, n, device='cuda', generator=g)
    G, na, nb = ops.xty(A, B)
    Cd = ops.cost_dist(G, na, nb)
    print(f'planted n={n}: fast {run(Cd, False)}, scipy-order {run(Cd, True)}', flush=True)
