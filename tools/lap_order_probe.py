"""Does the order in which the fast assignment solver takes its free rows matter?  Same instance, rows permuted."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuron_interpolation_model_fusion_amd import ops
def counters(n):
    a256 = lambda x: (x + 255) & ~255
    chunks = min((n + 15) // 16, 64); rpc = (n + chunks - 1) // chunks; chunks = (n + rpc - 1) // rpc
    off = a256(chunks * n * 4) + 2 * a256(n * 8)
    return ops.lap_solve.last_workspace[off:off + 12].view(torch.int32).tolist()
for n in (384, 2048):
    g = torch.Generator(device='cuda').manual_seed(n)
    A = torch.randn(4096, n, device='cuda', generator=g); B = torch.randn(4096, n, device='cuda', generator=g)
    G, na, nb = ops.xty(A, B); C = ops.cost_dist(G, na, nb)
    for trial in range(6):
        p = torch.arange(n, device='cuda') if trial == 0 else torch.randperm(n, device='cuda', generator=g)
        Cp = C[p].contiguous()
        ops.lap_solve(Cp); torch.cuda.synchronize()
        t0 = time.perf_counter(); col = ops.lap_solve(Cp); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        st, ps, fr = counters(n)
        print(f'n={n} trial {trial}: {dt*1e3:.2f} ms, {ps} passes, {fr} free, {st} steps, cost {float(Cp.double()[torch.arange(n), col.long()].sum()):.6f}', flush=True)
