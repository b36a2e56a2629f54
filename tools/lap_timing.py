"""Per-phase shader-clock breakdown of a Dijkstra step (needs the instrumented build of nif_lap.hip)."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuron_interpolation_model_fusion_amd import ops
g = torch.Generator(device='cuda').manual_seed(0)
for n in (128, 256, 512, 1024):
    A = torch.randn(1600, n, device='cuda', generator=g).clamp_min(0)
    B = torch.randn(1600, n, device='cuda', generator=g).clamp_min(0)
    G, na, nb = ops.xty(A, B)
    C = ops.cost_dist(G, na, nb)
    for so in (True,):
        ops.lap_solve(C, scipy_order='replay' if so else False); torch.cuda.synchronize()
        t0 = time.perf_counter(); ops.lap_solve(C, scipy_order='replay' if so else False); torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
        a256 = lambda x: (x + 255) & ~255
        chunks = min((n + 15) // 16, 64); rpc = (n + chunks - 1) // chunks; chunks = (n + rpc - 1) // rpc
        off = a256(chunks * n * 4) + 2 * a256(n * 8)
        st = ops.lap_solve.last_workspace[off:off + 28].view(torch.int32).cpu().tolist()
        steps = st[0]
        ph = [x * 16 / steps for x in st[1:7]]
        print(f'n={n} scipy={so}: {ms:.2f} ms, {steps} steps, {ms * 1e6 / steps:.0f} ns/step; cycles/step load {ph[5]:.0f}, relax {ph[0]:.0f}, wave-reduce {ph[1]:.0f}, '
              f'publish+barrier {ph[2]:.0f}, cross-wave {ph[3]:.0f}, bookkeeping {ph[4]:.0f}; sum {sum(ph):.0f}', flush=True)
