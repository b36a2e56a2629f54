"""Sweep of the augmenting-row-reduction knobs (pass budget as a multiple of n, chains left at which it stops, passes
a tied row waits) on unrelated-activation costs: total solve time, passes, Dijkstra steps."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuron_interpolation_model_fusion_amd import ops
def counters(n):
    a256 = lambda x: (x + 255) & ~255
    chunks = min((n + 15) // 16, 64); rpc = (n + chunks - 1) // chunks; chunks = (n + rpc - 1) // rpc
    off = a256(chunks * n * 4) + 2 * a256(n * 8)
    return ops.lap_solve.last_workspace[off:off + 12].view(torch.int32).tolist()
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for n in [int(a) for a in sys.argv[1:]] or [384, 1024, 2048]:
    g = torch.Generator(device='cuda').manual_seed(n)
    A = torch.randn(4096, n, device='cuda', generator=g); B = torch.randn(4096, n, device='cuda', generator=g)
    G, na, nb = ops.xty(A, B); C = ops.cost_dist(G, na, nb)
    for mult in (1, 2, 4):
        for mc in (3, 6):
            for mw in (0, 2, 4, 16):
                os.environ['NIF_LAP_ARR_PASSES'] = str(mult * n)
                os.environ['NIF_LAP_ARR_MIN_CHAINS'] = str(mc)
                os.environ['NIF_LAP_ARR_MAX_WAIT'] = str(mw)
                ms = timed(lambda: ops.lap_solve(C))
                st, ps, fr = counters(n)
                print(f'n={n} budget={mult}n min_chains={mc} max_wait={mw}: {ms:.2f} ms, {ps} passes, {fr} free, {st} steps', flush=True)
