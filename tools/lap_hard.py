"""Assignment on HARD instances (cost = distances between unrelated activations: the greedy start leaves a third of
the rows free): the augmenting-row-reduction phase off / on, Dijkstra steps and passes from the kernel's counters,
optimal cost against scipy on this host's CPU.  Usage: python tools/lap_hard.py [sizes...]"""
import os
import sys
import time

import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuron_interpolation_model_fusion_amd import ops  # noqa: E402


def counters(n):
    a256 = lambda x: (x + 255) & ~255
    chunks = min((n + 15) // 16, 64)
    rpc = (n + chunks - 1) // chunks
    chunks = (n + rpc - 1) // rpc
    off = a256(chunks * n * 4) + 2 * a256(n * 8)
    return ops.lap_solve.last_workspace[off:off + 12].view(torch.int32).tolist()  # steps, passes, free after ARR


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def cost(n, m, seed, kind):
    g = torch.Generator(device='cuda').manual_seed(seed)
    if kind == 'uniform':
        return torch.rand(n, n, device='cuda', generator=g)
    A = torch.randn(m, n, device='cuda', generator=g)
    B = torch.randn(m, n, device='cuda', generator=g)
    if kind == 'relu':
        A, B = A.clamp_min(0), B.clamp_min(0)
    G, na, nb = ops.xty(A, B)
    return ops.cost_dist(G, na, nb)


sizes = [int(a) for a in sys.argv[1:]] or [128, 384, 512, 1024, 2048]
for n in sizes:
    for kind in ('acts', 'relu', 'uniform'):
        C = cost(n, 4096 if n <= 2048 else 2048, n, kind)
        line = f'n={n} {kind}:'
        ref_cost = None
        if n <= 4096 or (kind == 'acts' and os.environ.get('LAP_HARD_SCIPY_ALL') == '1'):
            t0 = time.perf_counter()
            r, c = linear_sum_assignment(C.cpu().numpy())
            t_sp = (time.perf_counter() - t0) * 1e3
            ref_cost = float(C.cpu().numpy()[r, c].astype(np.float64).sum())
            line += f' scipy {t_sp:.1f} ms;'
        for label, env in (('arr off', '-1'), ('arr on', '0')):
            if n > 2048 and env == '-1':
                continue
            os.environ['NIF_LAP_ARR_PASSES'] = env
            ms = timed(lambda: ops.lap_solve(C), 3 if n <= 2048 else 1)
            col = ops.lap_solve(C)
            st, ps, fr = counters(n)
            got = float(C.cpu().numpy()[np.arange(n), col.cpu().numpy()].astype(np.float64).sum())
            ok = '' if ref_cost is None else (' optimal' if abs(got - ref_cost) <= 1e-9 * abs(ref_cost) else f' COST {got} vs {ref_cost}')
            line += f' [{label}: {ms:.2f} ms, {st} steps, {ps} passes, {fr} free after]{ok}'
        print(line, flush=True)
os.environ.pop('NIF_LAP_ARR_PASSES', None)
