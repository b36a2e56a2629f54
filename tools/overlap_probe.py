"""Do a chip-wide Gram launch and the single-workgroup assignment kernel overlap when issued on two streams?"""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuron_interpolation_model_fusion_amd import ops

dev = torch.device('cuda')
g = torch.Generator(device=dev).manual_seed(0)
m, n, K = 325_000, 384, 20
A = torch.randn(m, n, device=dev, generator=g); B = torch.randn(m, n, device=dev, generator=g)
G, na, nb = ops.xty(A, B, acc='f32'); C = ops.cost_dist(G, na, nb)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def grams(k):
    with torch.cuda.stream(s1):
        for _ in range(k):
            ops.xty(A, B, acc='f32')

def laps(k):
    with torch.cuda.stream(s2):
        for _ in range(k):
            ops.lap_solve(C)

def timed(f):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter(); f(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3

tg = timed(lambda: grams(K)); tl = timed(lambda: laps(K))
tb = timed(lambda: (grams(K), laps(K)))
tb2 = timed(lambda: (laps(K), grams(K)))
def inter():
    for _ in range(K):
        grams(1); laps(1)
ti = timed(inter)
print(f'{K} Grams {tg:.1f} ms, {K} assignments {tl:.1f} ms, both (Grams issued first) {tb:.1f} ms, both (assignments first) {tb2:.1f} ms, '
      f'interleaved issue {ti:.1f} ms')
