"""k-means kernels at the C4 / C5 shapes, for rocprofv3 --kernel-trace --stats (argv[1]: c4 | c5a | c5b)."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuron_interpolation_model_fusion_amd import ops

which = sys.argv[1] if len(sys.argv) > 1 else 'c4'
models, h, m = {'c4': (2, 384, 325000), 'c5a': (8, 768, 25000), 'c5b': (8, 3072, 25000)}[which]
g = torch.Generator(device='cuda').manual_seed(0)
N, k = models * h, h
base = torch.randn(m, h, device='cuda', generator=g)
A = torch.cat([base[:, torch.randperm(h, device='cuda')] + 0.4 * torch.randn(m, h, device='cuda', generator=g)
               for _ in range(models)], dim=1).contiguous()
G, _, _ = ops.xty(A, None, norms=False)
nn = torch.diagonal(G).contiguous()
if os.environ.get('KM_G', 'f64') == 'f32':
    G = G.float()
w = torch.ones(N, device='cuda')
km = ops.KMeans(G, nn, w, k, gdiag=nn)
a = km.assign_to_neurons(0)
for _ in range(3):
    a, nearest, cost, kp = km.lloyd_step(a)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    a2, nearest, cost, kp = km.lloyd_step(a)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(10):
    T, _ = km.targets(A, a)
torch.cuda.synchronize()
t2 = time.perf_counter()
picks, _ = km.farthest_init(0)
torch.cuda.synchronize()
t3 = time.perf_counter()
print(f'{which} G {G.dtype}: N={N} k={k} m={m}: lloyd step {(t1 - t0) * 100:.3f} ms ({(4.0 * N * N + 4.0 * k * N) / (t1 - t0) * 10 / 1e9:.0f} GB/s), '
      f'targets {(t2 - t1) * 100:.3f} ms ({(4.0 * m * N + 4.0 * m * k) / (t2 - t1) * 10 / 1e9:.0f} GB/s), '
      f'farthest init {(t3 - t2) * 1e3:.1f} ms')
