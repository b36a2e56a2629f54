"""Probe: torch.linalg.eigh (interim eigensolver) timing in f64 at the VGG11 normal-equation sizes,
and torch f64 GEMM / cholesky rates for comparison."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

g = torch.Generator(device='cuda').manual_seed(0)
for d in (28, 257, 513, 577, 1153, 2305, 4609):
    X = torch.randn(2 * d, d, device='cuda', dtype=torch.float64, generator=g)
    A = X.T @ X
    te = t(lambda: torch.linalg.eigh(A), 2)
    tv = t(lambda: torch.linalg.eigvalsh(A), 2)
    tc = t(lambda: torch.linalg.cholesky(A), 2)
    tg = t(lambda: A @ A, 2)
    print(f'd={d}: eigh {te:.1f} ms, eigvalsh {tv:.1f} ms, cholesky {tc:.1f} ms, gemm f64 {tg:.2f} ms '
          f'({2 * d ** 3 / tg / 1e9:.1f} TF)', flush=True)
