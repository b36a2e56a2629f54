"""torch.linalg.eigh (f64, device) per linalg backend; and f32 for comparison."""
import time, torch
print('magma available:', torch.cuda.has_magma)
for lib in ('default', 'cusolver', 'magma'):
    try:
        torch.backends.cuda.preferred_linalg_library(lib)
    except Exception as e:
        print(lib, 'unavailable', e); continue
    for d in (577, 1153, 2046, 3806):
        g = torch.Generator(device='cuda').manual_seed(0)
        X = torch.randn(2 * d, d, device='cuda', dtype=torch.float64, generator=g)
        G = X.T @ X
        try:
            torch.linalg.eigh(G); torch.cuda.synchronize()
            t0 = time.perf_counter(); torch.linalg.eigh(G); torch.cuda.synchronize()
            print(f'{lib} d={d}: {(time.perf_counter() - t0) * 1e3:.1f} ms', flush=True)
        except Exception as e:
            print(lib, d, 'failed', str(e)[:100]); break
