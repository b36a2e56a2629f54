"""nif_eigh_kept_f64 (hand-written) against torch.linalg.eigh (rocSOLVER) on the normal matrices of the C1 layers:
random-init VGG11 design matrices are emulated by a ReLU'd low-rank-plus-noise X with dead columns, so the cutoff is
active.  Prints ms per call and the kept count; correctness is tests/test_gpu_kernels.py's job."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuron_interpolation_model_fusion_amd import ops

sizes = [int(a) for a in sys.argv[1:]] or [577, 1153, 1600, 2305, 4609]
for d in sizes:
    g = torch.Generator(device='cuda').manual_seed(d)
    m = max(2 * d, 6400)
    r = max(d // 3, 8)
    X = (torch.randn(m, r, device='cuda', generator=g) @ torch.randn(r, d, device='cuda', generator=g) / r ** 0.5
         + 0.05 * torch.randn(m, d, device='cuda', generator=g)).clamp_min(0)
    G, _, _ = ops.xty(X, None, acc='f64', norms=False)
    cutoff = (max(m, d) * 1.1920929e-07) ** 2
    for name, fn in (('nif_eigh_kept', lambda: ops.eigh_kept(G, cutoff)), ('torch.linalg.eigh', lambda: torch.linalg.eigh(G))):
        out = fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / reps
        kept = out[0].shape[0] if name.startswith('nif') else int((out[0] > cutoff * out[0][-1]).sum())
        print(f'd={d} {name}: {ms:.1f} ms, kept {kept}', flush=True)
