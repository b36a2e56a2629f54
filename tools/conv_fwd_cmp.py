"""VGG11 conv levels on a CIFAR-sized batch: torch's conv (MIOpen picks the solver) next to im2row + one GEMM."""
import sys, os, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuron_interpolation_model_fusion_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 400
shapes = [(3, 64, 32), (64, 128, 16), (128, 256, 8), (256, 256, 8), (256, 512, 4), (512, 512, 4), (512, 512, 2), (512, 512, 2)]
g = torch.Generator(device='cuda').manual_seed(0)

def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        y = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, y

with torch.no_grad():
    for cin, cout, hw in shapes:
        x = torch.randn(B, cin, hw, hw, device='cuda', generator=g)
        w = torch.randn(cout, cin, 3, 3, device='cuda', generator=g) / (3 * cin ** 0.5)
        b = torch.randn(cout, device='cuda', generator=g)
        t_ref, y_ref = timed(lambda: F.conv2d(x, w, b, padding=1))
        def gemm():
            Xp = ops.design_matrix(x, (3, 3), 1, 1, ones=False)
            return torch.addmm(b, Xp, w.view(cout, -1).t()).view(B, hw, hw, cout).permute(0, 3, 1, 2)
        t_g, y_g = timed(gemm)
        err = float((y_g - y_ref).abs().max() / y_ref.abs().max())
        print(f'{cin:4d}->{cout:4d} @{hw:2d}x{hw:<2d} B={B}: torch conv {t_ref:8.3f} ms   im2row+GEMM {t_g:7.3f} ms   rel diff {err:.1e}', flush=True)
