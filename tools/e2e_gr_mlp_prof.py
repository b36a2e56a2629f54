"""cProfile of the warmed-up C3 fusion (host side)."""
import cProfile, pstats, io, contextlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuron_interpolation_model_fusion_amd.models import MLPNet
from neuron_interpolation_model_fusion_amd.fusion_algorithms import GitRebasin
dev = torch.device('cuda'); H, B = 8192, 4096
torch.manual_seed(0)
m0 = MLPNet(784, H, 10, use_bias=True, use_activations=False).to(dev)
m1 = m0.copy_model()
perm = torch.randperm(H, device=dev)
with torch.no_grad():
    fc1, fc2 = m1.level1[1], m1.level2[-1]
    fc1.weight.copy_(fc1.weight[perm] + 0.02 * torch.randn_like(fc1.weight)); fc1.bias.copy_(fc1.bias[perm]); fc2.weight.copy_(fc2.weight[:, perm])
X = torch.randn(B, 1, 28, 28, device=dev)
def run():
    with contextlib.redirect_stdout(io.StringIO()):
        GitRebasin(m0, m1).fuse(X, None, 'uniform')
    torch.cuda.synchronize()
run()
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14); print(s.getvalue()[:3500])
