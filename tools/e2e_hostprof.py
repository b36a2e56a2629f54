"""cProfile of one warmed-up end-to-end fusion (host side): where the wall-clock goes besides kernels."""
import cProfile, pstats, sys, os, io, contextlib
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuron_interpolation_model_fusion_amd.models import VGG
from neuron_interpolation_model_fusion_amd.fusion_algorithms import LSFusion, GitRebasin
algo = sys.argv[1] if len(sys.argv) > 1 else 'hf'
dev = torch.device('cuda')
models = []
for k in range(4):
    torch.manual_seed(100 + k)
    models.append(VGG('VGG11', 10, (3, 32, 32), batch_norm=False, use_bias=True, use_activations=False).eval().to(dev))
torch.manual_seed(0); np.random.seed(0)
X = torch.randn(400, 3, 32, 32, device=dev)
def run():
    with contextlib.redirect_stdout(io.StringIO()):
        if algo == 'hf':
            LSFusion(models[:2]).fuse(X, None, 'hungarian')
        elif algo == 'kf':
            LSFusion(models).fuse(X, None, 'k_means')
        else:
            GitRebasin(models[0], models[1]).fuse(X, None, 'uniform')
    torch.cuda.synchronize()
run()
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22); print(s.getvalue()[:6000])
