import cProfile, pstats, sys, os, io, contextlib, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from neuron_interpolation_model_fusion_amd.fusion_algorithms import GitRebasin, LSFusion
from neuron_interpolation_model_fusion_amd.models import VGG
from neuron_interpolation_model_fusion_amd.models.base_model import twin_copy
twin_copy.replay_init_draws = False
dev = torch.device('cuda')
models = []
for k in range(4):
    torch.manual_seed(100 + k)
    models.append(VGG('VGG11', 10, (3, 32, 32), batch_norm=False, use_bias=True, use_activations=False).eval().to(dev))
torch.manual_seed(0); np.random.seed(0)
X = torch.randn(400, 3, 32, 32, device=dev)
algo = sys.argv[1]
def run():
    with contextlib.redirect_stdout(io.StringIO()):
        if algo == 'gr':
            GitRebasin(models[0], models[1]).fuse(X, None, 'uniform')
        elif algo == 'kf':
            LSFusion(models).fuse(X, None, 'k_means')
        else:
            LSFusion(models[:2]).fuse(X, None, 'hungarian')
    torch.cuda.synchronize()
run(); run()
t0 = time.time(); run(); print('wall', time.time() - t0)
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28); print(s.getvalue()[:6000])
