"""ALF fit with graph replay on a small MLP, tracing every capture (argv: skip-eager-run flag)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuron_interpolation_model_fusion_amd.fusion_algorithms import ALFusion, ALFusionConfig
from neuron_interpolation_model_fusion_amd.fusion_algorithms import alf as alf_mod
from neuron_interpolation_model_fusion_amd.models import MLPNet
orig = alf_mod._GraphedBatch._capture
def traced(self, host, network, criterion):
    print('capture begin train=', self.train, 'x', tuple(self.x.shape), flush=True)
    orig(self, host, network, criterion)
    print('capture end', flush=True)
alf_mod._GraphedBatch._capture = traced
dev = torch.device('cuda')
torch.manual_seed(3)
models = [MLPNet(64, 48, 10, use_bias=True, use_activations=False).to(dev) for _ in range(2)]
X = torch.randn(1615, 64).to(dev)
for graphs in ((True,) if len(sys.argv) > 1 else (False, True)):
    cfg = ALFusionConfig(X=X, y=None, optimizer_per_level=['adam'] * 2, lr_per_level=[1e-3] * 2, weight_decay_per_level=[0.0] * 2,
                         epochs_per_level=[2] * 2, verbose=False, train_batch_size=16, use_scheduler=True)
    copies = [m.copy_model() for m in models]
    torch.manual_seed(11); np.random.seed(11)
    a = ALFusion(copies); a.use_hip_graphs = graphs
    fused = a.fuse(cfg)
    print('graphs', graphs, 'weight change', a.weight_change_per_level, flush=True)
