"""C4 shape: ALF neuron clustering + target build on every level of two random-init ViTs (CIFAR geometry:
32x32, patch 8 -> 65 tokens, hidden 384), B fusion samples; the SGD fit is excluded (epochs = 0) unless --epochs."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuron_interpolation_model_fusion_amd.models import ViT
from neuron_interpolation_model_fusion_amd.fusion_algorithms import ALFusion, ALFusionConfig

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=5000)
ap.add_argument('--epochs', type=int, default=0)
ap.add_argument('--depth', type=int, default=7)
a = ap.parse_args()
dev = torch.device('cuda')
models = []
for k in range(2):
    torch.manual_seed(100 + k)
    models.append(ViT(in_c=3, num_classes=100, img_size=32, patch=8, depth=a.depth, hidden=384, mlp_hidden=384, head=8).eval().to(dev))
torch.manual_seed(0)
X = torch.randn(a.batch, 3, 32, 32, device=dev)
nlev = len(models[0].get_ordered_trainable_named_levels)
alf = ALFusion(models)
alf.ni_scores = [{l: torch.ones(m.get_ordered_trainable_named_levels[l].output_width) for l in range(nlev)} for m in models]
cfg = ALFusionConfig(X=X, y=None, optimizer_per_level=['adam'] * nlev, lr_per_level=[1e-3] * nlev,
                     weight_decay_per_level=[0.0] * nlev, epochs_per_level=[a.epochs] * nlev, verbose=False,
                     normalize_acts_for_kmeans=True, init_kmeans_with_2nd_acts=False, use_weighted_loss=False)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    per = []
    for level in range(nlev - 1):
        t1 = time.time()
        with torch.no_grad():  # the level's input, carried forward one level at a time
            level_X = X if level == 0 else models[0].forward_level(level_X, level - 1)
        k = models[0].get_ordered_trainable_named_levels[level].output_width
        T, assign = alf._k_means_v2(level, k, cfg, level_X)
        torch.cuda.synchronize(); per.append(time.time() - t1)
    dt = time.time() - t0
    print(f'ALF k-means + targets, ViT depth {a.depth} x2, B={a.batch} (m={a.batch * 65} rows), run {rep}: {dt:.3f} s over {nlev - 1} levels '
          f'({(nlev - 1) / dt:.1f} levels/s); per level ms: {[round(p * 1e3, 1) for p in per]}', flush=True)
if a.epochs > 0:
    torch.cuda.synchronize(); t0 = time.time()
    fused = alf.fuse(cfg)
    torch.cuda.synchronize()
    print(f'full ALF fuse with {a.epochs} epoch(s) per level: {time.time() - t0:.2f} s')
