"""HF on the vgg_tiny fixture with torch's conv and with the im2row + GEMM conv: per-layer differences of the cost
matrices / LS solutions / assignments, and how far apart the two assignments' totals are under each cost."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import load_npz, build_models  # noqa: E402
from neuron_interpolation_model_fusion_amd.fusion_algorithms import LSFusion  # noqa: E402
import contextlib, io  # noqa: E402

dev = torch.device('cuda', 0)
z = load_npz('vgg_tiny')
runs = {}
for flag in ('0', '1'):
    os.environ['NIF_CONV_GEMM'] = flag
    models, X = build_models('vgg_tiny', z, dev)
    ls = LSFusion(models[:2])
    with contextlib.redirect_stdout(io.StringIO()):
        ls.fuse(X, None, fusion_method='hungarian', neuron_importance_method='uniform')
    runs[flag] = [(i['cost'].cpu().numpy().astype(np.float64), i['col'].cpu().numpy(), i['W'].cpu().numpy()) for i in ls.last_layer_info]
for l, ((c0, k0, w0), (c1, k1, w1)) in enumerate(zip(runs['0'], runs['1'])):
    ref, kref = z[f'hf.cost{l}'].astype(np.float64), z[f'hf.col{l}']
    n = len(kref)
    ar = np.arange(n)
    tot = lambda c, k: c[ar, k].sum()
    print(f'layer {l}: n={n} |c0-ref|/max={np.abs(c0 - ref).max() / ref.max():.2e} |c1-ref|/max={np.abs(c1 - ref).max() / ref.max():.2e} '
          f'|c1-c0|/max={np.abs(c1 - c0).max() / ref.max():.2e} dW={np.abs(w1 - w0).max() / np.abs(w0).max():.2e} '
          f'col0==ref {(k0 == kref).all()} col1==ref {(k1 == kref).all()} '
          f'gap under ref cost: {(tot(ref, k1) - tot(ref, kref)) / tot(ref, kref):.3e} under c1: {(tot(c1, kref) - tot(c1, k1)) / tot(c1, k1):.3e}')
