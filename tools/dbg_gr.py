import sys, numpy as np, torch
sys.path.insert(0, '.')
from tests.helpers import load_npz, build_models
from neuron_interpolation_model_fusion_amd.fusion_algorithms import GitRebasin
from scipy.optimize import linear_sum_assignment
import io, contextlib
z = load_npz('vgg_tiny')
models, X = build_models('vgg_tiny', z, torch.device('cuda'))
gr = GitRebasin(models[0], models[1]); gr.cost_matrices = []
with contextlib.redirect_stdout(io.StringIO()):
    gr.fuse(X, None, 'uniform')
for l in range(8):
    c = gr.cost_matrices[l].cpu().numpy().astype(np.float64); ref = z[f'gr.cost{l}'].astype(np.float64)
    mine = gr.permutations[l].cpu().numpy(); gold = z[f'gr.col{l}']
    n = len(gold); ar = np.arange(n)
    _, sp_mine = linear_sum_assignment(c)
    print(l, 'maxrel', np.abs(c-ref).max()/ref.max(), 'same', (mine==gold).all(), 'scipy(mycost)==mine', (sp_mine==mine).all(),
          'cost_ref[gold]', ref[ar,gold].sum(), 'cost_ref[mine]', ref[ar,mine].sum(), 'cost_mine[gold]', c[ar,gold].sum(), 'cost_mine[mine]', c[ar,mine].sum())
