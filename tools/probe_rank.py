"""Which C1 layers actually hit pinv's truncation? prints (m, d, kept rank, lambda_min/lambda_max vs cutoff)."""
import sys, os, io, contextlib
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuron_interpolation_model_fusion_amd import ops
from neuron_interpolation_model_fusion_amd.models import VGG
from neuron_interpolation_model_fusion_amd.fusion_algorithms import LSFusion
orig = ops.ls_solve
log = []
def traced(X, A, normalize=False):
    out = orig(X, A, normalize)
    m, d = X.shape
    zero_cols = int((X.abs().amax(0) == 0).sum().item())
    zero_rows = int((X.abs().amax(1) == 0).sum().item())
    log.append((m, d, min(m, d), out['rank'], zero_cols, zero_rows))
    return out
ops.ls_solve = traced
import neuron_interpolation_model_fusion_amd.fusion_algorithms.hf_kf as hk
hk.ops.ls_solve = traced
dev = torch.device('cuda')
models = []
for k in range(2):
    torch.manual_seed(100 + k)
    models.append(VGG('VGG11', 10, (3, 32, 32), batch_norm=False, use_bias=True, use_activations=False).eval().to(dev))
torch.manual_seed(0)
X = torch.randn(400, 3, 32, 32, device=dev)
with contextlib.redirect_stdout(io.StringIO()):
    LSFusion(models).fuse(X, None, 'hungarian')
for m, d, full, rank, zc, zr in log:
    print(f'm={m} d={d} full rank={full} kept={rank} zero columns={zc} zero rows={zr} {"TRUNCATED" if rank < full else ""}')
