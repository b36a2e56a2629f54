#!/usr/bin/env python3
"""Mint golden vectors by importing the reference (CPU, this container only).

Run from the repo root:   python tests/golden/make_golden.py

The reference lives at /root/reference (read-only) and never travels to the GPU
box; only the *outputs* written here (small ``.npz`` files of inputs and expected
outputs) are committed.  ``captum`` / ``torchsummary`` are absent from this image
and are replaced by empty stub modules (only the ``'uniform'`` importance path,
which never touches captum, is exercised) -- see SURVEY.md section 8(c).

``fusion_algorithms/otfusion.py`` does ``import ot`` and POT is not installed: OTFusion
parity is *unpinned* at the ``ot.emd`` boundary (otfusion.py:32).  Everything else of that
file -- marginals, edge alignment through the previous plan, both ground metrics, the
barycentric projection, the averaging, the conv / conv-to-linear reshapes -- is pinned by
``otf.npz``: the reference module is imported with a stand-in ``ot`` module whose ``emd`` is
an exact solver (scipy: assignment for uniform square problems, HiGHS for the transport
LP otherwise).  Where the optimal plan is unique that is what POT's network simplex
returns too; ``otf.<tag>.<mode>.dup<l>_<k>`` marks the problems whose cost matrix has
duplicated rows or columns (dead units of random-init nets), where it is not.

What is recorded, per fixture (keys are flat strings inside one npz):
  sd{k}.<param>               state_dict of input model k (reference class, seeded init or real ckpt)
  X                           fusion samples
  gr.cost{l}, gr.col{l}       GitRebasin cost matrix + LSA column indices per layer
  gr.fused.<param>            GitRebasin fused weights
  hf.w0_{l}, hf.w1_{l}        pinv(X) @ acts per layer (LS refit)
  hf.cost{l}, hf.col{l}       HF cost matrix + LSA column indices
  hf.fused.<param>
  kf{M}.assign{l}             KF best assignment per layer (M models), under np.random.seed(SEED)
  kf{M}.fused.<param>
  alf.T.{norm}{init2}, alf.assign.{norm}{init2}   ALFusion._k_means_v2 at level 0
  (alf_kd.npz) head_w, grp.T.{hw}{kd}, kd{A,B,C,D}.fused.<param>   last-level grouping / criteria / KD pass (alf.py:398-454)
  (otf.npz) <tag>.sd{k}.<param>, <tag>.X, <tag>.<mode>.cost{l}_{k}, .T{l}_{k}, .dup{l}_{k}, <tag>.<mode>.fused.<param>
                              OTFusion.fuse_models on three fixtures x {acts, wts}; models[:-1] aligned to models[-1]
"""
import contextlib
import io
import os
import sys
import types

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))
SEED = 1234


def _import_reference():
    sys.path.insert(0, REF)
    for name in ['captum', 'captum.attr', 'torchsummary']:
        sys.modules[name] = types.ModuleType(name)
    for c in ['LayerConductance', 'LayerDeepLift', 'LayerFeatureAblation']:
        setattr(sys.modules['captum.attr'], c, type(c, (), {}))
    sys.modules['torchsummary'].summary = lambda *a, **k: None


_import_reference()

import numpy as np  # noqa: E402
import scipy  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from fusion_algorithms import alf as ref_alf  # noqa: E402
from fusion_algorithms import git_rebasin as ref_gr  # noqa: E402
from fusion_algorithms import hf_kf as ref_hf  # noqa: E402
from models.mlp import MLPNet  # noqa: E402
from models.vgg import VGG  # noqa: E402
from models.vit import ViT  # noqa: E402

torch.set_num_threads(8)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def sd_to(out, prefix, model):
    for k, v in model.state_dict().items():
        out[f'{prefix}.{k}'] = v.detach().cpu().numpy()


class Recorder:
    """Wrap module-level callables of the reference to log their inputs/outputs."""

    def __init__(self):
        self.lsa = []          # (cost, col)
        self.hf_cost_args = []  # (w0, w1)
        self.centroid_calls = []  # assignment of every _calculate_centroids call

    def wrap_lsa(self, mod):
        orig = mod.linear_sum_assignment

        def f(cost):
            r, c = orig(cost)
            self.lsa.append((np.array(cost, dtype=np.float32), np.array(c, dtype=np.int64)))
            return r, c
        mod.linear_sum_assignment = f
        return orig

    def wrap_hf_cost(self):
        orig = ref_hf.LSFusion._get_cost_matrix_vectorized_chunked

        def f(X, acts0, acts1, w0, w1, s0, s1, *a, **k):
            self.hf_cost_args.append((w0.numpy().copy(), w1.numpy().copy()))
            return orig(X, acts0, acts1, w0, w1, s0, s1, *a, **k)
        ref_hf.LSFusion._get_cost_matrix_vectorized_chunked = staticmethod(f)
        return orig

    def wrap_centroids(self):
        orig = ref_hf.LSFusion._calculate_centroids

        def f(output_size, assignment, X, w, s):
            self.centroid_calls.append(np.array(torch.as_tensor(assignment).cpu().numpy(), dtype=np.int64))
            return orig(output_size, assignment, X, w, s)
        ref_hf.LSFusion._calculate_centroids = staticmethod(f)
        return orig


def run_gr(out, models, X):
    rec = Recorder()
    orig = rec.wrap_lsa(ref_gr)
    fused = quiet(ref_gr.GitRebasin(models[0], models[1]).fuse, X, None, 'uniform')
    ref_gr.linear_sum_assignment = orig
    for l, (cost, col) in enumerate(rec.lsa):
        out[f'gr.cost{l}'] = cost
        out[f'gr.col{l}'] = col
    sd_to(out, 'gr.fused', fused)


def run_hf(out, models, X, norm_acts=False):
    rec = Recorder()
    o1 = rec.wrap_lsa(ref_hf)
    o2 = rec.wrap_hf_cost()
    fused = quiet(ref_hf.LSFusion(models[:2]).fuse, X, None, fusion_method='hungarian',
                  neuron_importance_method='uniform', norm_acts=norm_acts)
    ref_hf.linear_sum_assignment = o1
    ref_hf.LSFusion._get_cost_matrix_vectorized_chunked = staticmethod(o2)
    tag = 'hfn' if norm_acts else 'hf'
    for l, (cost, col) in enumerate(rec.lsa):
        out[f'{tag}.cost{l}'] = cost
        out[f'{tag}.col{l}'] = col
    for l, (w0, w1) in enumerate(rec.hf_cost_args):
        out[f'{tag}.w0_{l}'] = w0
        out[f'{tag}.w1_{l}'] = w1
    sd_to(out, f'{tag}.fused', fused)


def run_kf(out, models, X):
    M = len(models)
    rec = Recorder()
    o = rec.wrap_centroids()
    # mark layer boundaries: the last _calculate_centroids call inside one
    # _fuse_layer_k_means call receives best_assignment (hf_kf.py:494)
    ends = []
    orig_layer = ref_hf.LSFusion._fuse_layer_k_means

    def layer_wrapper(self, *a, **k):
        r = orig_layer(self, *a, **k)
        ends.append(len(rec.centroid_calls))
        return r
    ref_hf.LSFusion._fuse_layer_k_means = layer_wrapper
    np.random.seed(SEED)
    fused = quiet(ref_hf.LSFusion(models).fuse, X, None, fusion_method='k_means',
                  neuron_importance_method='uniform')
    ref_hf.LSFusion._calculate_centroids = staticmethod(o)
    ref_hf.LSFusion._fuse_layer_k_means = orig_layer
    prev = 0
    ncalls = []
    for l, e in enumerate(ends):
        out[f'kf{M}.assign{l}'] = rec.centroid_calls[e - 1]
        ncalls.append(e - prev)
        prev = e
    out[f'kf{M}.ncalls'] = np.array(ncalls, dtype=np.int64)
    sd_to(out, f'kf{M}.fused', fused)


def run_alf_kmeans(out, models, X, level_idx=0, align_skip=False):
    alf = ref_alf.ALFusion(models)
    nlev = len(models[0].get_ordered_trainable_named_levels)
    alf.ni_scores = [{l: torch.ones(models[k].get_ordered_trainable_named_levels[l].output_width)
                      for l in range(nlev)} for k in range(len(models))]
    width = models[0].get_ordered_trainable_named_levels[level_idx].output_width
    for norm in (False, True):
        for init2 in (False, True):
            cfg = ref_alf.ALFusionConfig(
                X=X, y=None, optimizer_per_level=['adam'] * nlev, lr_per_level=[1e-3] * nlev,
                weight_decay_per_level=[0.0] * nlev, epochs_per_level=[1] * nlev,
                normalize_acts_for_kmeans=norm, init_kmeans_with_2nd_acts=init2, verbose=False,
                align_skip_connections=align_skip)
            level_X = X
            if align_skip:  # the level input = previous level's output of the (un-fused) first model
                with torch.no_grad():
                    level_X = models[0].forward_until_level(X, level_idx - 1)
            with torch.no_grad():
                T, assign = quiet(alf._k_means_v2, level_idx, width, cfg, level_X)
            tag = f'{int(norm)}{int(init2)}'
            out[f'alf.T.{tag}'] = T.numpy()
            out[f'alf.assign.{tag}'] = assign.numpy().astype(np.int64)


def run_alf_fuse(out, models, X, epochs=3):
    """End-to-end ALFusion.fuse with a short plain-SGD fit (torch RNG seeded).  SGD, not Adam: Adam's
    first steps are lr * sign(gradient), so float noise on near-zero gradients moves a weight by a full
    lr and a CPU-minted golden could not be compared with a GPU run at a tight tolerance."""
    nlev = len(models[0].get_ordered_trainable_named_levels)
    cfg = ref_alf.ALFusionConfig(
        X=X, y=None, optimizer_per_level=['sgd'] * nlev, lr_per_level=[5e-2] * nlev,
        weight_decay_per_level=[0.0] * nlev, epochs_per_level=[epochs] * nlev,
        normalize_acts_for_kmeans=True, init_kmeans_with_2nd_acts=False, verbose=False)
    torch.manual_seed(SEED)
    alf = ref_alf.ALFusion(models)
    import tqdm  # noqa: F401
    with contextlib.redirect_stderr(io.StringIO()):
        fused = quiet(alf.fuse, cfg)
    sd_to(out, 'alf.fused', fused)
    out['alf.weight_change'] = np.array(alf.weight_change_per_level, dtype=np.float64)


def meta(out):
    out['meta.versions'] = np.array([f'torch={torch.__version__}', f'numpy={np.__version__}',
                                     f'scipy={scipy.__version__}'])
    out['meta.seed'] = np.array([SEED])


def fixture_mlp_small():
    out = {}
    models = []
    for k in range(4):
        torch.manual_seed(100 + k)
        models.append(MLPNet(20, 32, 10, use_bias=True, use_activations=False).eval())
    torch.manual_seed(0)
    X = torch.randn(48, 1, 4, 5)
    out['X'] = X.numpy()
    for k, m in enumerate(models):
        sd_to(out, f'sd{k}', m)
    run_gr(out, models, X)
    run_hf(out, models, X)
    run_hf(out, models, X, norm_acts=True)
    run_kf(out, models[:2], X)
    run_kf(out, models, X)
    run_alf_kmeans(out, models[:2], X)
    run_alf_fuse(out, [m.copy_model() for m in models[:2]], X)
    meta(out)
    np.savez_compressed(os.path.join(OUT, 'mlp_small.npz'), **out)
    return out


def fixture_mlp_wide():
    """Over-determined MLP (m > d) with a wider hidden layer: exercises the m > d LS refit."""
    out = {}
    models = []
    for k in range(3):
        torch.manual_seed(200 + k)
        models.append(MLPNet(24, 96, 10, use_bias=True, use_activations=False).eval())
    torch.manual_seed(1)
    X = torch.randn(256, 1, 4, 6)
    out['X'] = X.numpy()
    for k, m in enumerate(models):
        sd_to(out, f'sd{k}', m)
    run_gr(out, models, X)
    run_hf(out, models, X)
    run_kf(out, models, X)
    run_alf_kmeans(out, models[:2], X)
    meta(out)
    np.savez_compressed(os.path.join(OUT, 'mlp_wide.npz'), **out)
    return out


def fixture_vgg_tiny():
    out = {}
    # a narrow configuration of the reference's own VGG class (widths are data, not code)
    VGG.cfg['VGG_tiny'] = [8, 'M', 16, 'M', 16, 16, 'M', 32, 32, 'M', 32, 32, 'M']
    models = []
    for k in range(3):
        torch.manual_seed(300 + k)
        models.append(VGG('VGG_tiny', 10, (3, 32, 32), batch_norm=False, use_bias=True,
                          use_activations=False).eval())
    torch.manual_seed(2)
    X = torch.randn(6, 3, 32, 32)
    out['X'] = X.numpy()
    out['cfg'] = np.array([str(c) for c in VGG.cfg['VGG_tiny']])
    for k, m in enumerate(models):
        sd_to(out, f'sd{k}', m)
    run_gr(out, models, X)
    run_hf(out, models, X)
    run_kf(out, models, X)
    run_alf_kmeans(out, models[:2], X, level_idx=2)
    meta(out)
    np.savez_compressed(os.path.join(OUT, 'vgg_tiny.npz'), **out)
    return out


def fixture_vit_tiny():
    """Two narrow ViTs (reference class): ALF k-means on a transformer level [B, hidden, seq], with
    the target/input re-alignment of skip connections (alf.py:290-294) switched on."""
    out = {}
    kw = dict(in_c=3, num_classes=10, img_size=16, patch=4, dropout=0.0, depth=3, hidden=16, mlp_hidden=32, head=2)
    models = []
    for k in range(2):
        torch.manual_seed(400 + k)
        models.append(ViT(**kw).eval())
    torch.manual_seed(5)
    X = torch.randn(12, 3, 16, 16)
    out['X'] = X.numpy()
    out['vit_kwargs'] = np.array([f'{a}={b}' for a, b in kw.items()])
    for k, m in enumerate(models):
        sd_to(out, f'sd{k}', m)
    run_alf_kmeans(out, models, X, level_idx=1, align_skip=True)
    meta(out)
    np.savez_compressed(os.path.join(OUT, 'vit_tiny.npz'), **out)
    return out


def fixture_mlp_real():
    """The reference's own MNIST split-by-2 checkpoints (real trained weights), synthetic X."""
    out = {}
    torch.serialization.add_safe_globals([MLPNet, nn.Linear, nn.Sequential, nn.Flatten, nn.ReLU, set])
    base = os.path.join(REF, 'example-base-models/non-iid/MLPs/MNIST/split-by-2/seed1')
    models = [torch.load(os.path.join(base, f'model_{i}.pt'), weights_only=True, map_location='cpu').eval()
              for i in range(2)]
    torch.manual_seed(3)
    X = torch.randn(400, 1, 28, 28)
    out['X'] = X.numpy().astype(np.float16)  # halves the fixture; tests up-cast the stored fp16 values
    X = torch.from_numpy(out['X'].astype(np.float32))
    for k, m in enumerate(models):
        sd_to(out, f'sd{k}', m)
    run_gr(out, models, X)
    run_hf(out, models, X)
    # keep the fixture small: drop the big LS solutions (785 x 256 each); cost + perm + fused stay
    for key in list(out):
        if key.startswith('hf.w'):
            del out[key]
    meta(out)
    np.savez_compressed(os.path.join(OUT, 'mlp_real.npz'), **out)
    return out


def fixture_alf_kd():
    """Last level of ALF (alf.py:398-454) and the head-only KD pass of run_alf.py:189-199: `_last_level_grouping`
    with / without `cls_head_weights` and `use_kd_on_head`, and `fuse` with `skip_levels = range(L - 1)` under each
    criterion branch of `_get_criterion` (KLDivLoss, head-weighted squared error, plain MSE on softened targets)."""
    out = {}
    models = []
    for k in range(2):
        torch.manual_seed(500 + k)
        models.append(MLPNet(20, 32, 10, use_bias=True, use_activations=False).eval())
    torch.manual_seed(7)
    X = torch.randn(64, 1, 4, 5)
    shares = torch.rand(2, 10, dtype=torch.float64) + 0.1
    head_w = shares / shares.sum(dim=0, keepdim=True)   # what get_classifier_head_weights returns: float64 shares
    out['X'] = X.numpy()
    out['head_w'] = head_w.numpy()
    for k, m in enumerate(models):
        sd_to(out, f'sd{k}', m)
    nlev = len(models[0].get_ordered_trainable_named_levels)

    def config(**kw):
        base = dict(X=X, y=None, optimizer_per_level=['sgd'] * nlev, lr_per_level=[5e-2] * nlev,
                    weight_decay_per_level=[0.0] * nlev, epochs_per_level=[3] * nlev, verbose=False)
        base.update(kw)
        return ref_alf.ALFusionConfig(**base)

    alf = ref_alf.ALFusion(models)
    for hw in (0, 1):
        for kd in (0, 1):
            cfg = config(cls_head_weights=head_w if hw else None, use_kd_on_head=bool(kd), kd_temperature=2.0)
            T, assign = alf._last_level_grouping(10, cfg)
            out[f'grp.T.{hw}{kd}'] = T.numpy()
            out[f'grp.assign.{hw}{kd}'] = assign.numpy().astype(np.int64)

    variants = {
        # run_alf.py --use_weighted_loss --run_last_layer_kd --use_kl_kd_on_head: KLDivLoss branch (alf.py:438-439)
        'kdA': dict(use_weighted_loss=True, cls_head_weights=head_w, use_kd_on_head=True, kd_temperature=2.0),
        # head weights without KD: importance-weighted squared error on the head (alf.py:441-454)
        'kdB': dict(use_weighted_loss=True, cls_head_weights=head_w, use_kd_on_head=False),
        # KD targets without the weighted loss: nn.MSELoss on the softened probabilities (alf.py:434-435)
        'kdC': dict(use_weighted_loss=False, cls_head_weights=None, use_kd_on_head=True, kd_temperature=2.0),
        # weighted loss requested but no head weights: still nn.MSELoss at the last level (alf.py:434)
        'kdD': dict(use_weighted_loss=True, cls_head_weights=None, use_kd_on_head=True, kd_temperature=3.0),
    }
    for tag, kw in variants.items():
        cfg = config(skip_levels=list(range(nlev - 1)), **kw)
        copies = [m.copy_model() for m in models]
        torch.manual_seed(SEED)
        a = ref_alf.ALFusion(copies)
        with contextlib.redirect_stderr(io.StringIO()):
            fused = quiet(a.fuse, cfg)
        sd_to(out, f'{tag}.fused', fused)
        out[f'{tag}.weight_change'] = np.array(a.weight_change_per_level, dtype=np.float64)
    meta(out)
    np.savez_compressed(os.path.join(OUT, 'alf_kd.npz'), **out)
    return out


def _install_ot_stand_in():
    """`import ot` for the reference's otfusion.py: only ot.emd is used (otfusion.py:32).  Exact optimum of the same
    transport problem by scipy; returns a tensor like POT's torch backend does (type of the inputs)."""
    from scipy.optimize import linear_sum_assignment, linprog
    mod = types.ModuleType('ot')

    def emd(a, b, M, *args, **kwargs):
        a64, b64 = a.detach().double().numpy(), b.detach().double().numpy()
        C = M.detach().double().numpy()
        n0, n1 = C.shape
        if n0 == n1 and (a64 == a64[0]).all() and (b64 == b64[0]).all():
            r, c = linear_sum_assignment(C)
            T = np.zeros_like(C)
            T[r, c] = a64[0]
        else:
            A_eq = np.zeros((n0 + n1, n0 * n1))
            for i in range(n0):
                A_eq[i, i * n1:(i + 1) * n1] = 1
            for j in range(n1):
                A_eq[n0 + j, j::n1] = 1
            res = linprog(C.ravel(), A_eq=A_eq[:-1], b_eq=np.concatenate([a64, b64 * (a64.sum() / b64.sum())])[:-1],
                          bounds=(0, None), method='highs')
            assert res.status == 0, res.message
            T = res.x.reshape(n0, n1)
        return torch.from_numpy(T).to(M.dtype)
    mod.emd = emd
    sys.modules['ot'] = mod


def run_otf(out, tag, models, X):
    """OTFusion.fuse_models(models[:-1] -> models[-1]) as run_otf.py:94-110 sets it up, both modes, uniform importances."""
    _install_ot_stand_in()
    from fusion_algorithms import otfusion as ref_otf
    for mode in ('acts', 'wts'):
        strategy = {'mode': mode, mode: {'activations_dataset': X, 'activation_labels': None,
                                         'neuron_importance_method': 'uniform', 'normalize_activations': False}}
        log = []
        orig = ref_otf.get_ot_matrix

        def recording(cost, mu, nu):
            T = orig(cost, mu, nu)
            log.append((cost.detach().cpu().numpy().copy(), T.detach().cpu().numpy().copy()))
            return T
        ref_otf.get_ot_matrix = recording
        copies = [m.copy_model().eval() for m in models]
        try:
            fused = quiet(ref_otf.OTFusion(copies[:-1], copies[-1]).fuse_models, strategy,
                          handle_skip=(mode == 'acts'), handle_bias=True)
        finally:
            ref_otf.get_ot_matrix = orig
        K = len(models) - 1
        for idx, (C, T) in enumerate(log):
            l, k = divmod(idx, K)
            out[f'{tag}.{mode}.cost{l}_{k}'] = C.astype(np.float32)
            out[f'{tag}.{mode}.T{l}_{k}'] = T.astype(np.float32)
            dup = len(np.unique(C, axis=0)) < C.shape[0] or len(np.unique(C, axis=1)) < C.shape[1]
            out[f'{tag}.{mode}.dup{l}_{k}'] = np.array([int(dup)])
        sd_to(out, f'{tag}.{mode}.fused', fused)


def fixture_otf():
    out = {}
    # the same three model families as mlp_small / mlp_wide / vgg_tiny (same seeds, so the same weights)
    models = []
    for k in range(4):
        torch.manual_seed(100 + k)
        models.append(MLPNet(20, 32, 10, use_bias=True, use_activations=False).eval())
    torch.manual_seed(0)
    X = torch.randn(48, 1, 4, 5)
    out['mlp_small.X'] = X.numpy()
    for k, m in enumerate(models):
        sd_to(out, f'mlp_small.sd{k}', m)
    run_otf(out, 'mlp_small', models, X)

    models = []
    for k in range(3):
        torch.manual_seed(200 + k)
        models.append(MLPNet(24, 96, 10, use_bias=True, use_activations=False).eval())
    torch.manual_seed(1)
    X = torch.randn(256, 1, 4, 6)
    out['mlp_wide.X'] = X.numpy()
    for k, m in enumerate(models):
        sd_to(out, f'mlp_wide.sd{k}', m)
    run_otf(out, 'mlp_wide', models, X)

    VGG.cfg['VGG_tiny'] = [8, 'M', 16, 'M', 16, 16, 'M', 32, 32, 'M', 32, 32, 'M']
    models = []
    for k in range(3):
        torch.manual_seed(300 + k)
        models.append(VGG('VGG_tiny', 10, (3, 32, 32), batch_norm=False, use_bias=True, use_activations=False).eval())
    torch.manual_seed(2)
    X = torch.randn(6, 3, 32, 32)
    out['vgg_tiny.X'] = X.numpy()
    out['vgg_tiny.cfg'] = np.array([str(c) for c in VGG.cfg['VGG_tiny']])
    for k, m in enumerate(models):
        sd_to(out, f'vgg_tiny.sd{k}', m)
    run_otf(out, 'vgg_tiny', models, X)
    meta(out)
    np.savez_compressed(os.path.join(OUT, 'otf.npz'), **out)
    return out


FIXTURES = dict(mlp_small=fixture_mlp_small, mlp_wide=fixture_mlp_wide, vgg_tiny=fixture_vgg_tiny,
                vit_tiny=fixture_vit_tiny, mlp_real=fixture_mlp_real, alf_kd=fixture_alf_kd, otf=fixture_otf)

if __name__ == '__main__':
    for fx in [FIXTURES[n] for n in sys.argv[1:]] or list(FIXTURES.values()):
        o = fx()
        print(fx.__name__, len(o), 'arrays')
