"""Oracle for OTFusion (reference fusion_algorithms/otfusion.py).

PARITY UNPINNED at ot.emd (otfusion.py:32): POT (pot~=0.9.5, requirements.txt:7) is not
installed and the reference module cannot be imported.  `emd` solves the same transport LP
exactly with scipy.  Test infrastructure: see oracle/__init__.py."""
import numpy as np
import torch
from scipy.optimize import linear_sum_assignment, linprog

from neuron_interpolation_model_fusion_amd.utils.fusion_utils import (
    get_num_layers, initialize_activations, normalize_activations, get_layer_weight, get_layer_width,
    get_preactivations_of_layer)
from neuron_interpolation_model_fusion_amd.utils.model_utils import is_conv_weight


def emd(mu, nu, cost):
    """Exact optimal transport plan (what ot.emd returns, otfusion.py:18-32).
    Uniform marginals on a square problem: a vertex of the Birkhoff polytope, P/n, P from
    linear_sum_assignment.  Otherwise the transport LP with HiGHS."""
    mu = np.asarray(mu, dtype=np.float64)
    nu = np.asarray(nu, dtype=np.float64)
    C = np.asarray(cost, dtype=np.float64)
    n0, n1 = C.shape
    if n0 == n1 and np.allclose(mu, mu[0]) and np.allclose(nu, nu[0]):
        r, c = linear_sum_assignment(C)
        T = np.zeros_like(C)
        T[r, c] = mu[0]
        return torch.from_numpy(T).float()
    A_eq = np.zeros((n0 + n1, n0 * n1))
    for i in range(n0):
        A_eq[i, i * n1:(i + 1) * n1] = 1
    for j in range(n1):
        A_eq[n0 + j, j::n1] = 1
    res = linprog(C.ravel(), A_eq=A_eq[:-1], b_eq=np.concatenate([mu, nu])[:-1], bounds=(0, None), method='highs')
    assert res.status == 0, res.message
    return torch.from_numpy(res.x.reshape(n0, n1)).float()


def sinkhorn_knopp(mu, nu, cost, reg, num_iter_max=1000, stop_thr=1e-9):
    """Entropic plan by Sinkhorn-Knopp, restating the published algorithm of POT's ``ot.sinkhorn(a, b, M, reg)``
    (method 'sinkhorn', POT 0.9.x bregman `sinkhorn_knopp`) in f64.  PARITY UNPINNED: POT is not installed and the
    reference never calls it (it uses ot.emd); this is the checker of the library's additional Sinkhorn mode."""
    a = np.asarray(mu, dtype=np.float64)
    b = np.asarray(nu, dtype=np.float64)
    M = np.asarray(cost, dtype=np.float64)
    u = np.ones(a.shape[0]) / a.shape[0]
    v = np.ones(b.shape[0]) / b.shape[0]
    K = np.exp(-M / reg)
    Kp = (1.0 / a).reshape(-1, 1) * K
    iters = 0
    for ii in range(num_iter_max):
        uprev, vprev = u, v
        KtU = K.T @ u
        with np.errstate(divide='ignore', invalid='ignore'):
            v = b / KtU
            u = 1.0 / (Kp @ v)
        if np.any(KtU == 0) or not np.all(np.isfinite(u)) or not np.all(np.isfinite(v)):
            u, v = uprev, vprev
            break
        iters = ii + 1
        if ii % 10 == 0:
            tmp2 = np.einsum('i,ij,j->j', u, K, v)
            if np.linalg.norm(tmp2 - b) < stop_thr:
                break
    return torch.from_numpy(u.reshape(-1, 1) * K * v.reshape(1, -1)).float(), iters


def ground_metric_acts(act1, act2, normalize=False):
    """Euclidean distance between neuron activation vectors, otfusion.py:135-161."""
    if normalize:
        act1, act2 = normalize_activations(act1), normalize_activations(act2)
    return torch.cdist(act1.T.cpu(), act2.T.cpu(), p=2)


def ground_metric_wts(aligned_weight, fused_weight):
    """otfusion.py:163-182 (flatten(1) is a no-op on 2-D weights)."""
    return torch.cdist(aligned_weight.flatten(1), fused_weight.flatten(1), p=2)


def otf_fuse(models, target_model, X=None, mode='acts', normalize=False, handle_bias=True, log=None, scores=None,
             target_scores=None, solver=None):
    """OTFusion.fuse_models with handle_skip=False, otfusion.py:193-361.  scores[k][l] / target_scores[l]:
    neuron importances already rescaled as otfusion.py:110-112 asks (None = the uniform marginals of
    :242,258); they are normalised to sum 1 here as :239-240,255-256 do.  solver(mu, nu, C) -> T replaces the exact
    plan (the library's additional Sinkhorn mode)."""
    for m in models:
        m.eval()
    target_model.eval()
    n_layers = get_num_layers(models + [target_model])
    first = target_model.get_ordered_trainable_named_layers[0].layer
    input_dim = first.in_channels if is_conv_weight(first.weight.data) else first.in_features
    fused = target_model.copy_model().eval()
    if mode == 'acts':
        for m in models:
            initialize_activations(m, X)
        initialize_activations(fused, X)  # captured once, before any layer is fused (otfusion.py:223-224)
    nu_prev = torch.ones(input_dim) / input_dim
    T_prev = [torch.diag(nu_prev) for _ in models]
    prev_conv = False
    for l in range(n_layers):
        total_w, is_conv = get_layer_weight(fused, l)
        layer = fused.get_ordered_trainable_named_layers[l].layer
        total_b = layer.bias.data.clone().detach() if handle_bias else None
        width_f = get_layer_width(fused, l)
        if target_scores is not None:
            nu = target_scores[l].clone().float()
            nu = nu / nu.sum()
        else:
            nu = torch.ones(width_f) / width_f
        for k, m in enumerate(models):
            wk, _ = get_layer_weight(m, l)
            width_k = get_layer_width(m, l)
            bk = m.get_ordered_trainable_named_layers[l].layer.bias if handle_bias else None
            if scores is not None:
                mu = scores[k][l].clone().float()
                mu = mu / mu.sum()
            else:
                mu = torch.ones(width_k) / width_k
            Tp = T_prev[k] @ torch.diag(1 / nu_prev)
            if is_conv:
                edges = (wk.permute(2, 0, 1) @ Tp.unsqueeze(0).repeat(wk.shape[2], 1, 1)).permute(1, 2, 0)
            elif prev_conv:
                r = wk.view(width_k, Tp.shape[0], -1).permute(2, 0, 1)
                edges = (r @ Tp.unsqueeze(0).repeat(r.shape[0], 1, 1)).permute(1, 2, 0).flatten(1)
            else:
                edges = wk @ Tp
            if mode == 'acts':
                a1, a2 = get_preactivations_of_layer(m, fused, l)
                C = ground_metric_acts(a1, a2, normalize)
            else:
                C = ground_metric_wts(edges, fused.get_ordered_trainable_named_layers[l].weight)
            T = emd(mu, nu, C.detach()) if solver is None else solver(mu, nu, C.detach())
            T_prev[k] = T
            Tc = T.T @ torch.diag(1 / mu)
            if l == n_layers - 1:
                Tc = torch.eye(width_k)
            if log is not None:
                log.append(dict(layer=l, model=k, cost=C.detach().numpy(), T=T.numpy()))
            aligned = Tc @ (edges.flatten(1) if is_conv else edges)
            total_w += aligned.view_as(total_w)
            if handle_bias:
                total_b += (Tc @ bk).view_as(total_b).detach()
        fused.set_weight_at_layer(l, total_w / (len(models) + 1))
        if handle_bias:
            fused.set_bias_at_layer(l, total_b / (len(models) + 1))
        nu_prev, prev_conv = nu, is_conv
    if mode == 'acts':
        for m in models:
            m.remove_hooks()
        target_model.remove_hooks()
        fused.remove_hooks()
        fused = fused.copy_model()
    return fused
