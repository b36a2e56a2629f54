"""Oracle for Git Re-Basin activation matching (reference fusion_algorithms/git_rebasin.py).
Test infrastructure: see oracle/__init__.py."""
import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

from neuron_interpolation_model_fusion_amd.utils.fusion_utils import (
    get_num_layers, initialize_activations, get_layer_weight, get_layer_width)


def gr_cost(acts0, acts1, s0, s1, chunk=64):
    """git_rebasin.py:60-78, evaluated in row chunks of the (m, n0, n1) broadcast so it fits memory:
    avg = (a0 s0 + a1 s1)/(s0 + s1); cost = s0 ||avg - a0|| + s1 ||avg - a1||."""
    n0, n1 = acts0.shape[1], acts1.shape[1]
    out = torch.zeros(n0, n1)
    se = s1.unsqueeze(0)
    for i0 in range(0, n0, chunk):
        i1 = min(i0 + chunk, n0)
        a0 = acts0[:, i0:i1].unsqueeze(2)
        a1 = acts1.unsqueeze(1)
        sb = s0[i0:i1].unsqueeze(1)
        avg = (a0 * sb + a1 * se) / (sb + se)
        out[i0:i1] = sb * torch.norm(avg - a0, dim=0) + se * torch.norm(avg - a1, dim=0)
    return out


def relu_rows(pre):
    """Post-ReLU activations as (rows, width); conv rows are (b, h, w). git_rebasin.py:97-101."""
    act = pre * (pre > 0)
    if act.dim() == 4:
        act = act.permute(0, 2, 3, 1).reshape(act.shape[0] * act.shape[2] * act.shape[3], -1)
    return act


def gr_fuse(model0, model1, X, scores0=None, scores1=None, log=None):
    """GitRebasin.fuse with handle_skip=False, git_rebasin.py:125-231."""
    n_layers = get_num_layers([model0, model1])
    if scores0 is None:
        scores0 = [torch.ones(get_layer_width(model0, l)) for l in range(n_layers)]
        scores1 = [torch.ones(get_layer_width(model1, l)) for l in range(n_layers)]
    initialize_activations(model0, X)
    initialize_activations(model1, X)
    fused = model0.copy_model().eval()
    prev_conv, prev_P = False, None
    for l in range(n_layers):
        w0, is_conv = get_layer_weight(model0, l)
        w1, _ = get_layer_weight(model1, l)
        if is_conv:
            w0 = w0.flatten(1)
        b0 = model0.get_ordered_trainable_named_layers[l].layer.bias
        b1 = model1.get_ordered_trainable_named_layers[l].layer.bias
        s0, s1 = scores0[l], scores1[l]
        a0 = relu_rows(model0.get_inputs_and_preactivations(l)[1])
        a1 = relu_rows(model1.get_inputs_and_preactivations(l)[1])
        cost = gr_cost(a0, a1, s0, s1)
        rows, cols = linear_sum_assignment(cost)
        if log is not None:
            log.append(dict(cost=cost.numpy(), col=np.asarray(cols)))  # the solver's answer, before the override
        if l == n_layers - 1:
            cols = rows  # git_rebasin.py:115-116
        P = torch.zeros_like(cost)
        P[rows, cols] = 1
        if l == 0:
            aligned = P @ (w1.flatten(1) if is_conv else w1)
        elif is_conv:
            # permute the input channels by the previous layer's matching, every kernel position alike
            t = torch.cat(w1.permute(0, 2, 1).unbind()) @ prev_P.T
            t = torch.stack(t.split(w1.shape[2]), dim=0).permute(0, 2, 1)
            aligned = P @ t.flatten(1)
        elif prev_conv:
            r = w1.reshape(w1.shape[0], prev_P.shape[0], -1)
            t = torch.cat(r.permute(0, 2, 1).unbind()) @ prev_P.T
            t = torch.stack(t.split(r.shape[2]), dim=0).permute(0, 2, 1)
            aligned = P @ t.flatten(1)
        else:
            aligned = P @ w1 @ prev_P.T
        s1p = P @ s1
        a, b = s0 / (s0 + s1p), s1p / (s0 + s1p)
        fused.set_weight_at_layer(l, w0 * a.unsqueeze(1) + aligned * b.unsqueeze(1))
        if b0 is not None:
            fused.set_bias_at_layer(l, b0 * a + (P @ b1) * b)
        prev_conv, prev_P = is_conv, P
    model0.remove_hooks()
    model1.remove_hooks()
    fused.remove_hooks()
    return fused.copy_model()
