"""Neuron-importance scores without captum (reference: utils/neuron_importance.py:11-103).

The reference wraps three captum 0.7 layer attributions (``requirements.txt:9``) and reduces them to
one score per neuron: ``sum_over_samples_and_positions |attribution| / N + 1e-8``.  captum is not
installable here, so the attributions are restated from its published algorithms in plain autograd:

* conductance (``LayerConductance(...).attribute(X, target=y, n_steps=10)``): inputs scaled from
  the zero baseline at the n_steps + 1 Gauss-Legendre nodes of [0, 1]; per step, the gradient of the
  target logit w.r.t. the layer output times the change of the layer output to the next node;
* DeepLIFT (``LayerDeepLift``, rescale rule, zero baseline, multiply-by-inputs): backpropagation of
  the target logit in which every ``nn.ReLU`` (and the other pointwise non-linear modules captum
  supports) passes ``grad_out * delta_out / delta_in`` and ``nn.MaxPool2d`` its cross-max rule,
  deltas being taken against a forward pass of the baseline; attribution = multiplier * (layer
  output on the input - layer output on the baseline);
* feature ablation (``LayerFeatureAblation``): every element of the layer output is zeroed in turn;
  attribution = change of the target logit.

PARITY UNPINNED: no captum output exists in the reference tree or in this container to check these
against; the tests bind them by their defining properties instead (completeness of DeepLIFT,
convergence of conductance to the logit difference, brute-force ablation).
These are model forward/backward passes (PyTorch); nothing here is a kernel of the matching path.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

EPSILON = 1e-8
_DEEPLIFT_EPS = 1e-10
_POINTWISE = (nn.ReLU, nn.ELU, nn.LeakyReLU, nn.Sigmoid, nn.Tanh, nn.Softplus)


def _capture(layer):
    """Forward hook keeping the layer's output tensor (with its graph)."""
    box = {}

    def hook(_module, _inputs, output):
        box['out'] = output
    return box, layer.register_forward_hook(hook)


def _target_sum(logits, target):
    return logits.gather(1, target.reshape(-1, 1).to(logits.device)).sum()


def _capture_all(layers):
    boxes, handles = [], []
    for layer in layers:
        box, handle = _capture(layer)
        boxes.append(box)
        handles.append(handle)
    return boxes, handles


def _grads_of_outputs(target, boxes):
    """d target / d (captured output) of every hooked layer from one backward pass.  With several layers in one pass a
    layer the target does not depend on gets zeros (asked for alone, autograd refuses it, as it would in the reference)."""
    outs = [box['out'] for box in boxes]
    grads = torch.autograd.grad(target, outs, allow_unused=len(outs) > 1)
    return [torch.zeros_like(o) if g is None else g for o, g in zip(outs, grads)]


def layers_conductance(model, layers, X, y, n_steps=10):
    """layer_conductance of several layers from ONE forward and ONE backward pass: the gradient of the target logit
    with respect to a layer's output is the same tensor whichever other outputs the backward pass is asked for, so
    the list equals [layer_conductance(model, l, X, y) for l in layers] (the reference runs one pass per layer)."""
    nodes = np.polynomial.legendre.leggauss(n_steps + 1)[0]
    alphas = [float(0.5 * (1.0 + t)) for t in nodes]
    B = X.shape[0]
    scaled = torch.cat([a * X for a in alphas], dim=0).requires_grad_()
    boxes, handles = _capture_all(layers)
    try:
        with torch.enable_grad():
            logits = model(scaled)
            grads = _grads_of_outputs(_target_sum(logits, y.repeat(n_steps + 1)), boxes)
    finally:
        for handle in handles:
            handle.remove()
    out = []
    for box, g in zip(boxes, grads):
        evals = box['out'].detach()
        step = (evals[B:] - evals[:-B]) * g[:-B]
        out.append(step.reshape(n_steps, B, *evals.shape[1:]).sum(dim=0))
    return out


def layer_conductance(model, layer, X, y, n_steps=10):
    """[B, *layer_output_shape] conductance of every layer unit for the target logit."""
    return layers_conductance(model, [layer], X, y, n_steps)[0]


class _RescaleFn(torch.autograd.Function):
    """One non-linear module under DeepLIFT's gradient override, on a batch laid out as
    [inputs ; baselines]: forward is the module itself; backward returns, instead of the derivative,
    ``grad_out * delta_out / delta_in`` (pointwise modules) or the cross-max rule (max pooling), and
    falls back to the ordinary gradient where |delta_in| < 1e-10."""

    @staticmethod
    def forward(ctx, x, module, plain_forward):
        with torch.enable_grad():
            inner = x.detach().requires_grad_()
            out = plain_forward(inner)
        ctx.module, ctx.inner, ctx.out = module, inner, out
        return out.detach()

    @staticmethod
    def backward(ctx, grad_out):
        module, x, out_graph = ctx.module, ctx.inner.detach(), ctx.out
        plain, = torch.autograd.grad(out_graph, ctx.inner, grad_out)
        out = out_graph.detach()
        a, a_ref = x.chunk(2)
        o, o_ref = out.chunk(2)
        delta_in = torch.cat(2 * [a - a_ref])
        if isinstance(module, nn.MaxPool2d):
            cross = torch.max(o, o_ref)
            delta_out = torch.cat([cross - o_ref, o - cross])
            _, idx = F.max_pool2d(x, module.kernel_size, module.stride, module.padding, module.dilation,
                                  module.ceil_mode, True)
            spread = F.max_unpool2d(grad_out * delta_out, idx, module.kernel_size, module.stride, module.padding,
                                    list(x.shape))
            s, s_ref = spread.chunk(2)
            new = torch.cat(2 * [s + s_ref]) / delta_in
        else:
            new = grad_out * torch.cat(2 * [o - o_ref]) / delta_in
        return torch.where(delta_in.abs() < _DEEPLIFT_EPS, plain, new), None, None


class _RescaleRule:
    """Routes a module's forward through _RescaleFn for the duration of one attribution."""

    def __init__(self, module):
        self.module = module
        plain_forward = module.forward
        module.forward = lambda x: _RescaleFn.apply(x, module, plain_forward)

    def remove(self):
        del self.module.forward  # the instance attribute shadowed the class method


def layers_deeplift(model, layers, X, y):
    """layer_deeplift of several layers from ONE forward and ONE backward pass (see layers_conductance)."""
    rules = [_RescaleRule(m) for m in model.modules() if isinstance(m, _POINTWISE + (nn.MaxPool2d,))]
    boxes, handles = _capture_all(layers)
    try:
        with torch.enable_grad():
            both = torch.cat([X, torch.zeros_like(X)], dim=0).requires_grad_()
            logits = model(both)
            grads = _grads_of_outputs(_target_sum(logits, y.repeat(2)), boxes)
    finally:
        for handle in handles:
            handle.remove()
        for r in rules:
            r.remove()
    out = []
    for box, g in zip(boxes, grads):
        o, o_ref = box['out'].detach().chunk(2)
        out.append(g.chunk(2)[0] * (o - o_ref))
    return out


def layer_deeplift(model, layer, X, y):
    """[B, *layer_output_shape] DeepLIFT (rescale) attribution of every layer unit, zero baseline."""
    return layers_deeplift(model, [layer], X, y)[0]


def layer_feature_ablation(model, layer, X, y):
    """[B, *layer_output_shape]: target logit minus the target logit with one layer unit set to zero."""
    state = {'flat_index': None}

    def hook(_module, _inputs, output):
        if state['flat_index'] is None:
            state['shape'] = output.shape
            return None
        ablated = output.clone()
        ablated.view(output.shape[0], -1)[:, state['flat_index']] = 0
        return ablated
    handle = layer.register_forward_hook(hook)
    try:
        with torch.no_grad():
            pick = y.reshape(-1, 1).to(X.device)
            base = model(X).gather(1, pick).reshape(-1)
            shape = state['shape']
            units = int(np.prod(shape[1:]))
            attr = torch.empty(shape[0], units, device=base.device, dtype=base.dtype)
            for u in range(units):
                state['flat_index'] = u
                attr[:, u] = base - model(X).gather(1, pick).reshape(-1)
    finally:
        handle.remove()
    return attr.reshape(shape)


_METHODS = {
    'conductance': layer_conductance,
    'deeplift': layer_deeplift,
    'featureablation': layer_feature_ablation,
}


_MULTI = {'conductance': layers_conductance, 'deeplift': layers_deeplift}


def get_scores(method, model, layers, output_widths, X, y, batch_size=None):
    """get_score for several layers of one model at once: conductance and DeepLIFT take every layer's attribution
    from the same forward / backward pass per batch (a fusion asks for all layers of every model,
    hf_kf.py:527-536: L passes per batch become one); feature ablation has no shared pass and runs layer by layer.
    Per layer the arithmetic is get_score's: the same sums in the same order, so the same numbers."""
    if method not in _MULTI:
        return [get_score(method, model, layer, w, X, y, batch_size) for layer, w in zip(layers, output_widths)]
    N = X.shape[0]
    batch_size = batch_size or N
    scores = [torch.zeros(w) for w in output_widths]
    for lo in range(0, N, batch_size):
        attrs = _MULTI[method](model, layers, X[lo:lo + batch_size], y[lo:lo + batch_size])
        sums = [a.detach().abs().sum(0) for a in attrs]
        for sc, a in zip(scores, [t.cpu() for t in sums]):
            sc += a.sum(dim=tuple(range(1, a.ndim))) if a.ndim > 1 else a
    device = next(model.parameters()).device
    return [(sc / N).to(device) + EPSILON for sc in scores]


def get_score(method, model, layer, output_width, X, y, batch_size=None):
    """One score per neuron: mean over samples of |attribution| summed over positions, + 1e-8
    (neuron_importance.py:43-60 ``_accumulate_scores`` / ``_normalize_scores``), on the model's device."""
    attribute = _METHODS[method]
    N = X.shape[0]
    batch_size = batch_size or N
    scores = torch.zeros(output_width)
    for lo in range(0, N, batch_size):
        a = attribute(model, layer, X[lo:lo + batch_size], y[lo:lo + batch_size]).detach().abs().sum(0).cpu()
        scores += a.sum(dim=tuple(range(1, a.ndim))) if a.ndim > 1 else a
    device = next(model.parameters()).device
    return (scores / N).to(device) + EPSILON
