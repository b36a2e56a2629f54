"""CPU tests of the captum-free attributions (parity with captum itself is unpinned: captum is not
installable here and the reference holds no attribution outputs), bound by their defining properties."""
import torch
import torch.nn as nn

from neuron_interpolation_model_fusion_amd.models import MLPNet, VGG
from neuron_interpolation_model_fusion_amd.utils import neuron_importance as ni
from neuron_interpolation_model_fusion_amd.utils.fusion_utils import compute_neuron_importance_scores


def _mlp():
    torch.manual_seed(0)
    return MLPNet(20, 16, 5, use_bias=True, use_activations=False).double()


def _vgg():
    torch.manual_seed(1)
    VGG.cfg.setdefault('VGG_ni_test', [6, 'M', 8, 'M'])
    return VGG('VGG_ni_test', 5, (3, 8, 8), batch_norm=False, use_bias=True, use_activations=False).double()


def _target_logit(model, X, y):
    return model(X).gather(1, y.view(-1, 1)).reshape(-1)


def test_deeplift_completeness_mlp_and_maxpool_vgg():
    """Rescale rule: the attributions of any layer sum to logit(x) - logit(baseline), exactly."""
    for model, X in ((_mlp(), torch.randn(7, 20).double()), (_vgg(), torch.randn(5, 3, 8, 8).double())):
        y = torch.arange(X.shape[0]) % 5
        delta = _target_logit(model, X, y) - _target_logit(model, torch.zeros_like(X), y)
        for rep in model.get_ordered_trainable_named_layers[:-1]:
            attr = ni.layer_deeplift(model, rep.layer, X, y)
            torch.testing.assert_close(attr.flatten(1).sum(1), delta, rtol=1e-9, atol=1e-10)
        for level in model.get_ordered_trainable_named_levels[:-1]:
            attr = ni.layer_deeplift(model, level.network, X, y)
            torch.testing.assert_close(attr.flatten(1).sum(1), delta, rtol=1e-9, atol=1e-10)


def test_deeplift_of_the_output_layer_is_the_logit_change():
    model, X = _mlp(), torch.randn(4, 20).double()
    y = torch.tensor([0, 1, 2, 3])
    last = model.get_ordered_trainable_named_layers[-1].layer
    attr = ni.layer_deeplift(model, last, X, y)
    out, ref = model(X), model(torch.zeros_like(X))
    expect = torch.zeros_like(out).scatter_(1, y.view(-1, 1), (out - ref).gather(1, y.view(-1, 1)))
    torch.testing.assert_close(attr, expect)


def test_conductance_sums_to_the_logit_change_between_the_end_nodes():
    """sum_j cond_j -> F(a_n x) - F(a_0 x) as the quadrature gets finer (first-order in the step)."""
    model, X = _mlp(), torch.randn(6, 20).double()
    y = torch.arange(6) % 5
    layer = model.get_ordered_trainable_named_layers[0].layer
    import numpy as np
    for n_steps, tol in ((10, 0.2), (200, 0.02)):
        nodes = np.polynomial.legendre.leggauss(n_steps + 1)[0]
        a0, an = 0.5 * (1 + nodes[0]), 0.5 * (1 + nodes[-1])
        delta = _target_logit(model, an * X, y) - _target_logit(model, a0 * X, y)
        cond = ni.layer_conductance(model, layer, X, y, n_steps=n_steps)
        assert cond.shape == (6, 16)
        err = (cond.sum(1) - delta).abs().max() / delta.abs().max()
        assert err < tol, (n_steps, float(err))


def test_conductance_of_a_linear_model_is_exact():
    torch.manual_seed(2)
    model = nn.Sequential(nn.Linear(9, 7), nn.Linear(7, 3)).double()
    X, y = torch.randn(5, 9).double(), torch.tensor([0, 1, 2, 0, 1])
    cond = ni.layer_conductance(model, model[0], X, y, n_steps=10)
    import numpy as np
    nodes = np.polynomial.legendre.leggauss(11)[0]
    span = 0.5 * (nodes[-1] - nodes[0])
    h = X @ model[0].weight.T  # bias cancels in the differences
    expect = span * h * model[1].weight[y]
    torch.testing.assert_close(cond, expect)


def test_feature_ablation_matches_brute_force():
    model, X = _mlp(), torch.randn(3, 20).double()
    y = torch.tensor([4, 0, 2])
    layer = model.get_ordered_trainable_named_layers[0].layer
    attr = ni.layer_feature_ablation(model, layer, X, y)
    h = model.level1(X)
    base = model.level2(h).gather(1, y.view(-1, 1)).reshape(-1)
    for u in range(16):
        hu = h.clone()
        hu[:, u] = 0
        torch.testing.assert_close(attr[:, u], base - model.level2(hu).gather(1, y.view(-1, 1)).reshape(-1))


def test_scores_shape_reduction_and_dispatch():
    model, X = _vgg().float(), torch.randn(12, 3, 8, 8)
    y = torch.arange(12) % 5
    for method in ('conductance', 'deeplift'):
        for logic, n in (('layer', len(model.get_ordered_trainable_named_layers)),
                         ('level', len(model.get_ordered_trainable_named_levels))):
            for idx in range(n):
                s = compute_neuron_importance_scores(method, model, idx, X, y, logic, batch_size=5)
                width = (model.get_ordered_trainable_named_layers[idx].weight.shape[0] if logic == 'layer'
                         else model.get_ordered_trainable_named_levels[idx].output_width)
                assert s.shape == (width,) and s.device.type == 'cpu' and bool((s >= ni.EPSILON * 0.99).all())
    # batching does not change the result: the score is a sum over samples
    a = compute_neuron_importance_scores('deeplift', model, 1, X, y, 'layer', batch_size=5)
    b = compute_neuron_importance_scores('deeplift', model, 1, X, y, 'layer', batch_size=12)
    torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-7)
    assert torch.equal(compute_neuron_importance_scores('uniform', model, 0, X, y, 'layer'), torch.ones(6))
