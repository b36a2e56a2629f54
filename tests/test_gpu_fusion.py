"""-m gpu: the host classes (HIP path through the C ABI) against the golden vectors and the oracle."""
import numpy as np
import pytest
import torch

from tests.helpers import load_npz, build_models, fused_state, assert_state_close

pytestmark = pytest.mark.gpu

FIXTURES = ['mlp_small', 'mlp_wide', 'vgg_tiny', 'mlp_real']


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'these tests need the GPU box'
    return torch.device('cuda', 0)


@pytest.mark.parametrize('name', FIXTURES)
def test_git_rebasin_matches_reference_golden(dev, name):
    from neuron_interpolation_model_fusion_amd.fusion_algorithms import GitRebasin
    z = load_npz(name)
    models, X = build_models(name, z, dev)
    gr = GitRebasin(models[0], models[1])
    gr.cost_matrices = []
    fused = gr.fuse(X, None, 'uniform')
    n_layers = len(gr.cost_matrices)
    for l in range(n_layers):
        cost = gr.cost_matrices[l].cpu().numpy()
        ref = z[f'gr.cost{l}']
        np.testing.assert_allclose(cost, ref, rtol=1e-4, atol=3e-5 * ref.max())  # Gram-form cancellation on near-identical columns
        if l < n_layers - 1:  # the last layer's matching is forced to the identity
            assert (gr.permutations[l].cpu().numpy() == z[f'gr.col{l}']).all(), f'layer {l}'
    assert_state_close(fused, fused_state(z, 'gr'), rtol=1e-3)


@pytest.mark.parametrize('name', ['mlp_small', 'mlp_wide', 'vgg_tiny'])
@pytest.mark.parametrize('mode', ['acts', 'wts'])
def test_otfusion_matches_oracle(dev, name, mode):
    """OTFusion parity is unpinned at ot.emd (POT absent): the checker is the oracle restatement."""
    from neuron_interpolation_model_fusion_amd.fusion_algorithms import OTFusion
    from oracle import otfusion as o_otf
    z = load_npz(name)
    models, X = build_models(name, z, dev)
    cpu_models, Xc = build_models(name, z, 'cpu')
    strategy = {'mode': mode,
                'acts': {'activations_dataset': X, 'activation_labels': None, 'neuron_importance_method': 'uniform',
                         'normalize_activations': False},
                'wts': {'neuron_importance_method': 'uniform'}}
    fused = OTFusion(models[:-1], models[-1]).fuse_models(strategy, handle_skip=(mode == 'acts'))
    ref = o_otf.otf_fuse(cpu_models[:-1], cpu_models[-1], Xc, mode=mode)
    assert_state_close(fused, dict(ref.state_dict()), rtol=1e-3)


def test_otfusion_normalized_acts(dev):
    from neuron_interpolation_model_fusion_amd.fusion_algorithms import OTFusion
    from oracle import otfusion as o_otf
    z = load_npz('mlp_wide')
    models, X = build_models('mlp_wide', z, dev)
    cpu_models, Xc = build_models('mlp_wide', z, 'cpu')
    strategy = {'mode': 'acts',
                'acts': {'activations_dataset': X, 'activation_labels': None, 'neuron_importance_method': 'uniform',
                         'normalize_activations': True}}
    fused = OTFusion(models[:1], models[1]).fuse_models(strategy)
    ref = o_otf.otf_fuse(cpu_models[:1], cpu_models[1], Xc, mode='acts', normalize=True)
    assert_state_close(fused, dict(ref.state_dict()), rtol=1e-3)


@pytest.mark.parametrize('name', FIXTURES)
def test_hf_matches_reference_golden(dev, name):
    from neuron_interpolation_model_fusion_amd.fusion_algorithms import LSFusion
    z = load_npz(name)
    models, X = build_models(name, z, dev)
    ls = LSFusion(models[:2])
    fused = ls.fuse(X, None, fusion_method='hungarian', neuron_importance_method='uniform')
    n_layers = len(ls.last_layer_info)
    for l, info in enumerate(ls.last_layer_info):
        ref = z[f'hf.cost{l}']
        np.testing.assert_allclose(info['cost'].cpu().numpy(), ref, rtol=2e-3, atol=2e-4 * ref.max())
        if f'hf.w0_{l}' in z.files:
            wref = np.concatenate([z[f'hf.w0_{l}'], z[f'hf.w1_{l}']], axis=1)
            w = info['W'].cpu().numpy()
            assert np.abs(w - wref).max() <= 2e-3 * np.abs(wref).max(), f'LS solution, layer {l}'
        assert (info['col'].cpu().numpy() == z[f'hf.col{l}']).all(), f'layer {l}'
    assert_state_close(fused, fused_state(z, 'hf'), rtol=1e-3)


def test_hf_norm_acts_matches_reference_golden(dev):
    from neuron_interpolation_model_fusion_amd.fusion_algorithms import LSFusion
    z = load_npz('mlp_small')
    models, X = build_models('mlp_small', z, dev)
    ls = LSFusion(models[:2])
    fused = ls.fuse(X, None, fusion_method='hungarian', neuron_importance_method='uniform', norm_acts=True)
    for l, info in enumerate(ls.last_layer_info):
        assert (info['col'].cpu().numpy() == z[f'hfn.col{l}']).all(), f'layer {l}'
    assert_state_close(fused, fused_state(z, 'hfn'), rtol=1e-3)


@pytest.mark.parametrize('name,count', [('mlp_small', 2), ('mlp_small', 4), ('mlp_wide', 3), ('vgg_tiny', 3)])
def test_kf_matches_reference_golden(dev, name, count):
    from neuron_interpolation_model_fusion_amd.fusion_algorithms import LSFusion
    z = load_npz(name)
    models, X = build_models(name, z, dev)
    np.random.seed(int(z['meta.seed'][0]))
    ls = LSFusion(models[:count])
    fused = ls.fuse(X, None, fusion_method='k_means', neuron_importance_method='uniform')
    for l, info in enumerate(ls.last_layer_info):
        assert (info['assign'].cpu().numpy() == z[f'kf{count}.assign{l}']).all(), f'layer {l}'
    assert_state_close(fused, fused_state(z, f'kf{count}'), rtol=1e-3)
