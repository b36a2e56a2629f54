"""not-gpu: pins the CPU oracle against the golden vectors minted from the imported reference."""
import numpy as np
import pytest
import torch

from oracle import alf as o_alf
from oracle import git_rebasin as o_gr
from oracle import hf_kf as o_hf
from oracle import otfusion as o_otf
from tests.helpers import load_npz, build_models, fused_state, assert_state_close, n_layers, SubNpz

FIXTURES = ['mlp_small', 'mlp_wide', 'vgg_tiny', 'mlp_real']


@pytest.mark.parametrize('name', FIXTURES)
def test_gr_oracle(name):
    z = load_npz(name)
    models, X = build_models(name, z)
    log = []
    fused = o_gr.gr_fuse(models[0], models[1], X, log=log)
    for l, rec in enumerate(log):
        np.testing.assert_allclose(rec['cost'], z[f'gr.cost{l}'], rtol=2e-5, atol=1e-6)
        assert (rec['col'] == z[f'gr.col{l}']).all()
    assert_state_close(fused, fused_state(z, 'gr'), rtol=1e-5)


@pytest.mark.parametrize('name', FIXTURES)
def test_hf_oracle(name):
    z = load_npz(name)
    models, X = build_models(name, z)
    log = []
    fused = o_hf.ls_fuse(models[:2], X, 'hungarian', log=log)
    for l, rec in enumerate(log):
        np.testing.assert_allclose(rec['cost'], z[f'hf.cost{l}'], rtol=1e-3, atol=1e-6)
        assert (rec['col'] == z[f'hf.col{l}']).all()
    assert_state_close(fused, fused_state(z, 'hf'), rtol=1e-3)


def test_hf_oracle_norm_acts():
    z = load_npz('mlp_small')
    models, X = build_models('mlp_small', z)
    log = []
    fused = o_hf.ls_fuse(models[:2], X, 'hungarian', norm_acts=True, log=log)
    for l, rec in enumerate(log):
        assert (rec['col'] == z[f'hfn.col{l}']).all()
    assert_state_close(fused, fused_state(z, 'hfn'), rtol=1e-3)


@pytest.mark.parametrize('name,count', [('mlp_small', 2), ('mlp_small', 4), ('mlp_wide', 3), ('vgg_tiny', 3)])
def test_kf_oracle(name, count):
    z = load_npz(name)
    models, X = build_models(name, z)
    np.random.seed(int(z['meta.seed'][0]))
    log = []
    fused = o_hf.ls_fuse(models[:count], X, 'k_means', log=log)
    for l, rec in enumerate(log):
        assert (rec['assign'] == z[f'kf{count}.assign{l}']).all(), f'layer {l}'
    assert_state_close(fused, fused_state(z, f'kf{count}'), rtol=1e-3)


@pytest.mark.parametrize('name,level', [('mlp_small', 0), ('mlp_wide', 0), ('vgg_tiny', 2)])
def test_alf_kmeans_oracle(name, level):
    z = load_npz(name)
    models, X = build_models(name, z)
    models = models[:2]
    k = models[0].get_ordered_trainable_named_levels[level].output_width
    with torch.no_grad():
        acts = [m.forward_until_level(X, level) for m in models]
    w = torch.ones(2 * k)
    for norm in (False, True):
        for init2 in (False, True):
            T, assign = o_alf.kmeans_v2(acts, k, w, norm, init2)
            tag = f'{int(norm)}{int(init2)}'
            assert (assign.numpy() == z[f'alf.assign.{tag}']).all(), tag
            np.testing.assert_allclose(T.numpy(), z[f'alf.T.{tag}'], rtol=1e-5, atol=1e-6)


def test_otf_oracle_properties():
    """OTFusion parity is unpinned (no POT); check the restatement's invariants instead."""
    z = load_npz('mlp_small')
    models, X = build_models('mlp_small', z)
    # identical models => identity plans => fused == input
    twin = models[0].copy_model()
    fused = o_otf.otf_fuse([twin], models[0], X, mode='acts')
    assert_state_close(fused, {k: v for k, v in models[0].state_dict().items()}, rtol=1e-6)
    # a hidden-unit permutation of the same network is undone by OT (acts and wts modes)
    perm = torch.randperm(32, generator=torch.Generator().manual_seed(0))
    shuffled = models[0].copy_model()
    sd = shuffled.state_dict()
    sd['level1.1.weight'] = sd['level1.1.weight'][perm]
    sd['level1.1.bias'] = sd['level1.1.bias'][perm]
    sd['level2.1.weight'] = sd['level2.1.weight'][:, perm]
    shuffled.load_state_dict(sd)
    for mode in ('acts', 'wts'):
        fused = o_otf.otf_fuse([shuffled], models[0], X, mode=mode)
        assert_state_close(fused, dict(models[0].state_dict()), rtol=1e-5)
    # LP path agrees with the assignment path on a uniform square problem
    C = torch.rand(6, 6, generator=torch.Generator().manual_seed(1))
    mu = np.full(6, 1 / 6)
    T1 = o_otf.emd(mu, mu, C)
    mu2 = mu.copy()
    mu2[0] += 1e-3  # breaks the "uniform" shortcut
    mu2 /= mu2.sum()
    T2 = o_otf.emd(mu2, mu, C)
    assert abs(float((T1 * C).sum()) - float((T2 * C).sum())) < 1e-2
    assert torch.allclose(T2.sum(1), torch.from_numpy(mu2).float(), atol=1e-6)


def test_alf_kmeans_oracle_vit_with_skip_alignment():
    """Transformer level [B, hidden, seq] with the target/input re-alignment (alf.py:290-294)."""
    z = load_npz('vit_tiny')
    models, X = build_models('vit_tiny', z)
    level = 1
    k = models[0].get_ordered_trainable_named_levels[level].output_width
    with torch.no_grad():
        acts = [m.forward_until_level(X, level) for m in models]
        level_X = models[0].forward_until_level(X, level - 1)
    w = torch.ones(2 * k)
    for norm in (False, True):
        for init2 in (False, True):
            T, assign = o_alf.kmeans_v2(acts, k, w, norm, init2, level_X=level_X, align_skip=True)
            tag = f'{int(norm)}{int(init2)}'
            assert (assign.numpy() == z[f'alf.assign.{tag}']).all(), tag
            np.testing.assert_allclose(T.numpy(), z[f'alf.T.{tag}'], rtol=1e-5, atol=1e-6)


def test_alf_last_level_grouping_oracle():
    """alf.py:398-412 with / without cls_head_weights and use_kd_on_head (goldens: tests/golden/alf_kd.npz)."""
    z = load_npz('alf_kd')
    models, X = build_models('mlp_kd', z)
    head_w = torch.from_numpy(z['head_w'])
    for hw in (0, 1):
        for kd in (0, 1):
            T, assign = o_alf.last_level_grouping(models, X, 10, head_w if hw else None, bool(kd), 2.0)
            assert (assign.numpy() == z[f'grp.assign.{hw}{kd}']).all()
            ref = z[f'grp.T.{hw}{kd}']
            assert T.numpy().dtype == ref.dtype  # float64 head weights promote the targets, as in the reference
            np.testing.assert_allclose(T.numpy(), ref, rtol=1e-6, atol=1e-7)


def test_alf_head_criterion_oracle_matches_torch_modules():
    g = torch.Generator().manual_seed(0)
    p = torch.randn(8, 10, generator=g)
    t = torch.softmax(torch.randn(8, 10, generator=g), dim=1)
    w = torch.rand(10, generator=g)
    hw = torch.ones(2, 10)
    assert torch.allclose(o_alf.head_criterion(False, hw, True, w)(p, t), torch.nn.MSELoss()(p, t))
    assert torch.allclose(o_alf.head_criterion(True, None, True, w)(p, t), torch.nn.MSELoss()(p, t))
    assert torch.allclose(o_alf.head_criterion(True, hw, True, w)(p, t), torch.nn.KLDivLoss(reduction='batchmean')(p, t))
    assert torch.allclose(o_alf.head_criterion(True, hw, False, w)(p, t), ((p - t) ** 2 * w).mean())


@pytest.mark.parametrize('name', ['mlp_small', 'mlp_wide', 'vgg_tiny'])
@pytest.mark.parametrize('mode', ['acts', 'wts'])
def test_otf_oracle_against_the_reference_run_with_a_stand_in_exact_solver(name, mode):
    """otfusion.py:193-361 minus ot.emd: the reference module itself was run (make_golden.py otf) with `ot.emd` replaced
    by an exact scipy solve; the oracle restatement must give the same ground metrics, the same plans (no cost matrix
    of these fixtures has duplicated rows / columns, so the optimal vertex is unique) and the same fused weights.
    Parity stays unpinned at ot.emd itself (POT absent)."""
    z = SubNpz(load_npz('otf'), name)
    models, X = build_models(name, z)
    log = []
    fused = o_otf.otf_fuse(models[:-1], models[-1], X, mode=mode, log=log)
    assert len(log) == len([k for k in z.files if k.startswith(f'{mode}.cost')])
    for rec in log:
        l, k = rec['layer'], rec['model']
        assert int(z[f'{mode}.dup{l}_{k}'][0]) == 0
        np.testing.assert_allclose(rec['cost'], z[f'{mode}.cost{l}_{k}'], rtol=1e-5, atol=1e-6)
        Tg = z[f'{mode}.T{l}_{k}']
        assert (rec['T'].argmax(1) == Tg.argmax(1)).all(), (l, k)
        np.testing.assert_allclose(rec['T'], Tg, rtol=1e-6, atol=1e-9)
    assert_state_close(fused, fused_state(z, mode), rtol=1e-5)
