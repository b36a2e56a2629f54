"""CPU tests of host-side logic that has to reproduce torch's own bookkeeping draw for draw."""
import torch
from torch.utils.data import DataLoader, TensorDataset, random_split

from neuron_interpolation_model_fusion_amd.fusion_algorithms.alf import ALFusion


def _reference_orders(n, batch_size, epochs, seed):
    """What the reference's loaders (alf.py:517-520) visit: per epoch the shuffled train pass, then the val pass."""
    torch.manual_seed(seed)
    train, val = random_split(TensorDataset(torch.arange(n)), [0.8, 0.2])
    train_dl = DataLoader(train, batch_size=batch_size, shuffle=True)
    val_dl = DataLoader(val, batch_size=batch_size, shuffle=False)
    out = []
    for _ in range(epochs):
        out.append(torch.cat([b[0] for b in train_dl]))
        out.append(torch.cat([b[0] for b in val_dl]))
    return out, torch.rand(3)  # the trailing draw shows the generator ended in the same state


def test_on_device_batching_visits_samples_in_the_loaders_order():
    for n, bs, seed in ((37, 8, 5), (10, 32, 0), (101, 7, 123)):
        ref, ref_tail = _reference_orders(n, bs, 3, seed)
        torch.manual_seed(seed)
        train_idx, val_idx = ALFusion._split_indices(n)
        got = []
        for _ in range(3):
            got.append(train_idx[ALFusion._loader_order(train_idx.shape[0], True)])
            ALFusion._loader_order(val_idx.shape[0], False)
            got.append(val_idx)
        tail = torch.rand(3)
        assert len(ref) == len(got)
        for r, g in zip(ref, got):
            assert torch.equal(r, g)
        assert torch.equal(ref_tail, tail)


def test_eval_model_metrics_on_a_known_classifier():
    from neuron_interpolation_model_fusion_amd.utils.run_utils import eval_model
    torch.manual_seed(0)
    n, c = 200, 4
    y = torch.arange(n) % c
    logits = torch.nn.functional.one_hot(y, c).float() * 4 + torch.randn(n, c)

    class Fixed(torch.nn.Module):
        def forward(self, idx):
            return logits[idx.long().reshape(-1)]
    batches = [(torch.arange(i, min(i + 32, n)), y[i:i + 32]) for i in range(0, n, 32)]
    stats = eval_model(Fixed(), batches, device='cpu', num_classes=c)
    pred = logits.argmax(1)
    assert abs(stats['accuracy'] - float((pred == y).double().mean())) < 1e-12
    cm = stats['confusion_matrix']
    assert cm.sum() == n and int(cm.diag().sum()) == int((pred == y).sum())
    for t in range(c):
        for p in range(c):
            assert int(cm[t, p]) == int(((y == t) & (pred == p)).sum())
    assert abs(stats['loss'] - torch.nn.functional.cross_entropy(logits, y).item()) < 1e-6
    assert 0.5 < stats['roc_auc'] <= 1.0 and 0.0 < stats['f1'] <= 1.0


REF_CKPT = '/root/reference/example-base-models/non-iid/MLPs/MNIST/split-by-2/seed1'


def test_reference_whole_module_checkpoints_load_into_our_classes(tmp_path):
    """Data files of the reference (safe loader only); skipped where the reference tree is absent."""
    import os
    import pytest
    if not os.path.isdir(REF_CKPT):
        pytest.skip('reference checkpoints not present on this machine')
    from neuron_interpolation_model_fusion_amd.models import MLPNet
    from neuron_interpolation_model_fusion_amd.utils.run_utils import load_model
    from tests.helpers import load_npz
    m = load_model(os.path.join(REF_CKPT, 'model_0.pt'))
    assert type(m) is MLPNet and (m.input_size, m.hidden_size, m.output_size) == (784, 256, 10)
    z = load_npz('mlp_real')  # the same checkpoint's weights as minted into the golden fixture
    for k, v in m.state_dict().items():
        assert torch.equal(v, torch.from_numpy(z[f'sd0.{k}'])), k
    assert len(m.train_indices) > 0
    assert m(torch.zeros(2, 1, 28, 28)).shape == (2, 10)
    # a model saved from here loads back through the same allow-list
    torch.save(m.copy_model(), tmp_path / 'again.pt')
    again = load_model(tmp_path / 'again.pt')
    assert all(torch.equal(a, b) for a, b in zip(again.state_dict().values(), m.state_dict().values()))


def test_forward_level_walks_the_same_outputs_as_forward_until_level():
    from neuron_interpolation_model_fusion_amd.models import MLPNet, VGG, ViT
    torch.manual_seed(0)
    VGG.cfg.setdefault('VGG_fl_test', [6, 'M', 8, 8, 'M'])
    cases = [
        (MLPNet(48, 20, 7, use_bias=True, use_activations=False), torch.randn(5, 3, 4, 4)),
        (MLPNet(48, 20, 7, use_bias=True, use_activations=True), torch.randn(5, 48)),
        (VGG('VGG_fl_test', 10, (3, 8, 8), batch_norm=True, use_bias=True, use_activations=False).eval(), torch.randn(4, 3, 8, 8)),
        (ViT(in_c=3, num_classes=10, img_size=8, patch=4, depth=3, hidden=16, mlp_hidden=32, head=2).eval(), torch.randn(4, 3, 8, 8)),
    ]
    for model, X in cases:
        with torch.no_grad():
            acts = X
            for level in range(len(model.get_ordered_trainable_named_levels)):
                acts = model.forward_level(acts, level)
                assert torch.equal(acts, model.forward_until_level(X, level)), (type(model).__name__, level)
            assert torch.equal(acts, model(X))


def test_twin_copy_default_replays_the_init_draws_and_fast_mode_skips_them():
    """copy_model() under twin_copy(): by default torch's generator advances exactly as a plain copy_model() (the
    reference's behaviour, models/mlp.py:46-53); in the opt-in fast mode the generator is untouched.  Same weights
    either way."""
    import torch
    from neuron_interpolation_model_fusion_amd.models import MLPNet
    from neuron_interpolation_model_fusion_amd.models.base_model import twin_copy
    assert twin_copy.replay_init_draws is True  # parity mode is the default (NIF_FAST_TWIN_COPY unset)
    torch.manual_seed(5)
    m = MLPNet(20, 16, 4, use_bias=True, use_activations=False)
    state = torch.get_rng_state()
    plain = m.copy_model()
    after_plain = torch.get_rng_state()
    assert not torch.equal(after_plain, state)
    torch.set_rng_state(state)
    with twin_copy():
        c = m.copy_model()
    assert torch.equal(torch.get_rng_state(), after_plain)
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), c.state_dict().values()))
    torch.set_rng_state(state)
    twin_copy.replay_init_draws = False
    try:
        with twin_copy():
            fast = m.copy_model()
        assert torch.equal(torch.get_rng_state(), state)
        assert torch.nn.init.kaiming_uniform_.__module__ == 'torch.nn.init'  # restored
    finally:
        twin_copy.replay_init_draws = True
    assert all(torch.equal(a, b) for a, b in zip(fast.state_dict().values(), plain.state_dict().values()))


def test_alf_last_level_grouping_and_criterion_branches_match_reference_golden():
    """alf.py:398-454 on the host class itself (plain torch, runs on the CPU): the class-wise logit targets with /
    without head weights and softening against tests/golden/alf_kd.npz, and the criterion each branch selects."""
    import numpy as np
    from neuron_interpolation_model_fusion_amd.fusion_algorithms import ALFusionConfig
    from tests.helpers import load_npz, build_models
    z = load_npz('alf_kd')
    models, X = build_models('mlp_kd', z)
    head_w = torch.from_numpy(z['head_w'])
    alf = ALFusion(models)
    alf.ni_scores = [{1: torch.ones(10)} for _ in models]

    def cfg(**kw):
        return ALFusionConfig(X=X, y=None, optimizer_per_level=['sgd'] * 2, lr_per_level=[0.05] * 2,
                              weight_decay_per_level=[0.0] * 2, epochs_per_level=[1] * 2, verbose=False, **kw)
    for hw in (0, 1):
        for kd in (0, 1):
            T, assign = alf._last_level_grouping(10, cfg(cls_head_weights=head_w if hw else None,
                                                          use_kd_on_head=bool(kd), kd_temperature=2.0))
            assert (assign.numpy() == z[f'grp.assign.{hw}{kd}']).all()
            np.testing.assert_allclose(T.numpy(), z[f'grp.T.{hw}{kd}'], rtol=1e-6, atol=1e-7)
    assign = torch.cat([torch.arange(10), torch.arange(10)])
    kl = alf._get_criterion(1, 10, assign, cfg(use_weighted_loss=True, cls_head_weights=head_w, use_kd_on_head=True), True)
    assert isinstance(kl, torch.nn.KLDivLoss) and kl.reduction == 'batchmean'
    for kw in (dict(use_weighted_loss=False, cls_head_weights=head_w, use_kd_on_head=True),
               dict(use_weighted_loss=True, cls_head_weights=None, use_kd_on_head=True)):
        assert alf._get_criterion(1, 10, assign, cfg(**kw), True).__name__ == 'mse_loss'
    wl = alf._get_criterion(1, 10, assign, cfg(use_weighted_loss=True, cls_head_weights=head_w), True)
    assert wl.__name__ == 'weighted_loss'
    # a hidden level never takes the head branches
    assert alf.ni_scores[0].setdefault(0, torch.ones(32)) is not None and alf.ni_scores[1].setdefault(0, torch.ones(32)) is not None
    hidden = alf._get_criterion(0, 32, torch.cat([torch.arange(32)] * 2),
                                cfg(use_weighted_loss=True, cls_head_weights=head_w, use_kd_on_head=True), False)
    assert hidden.__name__ == 'weighted_loss'


def test_hook_capture_aliases_unless_the_model_has_an_inplace_module():
    """K1: captures are the forward's own tensors (no clone pass) unless an in-place module could overwrite them."""
    import torch
    from torch import nn
    from neuron_interpolation_model_fusion_amd.models import MLPNet
    torch.manual_seed(0)
    m = MLPNet(12, 16, 4, use_bias=True, use_activations=True).eval()
    x = torch.randn(5, 1, 3, 4)
    m.register_hooks()
    with torch.no_grad():
        out = m(x)
    inp, pre = m.get_inputs_and_preactivations(1)
    assert pre.data_ptr() == out.data_ptr()  # the last layer's capture IS the output tensor
    fc1 = m.get_ordered_trainable_named_layers[0].layer
    ref = fc1(x.flatten(1))
    assert torch.allclose(m.get_inputs_and_preactivations(0)[1], ref)
    m.remove_hooks()
    # an in-place activation after the layer: the capture must survive it
    m.level1.append(nn.ReLU(inplace=True))
    m.register_hooks()
    with torch.no_grad():
        m(x)
    pre0 = m.get_inputs_and_preactivations(0)[1]
    assert torch.allclose(pre0, ref) and bool((pre0 < 0).any())
    m.remove_hooks()


def test_hook_capture_clones_by_default_and_reports_inplace_writes_on_aliasing_classes():
    """A BaseModel subclass that does not opt in gets the reference's clones (base_model.py:110-118), so a functional
    in-place op in its forward cannot corrupt the captures; a class that opted in and is written to anyway raises when
    the capture is read."""
    import pytest
    import torch
    import torch.nn.functional as F
    from neuron_interpolation_model_fusion_amd.models import MLPNet

    class InplaceMLP(MLPNet):
        captures_may_alias = False

        def forward(self, x):
            h = self.level1(x)
            F.relu(h, inplace=True)     # functional in-place op: no module attribute gives it away
            return self.level2(h)

    torch.manual_seed(0)
    m = InplaceMLP(12, 16, 4, use_bias=True, use_activations=False).eval()
    x = torch.randn(5, 1, 3, 4)
    ref = m.get_ordered_trainable_named_layers[0].layer(x.flatten(1))
    m.register_hooks()
    with torch.no_grad():
        m(x)
    pre0 = m.get_inputs_and_preactivations(0)[1]
    assert torch.allclose(pre0, ref) and bool((pre0 < 0).any())   # the clone kept the pre-activations
    m.remove_hooks()

    class WronglyAliasing(InplaceMLP):
        captures_may_alias = True

    torch.manual_seed(0)
    w = WronglyAliasing(12, 16, 4, use_bias=True, use_activations=False).eval()
    w.register_hooks()
    with torch.no_grad():
        w(x)
    with pytest.raises(RuntimeError, match='modified in place'):
        w.get_inputs_and_preactivations(0)
    w.remove_hooks()


def test_alf_flatten_and_join_equals_flatten_normalise_concatenate():
    """ALFusion._flatten_and_join writes each model's block straight into the joined matrices; the values are those of
    the reference's separate steps (alf.py:104-115, 181-189), bit for bit."""
    import torch
    from neuron_interpolation_model_fusion_amd.fusion_algorithms.alf import ALFusion
    from neuron_interpolation_model_fusion_amd.utils.fusion_utils import normalize_activations_per_sample
    g = torch.Generator().manual_seed(3)
    for shapes in ([(7, 5), (7, 9)], [(4, 6, 3), (4, 10, 3)], [(3, 4, 2, 5), (3, 8, 2, 5)]):
        acts = [torch.randn(*sh, generator=g) for sh in shapes]
        flat = [ALFusion._flatten_acts(None, a) for a in acts]
        want = torch.cat(flat, dim=1)
        want_n = torch.cat([normalize_activations_per_sample(f) for f in flat], dim=1)
        joined, normed = ALFusion._flatten_and_join(None, acts, True)
        assert joined.is_contiguous() and normed.is_contiguous()
        assert torch.equal(joined, want) and torch.equal(normed, want_n)
        joined2, same = ALFusion._flatten_and_join(None, acts, False)
        assert same is joined2 and torch.equal(joined2, want)


def test_duplicate_line_filter_of_the_scipy_order_assignment():
    """ops._has_duplicate_lines (the cheap test that spares a fast solve whose certificate would fail): identical rows or
    identical columns are found wherever they sit; matrices without them -- also ones whose lines share a sum -- pass."""
    from neuron_interpolation_model_fusion_amd import ops
    g = torch.Generator().manual_seed(3)
    C = torch.rand(40, 40, generator=g)
    assert not bool(ops._has_duplicate_lines(C))
    D = C.clone()
    D[31] = D[4]
    assert bool(ops._has_duplicate_lines(D))
    D = C.clone()
    D[:, 0] = D[:, 39]
    assert bool(ops._has_duplicate_lines(D))
    # equal sums, different maxima: not a duplicate (a false alarm would only cost the slower replay kernel)
    E = torch.randint(100, 1000, (40, 40), generator=g).float()   # integers: the sums are exact
    E[7] = E[9]
    k = int(E[9].argmax())
    E[7, k] += 8.0
    E[7, (k + 1) % 40] -= 8.0
    assert float(E[7].double().sum()) == float(E[9].double().sum())
    assert not bool(ops._has_duplicate_lines(E))
    # dead units of a random-initialised network: all-zero activations give identical cost lines
    Z = C.clone()
    Z[:, 5] = 1.5
    Z[:, 17] = 1.5
    assert bool(ops._has_duplicate_lines(Z))
