"""-m gpu: the capture leg's convolution (models/conv.py: im2row + one f32 GEMM) against torch's own conv2d in f64,
its opt-in switch (NIF_CONV_GEMM=1) and its fall-backs (autograd, other dtypes)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # B, C, H, W, cout, kernel, stride, padding, bias
    (5, 3, 32, 32, 8, (3, 3), (1, 1), (1, 1), True),      # VGG's first level
    (4, 64, 8, 8, 128, (3, 3), (1, 1), (1, 1), True),     # a middle level
    (3, 16, 9, 7, 5, (3, 2), (2, 1), (0, 1), False),      # ragged map, strides, no bias
    (2, 7, 5, 5, 3, (1, 1), (1, 1), (0, 0), True),        # 1 x 1
    (1, 1, 4, 4, 1, (4, 4), (1, 1), (0, 0), True),        # a single output pixel
]


@pytest.mark.parametrize('case', CASES)
def test_gemm_conv_equals_conv2d(case, monkeypatch):
    monkeypatch.setenv('NIF_CONV_GEMM', '1')
    from neuron_interpolation_model_fusion_amd.models.conv import GemmConv2d
    B, C, H, W, cout, k, s, p, bias = case
    dev = torch.device('cuda', 0)
    torch.manual_seed(B * 100 + C)
    conv = GemmConv2d(C, cout, kernel_size=k, stride=s, padding=p, bias=bias).to(dev)
    x = torch.randn(B, C, H, W, device=dev)
    with torch.no_grad():
        assert conv._gemm_path(x)
        got = conv(x)
    ref = F.conv2d(x.double(), conv.weight.double(), conv.bias.double() if bias else None, stride=s, padding=p)
    assert got.shape == ref.shape
    # f32 products summed over C * kh * kw terms against the f64 result
    tol = 1e-5 * (C * k[0] * k[1]) ** 0.5
    assert (got.double() - ref).abs().max().item() <= tol * ref.abs().max().item() + 1e-6
    # a channels-last (strided) input is read as the same numbers
    with torch.no_grad():
        again = conv(x.contiguous(memory_format=torch.channels_last))
    assert torch.equal(again, got)


def test_gemm_conv_leaves_autograd_and_the_switch_to_torch(monkeypatch):
    from neuron_interpolation_model_fusion_amd.models.conv import GemmConv2d
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    conv = GemmConv2d(4, 6, kernel_size=3, padding=1).to(dev)
    x = torch.randn(2, 4, 8, 8, device=dev)
    with torch.no_grad():
        assert not conv._gemm_path(x)          # opt-in: off unless NIF_CONV_GEMM=1
    monkeypatch.setenv('NIF_CONV_GEMM', '1')
    assert not conv._gemm_path(x)              # grad enabled, parameters require grad
    y = conv(x)
    y.sum().backward()
    assert conv.weight.grad is not None
    with torch.no_grad():
        assert conv._gemm_path(x)
        assert not conv._gemm_path(x.double())
        fast = conv(x)
        monkeypatch.setenv('NIF_CONV_GEMM', '0')
        assert not conv._gemm_path(x)
        slow = conv(x)
    assert torch.allclose(fast, slow, rtol=1e-4, atol=1e-5)


def test_vgg_forward_and_captures_with_and_without_the_gemm_path(monkeypatch):
    from neuron_interpolation_model_fusion_amd.models import VGG
    from neuron_interpolation_model_fusion_amd.utils.fusion_utils import initialize_activations
    dev = torch.device('cuda', 0)
    torch.manual_seed(1)
    VGG.cfg['VGG_conv_test'] = [8, 'M', 16, 16, 'M', 24, 'M']
    model = VGG('VGG_conv_test', 10, (3, 32, 32), batch_norm=False, use_bias=True, use_activations=False).eval().to(dev)
    X = torch.randn(20, 3, 32, 32, device=dev)
    n = len(model.get_ordered_trainable_named_layers)

    def run():
        initialize_activations(model, X)
        caps = [model.get_inputs_and_preactivations(l) for l in range(n)]
        with torch.no_grad():
            out = model(X)
        model.remove_hooks()
        return [(i.clone(), o.clone()) for i, o in caps], out

    monkeypatch.setenv('NIF_CONV_GEMM', '1')
    fast, out_fast = run()
    monkeypatch.setenv('NIF_CONV_GEMM', '0')
    slow, out_slow = run()
    assert torch.allclose(out_fast, out_slow, rtol=1e-4, atol=1e-5)
    for (i0, o0), (i1, o1) in zip(fast, slow):
        assert i0.shape == i1.shape and o0.shape == o1.shape
        assert torch.allclose(i0, i1, rtol=1e-4, atol=1e-5)
        assert torch.allclose(o0, o1, rtol=1e-4, atol=1e-5)
