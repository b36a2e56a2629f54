"""Convolution of the capture leg (reference: models/vgg.py:60-75 builds plain nn.Conv2d; the fusions run them
under torch.no_grad() to capture layer inputs and pre-activations, fusion_utils.py:29-35, hf_kf.py:549).

On gfx950 torch's conv goes through MIOpen.  The first time a process (on a machine whose MIOpen user database has
not seen the shape) runs a conv shape, MIOpen tries its solvers -- the per-image im2col + GEMM loop and the naive direct
kernel among them (they are what profiles/r3_e_bench_c1_kernel_stats.csv shows for the warm-up fusion: 41 % of that
profile's GPU time).  The capture forward needs no autograd, so it can be done the way the LS refit already reads a
conv level: ONE im2row launch (nif_im2row_f32, rows (sample, out_y, out_x), columns (in_channel, ky, kx) + a ones
column for the bias) and ONE f32 GEMM against [W | b] -- the same design matrix the refit builds.  The result is handed
on as a channels-last view of the GEMM's (rows, out_channels) output, no transposing copy.

Measured (tools/conv_first_call.sh, VGG11 x 400 samples, profiles/r3_m_conv_first_call.txt): the FIRST Git Re-Basin
fusion on a fresh machine 6.40 s -> 0.36 s (MIOpen searches and compiles per shape once per machine, then keeps the
result in its user database); every later fusion is the same within 3 ms either way (HF 0.225 / 0.227 s, OTF 0.027 /
0.027 s, GR 0.034 / 0.037 s: MIOpen's tuned choice is Winograd, 2.25 x fewer multiplications).  So the path is an
OPT-IN (NIF_CONV_GEMM=1) for one-shot runs on machines MIOpen has not seen the shapes on, and for shapes where its
immediate mode falls back to the per-image GEMM loop; the default stays torch's conv.

Anything the path does not cover (autograd, CPU tensors, groups, dilation, non-zero padding modes, other dtypes) takes
nn.Conv2d's own forward.
"""
import os

import torch
import torch.nn as nn

from .. import ops


def conv2d_gemm(x, weight, bias, stride, padding):
    """F.conv2d(x, weight, bias, stride, padding) for a CUDA f32 [B, C, H, W] input as im2row + one GEMM.
    Returns [B, out_channels, OH, OW] in channels-last strides."""
    B, _, H, W = x.shape
    cout, _, kh, kw = weight.shape
    Xp = ops.design_matrix(x, (kh, kw), stride, padding, ones=bias is not None)
    Wm = weight.reshape(cout, -1)
    if bias is not None:
        Wm = torch.cat([Wm, bias.reshape(cout, 1)], dim=1)
    out = torch.mm(Xp, Wm.t())
    OH = (H + 2 * padding[0] - kh) // stride[0] + 1
    OW = (W + 2 * padding[1] - kw) // stride[1] + 1
    return out.view(B, OH, OW, cout).permute(0, 3, 1, 2)


class GemmConv2d(nn.Conv2d):
    """nn.Conv2d (same parameters, same state_dict keys) whose no-grad GPU forward is im2row + GEMM under NIF_CONV_GEMM=1."""

    def _gemm_path(self, x):
        if os.environ.get('NIF_CONV_GEMM', '0') != '1':
            return False
        if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and self.weight.dtype == torch.float32):
            return False
        if self.groups != 1 or tuple(self.dilation) != (1, 1) or self.padding_mode != 'zeros' \
                or isinstance(self.padding, str):
            return False
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad
                                        or (self.bias is not None and self.bias.requires_grad)):
            return False
        return True

    def forward(self, x):
        if self._gemm_path(x):
            return conv2d_gemm(x, self.weight, self.bias, tuple(self.stride), tuple(self.padding))
        return super().forward(x)
