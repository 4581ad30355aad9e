"""HIP kernels against PyTorch's own operators (ATen on the CPU, fp32 inputs / fp64 arithmetic) -- ONE hop to the
arithmetic the reference actually runs (nn.Conv2d / ConvTranspose2d / Linear / BatchNorm2d, vae.py:198-216, 264-277), without
the test suite's kernel emulation in between (tests/test_kernels_gpu.py compares with tests/emu_backend.py, which is itself
checked against these operators only through the layer and model tests).  Shapes cover every kernel family a launch can
land on: the wave-specialised ring kernels (64x64 and 128x128 tiles), the register-staged kernels (32-channel outputs, the
k4 s1 p0 walk), the patch-resident transposed convolution, and the weight-gradient kernels (one-tap and four-tap)."""
import pytest
import torch
import torch.nn.functional as F

from mmdyn_hip import layers, ops
from mmdyn_hip.ops import CONV, TCONV_S2P1, DENSE

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def nhwc_rows(x):            # [B, C, H, W] -> [B*H*W, C]
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


def from_rows(r, B, H, C):   # [B*H*H, C] -> [B, C, H, H]
    return r.reshape(B, H, H, C).permute(0, 3, 1, 2)


@pytest.mark.parametrize("B,Hi,Cin,Cout,stride,pad", [(6, 32, 32, 64, 2, 1), (5, 16, 64, 128, 2, 1), (7, 8, 128, 256, 1, 0),
                                                      (3, 64, 32, 32, 2, 1), (70, 16, 64, 128, 2, 1)])
def test_conv2d_forward_and_weight_gradient(B, Hi, Cin, Cout, stride, pad):
    """nn.Conv2d(Cin, Cout, 4, stride, pad): forward (implicit GEMM, CONV mode) and dL/dW (wgrad_tn + wgrad_reduce)."""
    x, W = rnd(B, Cin, Hi, Hi, seed=1), rnd(Cout, Cin, 4, 4, seed=2, scale=0.1)
    Ho = (Hi + 2 * pad - 4) // stride + 1
    Wp = layers.pack_conv(W.to(DEV), swap=False)
    y, _, _ = layers.conv_like(nhwc_rows(x).to(DEV), Wp, CONV, 1, B, Hi, Cin, Ho, Cout, stride, -pad)
    xd, Wd = x.double().requires_grad_(True), W.double().requires_grad_(True)
    ref = F.conv2d(xd, Wd, stride=stride, padding=pad)
    assert rel(from_rows(y, B, Ho, Cout), ref) < 2e-6
    dy = rnd(B, Cout, Ho, Ho, seed=3)
    gx, gW = torch.autograd.grad(ref, (xd, Wd), dy.double())
    gW_hip = torch.zeros(Cout, Cin, 4, 4, device=DEV)
    layers.wgrad(nhwc_rows(dy).to(DEV), nhwc_rows(x).to(DEV), gW_hip, CONV, B, Ho, Cout, Hi, Cin, stride, -pad)
    assert rel(gW_hip, gW) < 5e-6
    if stride == 2:          # dL/dx of a k4 s2 p1 convolution = the k4 s2 p1 TRANSPOSED convolution with the same weights
        Ws = layers.pack_conv(W.to(DEV), swap=True)          # [16][Cin][Cout]
        dx, _, _ = layers.conv_like(nhwc_rows(dy).to(DEV), Ws, TCONV_S2P1, 1, B, Ho, Cout, Hi, Cin)
        assert rel(from_rows(dx, B, Hi, Cin), gx) < 5e-6


@pytest.mark.parametrize("B,Hi,Cin,Cout", [(5, 8, 128, 64), (4, 16, 64, 32), (66, 8, 128, 64), (3, 32, 32, 32)])
def test_conv_transpose2d_s2p1_forward_input_and_weight_gradient(B, Hi, Cin, Cout):
    """nn.ConvTranspose2d(Cin, Cout, 4, 2, 1): forward, dL/dx (= a k4 s2 p1 convolution of dy) and dL/dW."""
    x, W = rnd(B, Cin, Hi, Hi, seed=4), rnd(Cin, Cout, 4, 4, seed=5, scale=0.1)
    Ho = 2 * Hi
    Ws = layers.pack_conv(W.to(DEV), swap=True)               # forward operand: [16][Cout][Cin]
    y, _, _ = layers.conv_like(nhwc_rows(x).to(DEV), Ws, TCONV_S2P1, 1, B, Hi, Cin, Ho, Cout)
    xd, Wd = x.double().requires_grad_(True), W.double().requires_grad_(True)
    ref = F.conv_transpose2d(xd, Wd, stride=2, padding=1)
    assert rel(from_rows(y, B, Ho, Cout), ref) < 2e-6
    dy = rnd(B, Cout, Ho, Ho, seed=6)
    gx, gW = torch.autograd.grad(ref, (xd, Wd), dy.double())
    Wk = layers.pack_conv(W.to(DEV), swap=False)              # input-gradient operand: [16][Cin][Cout]
    dx, _, _ = layers.conv_like(nhwc_rows(dy).to(DEV), Wk, CONV, 1, B, Ho, Cout, Hi, Cin, 2, -1)
    assert rel(from_rows(dx, B, Hi, Cin), gx) < 5e-6
    gW_hip = torch.zeros(Cin, Cout, 4, 4, device=DEV)
    layers.wgrad(nhwc_rows(x).to(DEV), nhwc_rows(dy).to(DEV), gW_hip, CONV, B, Hi, Cin, Ho, Cout, 2, -1)
    assert rel(gW_hip, gW) < 5e-6


@pytest.mark.parametrize("G,Bg", [(1, 5), (4, 70), (4, 256)])
def test_conv_transpose2d_s1p0(G, Bg):
    """nn.ConvTranspose2d(256, 128, 4, 1, 0) on 5x5 inputs (the decoder's first layer): the tap-skipping walk, and below
    256 blocks the column-matrix route."""
    B = G * Bg
    x, W = rnd(B, 256, 5, 5, seed=7), rnd(256, 128, 4, 4, seed=8, scale=0.1)
    Ws = layers.pack_conv(W.to(DEV), swap=True)
    y, _, _ = layers.tconv_s1p0(nhwc_rows(x).to(DEV), Ws, G, Bg, 256, 128)
    ref = F.conv_transpose2d(x.double(), W.double(), stride=1, padding=0)
    assert rel(from_rows(y, B, 8, 128), ref) < 2e-6


@pytest.mark.parametrize("rows,K,N", [(256, 6400, 512), (1024, 512, 512), (1024, 256, 6400), (37, 512, 256), (1000, 6400, 256)])
def test_linear_forward_and_gradients(rows, K, N):
    x, W, b = rnd(rows, K, seed=9), rnd(N, K, seed=10, scale=0.05), rnd(N, seed=11)
    y, _ = layers.dense(x.to(DEV), W.to(DEV), b.to(DEV), rows, K, N)
    xd, Wd = x.double().requires_grad_(True), W.double().requires_grad_(True)
    ref = F.linear(xd, Wd, b.double())
    assert rel(y, ref) < 2e-6
    dy = rnd(rows, N, seed=12)
    gx, gW = torch.autograd.grad(ref, (xd, Wd), dy.double())
    gW_hip = torch.zeros(N, K, device=DEV)
    layers.wgrad(dy.to(DEV), x.to(DEV), gW_hip, DENSE, rows, 1, N, 1, K)
    assert rel(gW_hip, gW) < 5e-6
    Wt = layers.repack(W.to(DEV), N, K, K, N, 1)
    dx, _ = layers.dense(dy.to(DEV), Wt, None, rows, N, K)
    assert rel(dx, gx) < 5e-6


@pytest.mark.parametrize("B,H,C,G", [(8, 16, 64, 1), (8, 8, 128, 4), (6, 32, 32, 2)])
def test_batchnorm_swish_forward_backward(B, H, C, G):
    """Train-mode nn.BatchNorm2d + Swish per group: forward, running buffers, dL/dy, dL/dgamma, dL/dbeta."""
    y = rnd(B, C, H, H, seed=13) * 2 + 0.3
    gamma, beta = rnd(C, seed=14) + 1.5, rnd(C, seed=15)
    rows = nhwc_rows(y).to(DEV)
    Bg = B // G
    T = ops.B.colstats_tiles(Bg * H * H)
    part = torch.zeros(G, T, 2, C, device=DEV)
    ops.B.colstats(rows, part, G, Bg * H * H, C)
    rm, rv, nbt = torch.zeros(C, device=DEV), torch.ones(C, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)
    bn = layers.BNState(gamma.to(DEV), beta.to(DEV), rm, rv, nbt)
    a, mean, rstd = layers.bn_swish_from_partials(rows, part, T, bn, G, Bg * H * H, C)
    mod = torch.nn.BatchNorm2d(C).double().train()
    with torch.no_grad():
        mod.weight.copy_(gamma)
        mod.bias.copy_(beta)
    yd = y.double().requires_grad_(True)
    outs = []
    for g in range(G):
        u = mod(yd[g * Bg:(g + 1) * Bg])
        outs.append(u * torch.sigmoid(u))
    ref = torch.cat(outs)
    assert rel(from_rows(a, B, H, C), ref) < 2e-6
    assert rel(rm, mod.running_mean) < 1e-5 and rel(rv, mod.running_var) < 1e-5 and int(nbt) == G
    da = rnd(B, C, H, H, seed=16)
    gy, gg, gb = torch.autograd.grad(ref, (yd, mod.weight, mod.bias), da.double())
    dgamma, dbeta = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dy = layers.bn_swish_backward(nhwc_rows(da).to(DEV), rows, mean, rstd, bn, dgamma, dbeta, G, Bg * H * H, C)
    assert rel(from_rows(dy, B, H, C), gy) < 1e-5
    assert rel(dgamma, gg) < 1e-5 and rel(dbeta, gb) < 1e-5
