"""Reference-held known answers run through the HIP kernels themselves (tests/golden/small_ops.npz, produced by
running the reference: tests/golden/make_golden.py::gen_small_ops): the ProductOfExperts vectors with extreme
log-variances (the double-eps edge case of vae.py:311-318), the plain and masked ELBO values of
problems.py:401-458, and the 3-channel loss mask the dataset reader yields (datasets.py: segmentation PNG)."""
import os

import numpy as np
import pytest
import torch

from mmdyn_hip import ops
from oracle import mvae_oracle as O

pytestmark = pytest.mark.gpu

HIP = ops.HipBackend()
DEV = "cuda"


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "small_ops.npz"))


def _poe_pass(mu, lv, dmu=None, dlv=None):
    """Experts 1..3 of the golden tensor as one pass descriptor (expert 0 is the prior: with_prior=1)."""
    n = mu.shape[0]
    pad = [None] * (3 - n)
    return [{"mu": list(mu) + pad, "lv": list(lv) + pad, "dmu": (list(dmu) if dmu is not None else [None] * n) + pad,
             "dlv": (list(dlv) if dlv is not None else [None] * n) + pad, "ld": [mu.shape[-1]] * 3}]


def test_poe_extreme_logvar_matches_reference(g):
    mu, lv = torch.tensor(g["poe/mu"]), torch.tensor(g["poe/logvar"])         # [4 experts][5][16]; expert 0 = prior
    assert float(mu[0].abs().max()) == 0.0 and float(lv[0].abs().max()) == 0.0
    B, L = mu.shape[1], mu.shape[2]
    mud, lvd = mu[1:].contiguous().to(DEV), lv[1:].contiguous().to(DEV)
    out_mu, out_lv, z = (torch.empty(1, B, L, device=DEV) for _ in range(3))
    kl = torch.zeros(1, dtype=torch.float64, device=DEV)
    eps = torch.zeros(1, B, L, device=DEV)
    HIP.poe_fwd(_poe_pass(mud, lvd), eps, out_mu, out_lv, z, kl, 1, 1, B, L)
    # the reference's own outputs, including lv = -30 / -18 / +20 / +40 in one expert
    np.testing.assert_allclose(out_mu[0].cpu().numpy(), g["poe/out_mu"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(out_lv[0].cpu().numpy(), g["poe/out_logvar"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(z[0].cpu().numpy(), g["poe/out_mu"], rtol=2e-6, atol=1e-7)       # eps = 0: z = mu
    kl_ref = O.kl_divergence(torch.tensor(g["poe/out_mu"]), torch.tensor(g["poe/out_logvar"]))
    assert float(kl.cpu()) == pytest.approx(float(kl_ref), rel=1e-5)


def test_poe_backward_extreme_logvar_matches_oracle_autograd(g):
    mu = torch.tensor(g["poe/mu"])[1:].clone().requires_grad_(True)
    lv = torch.tensor(g["poe/logvar"])[1:].clone().requires_grad_(True)
    B, L = mu.shape[1], mu.shape[2]
    gen = torch.Generator().manual_seed(11)
    eps, dz = torch.randn(B, L, generator=gen), torch.randn(B, L, generator=gen)
    klw = 0.37 / B
    prior = torch.zeros(1, B, L)
    pm, plv = O.product_of_experts(torch.cat([prior, mu]), torch.cat([prior, lv]))
    zz = O.reparametrize(pm, plv, eps)
    (klw * O.kl_divergence(pm, plv) + (zz * dz).sum()).backward()
    mud, lvd = mu.detach().to(DEV), lv.detach().to(DEV)
    dmu, dlv = torch.zeros_like(mud), torch.zeros_like(lvd)
    HIP.poe_bwd(_poe_pass(mud, lvd, dmu, dlv), eps[None].to(DEV), pm.detach()[None].to(DEV), plv.detach()[None].to(DEV),
                dz[None].to(DEV), None, None, klw, 1, 1, B, L)
    # gradients span 1e-13 .. 1e+8 across the extreme entries: compare element-wise, relative
    np.testing.assert_allclose(dmu.cpu().numpy(), mu.grad.numpy(), rtol=2e-4, atol=1e-12)
    np.testing.assert_allclose(dlv.cpu().numpy(), lv.grad.numpy(), rtol=2e-4, atol=1e-12)


def _bce(logits, target, mask=None, mask_channels=1):
    x, t = logits.contiguous().to(DEV), target.contiguous().to(DEV)
    acc = torch.zeros(1, dtype=torch.float64, device=DEV)
    d = torch.empty_like(x)
    n = x.numel()
    hw = x.shape[-1] * x.shape[-2]
    HIP.bce_logits(x, t, None if mask is None else mask.contiguous().to(DEV), d, acc, n, n // x.shape[0], hw, 1.0,
                   mask_channels)
    return acc, d


def _mse(r, t):
    acc = torch.zeros(1, dtype=torch.float64, device=DEV)
    HIP.mse(r.contiguous().to(DEV), t.contiguous().to(DEV), None, acc, r.numel(), 1.0)
    return acc


def _kl(means, log_var):
    B, L = means.shape
    acc = torch.zeros(1, dtype=torch.float64, device=DEV)
    HIP.reparam_fwd(means.contiguous().to(DEV), log_var.contiguous().to(DEV), None, None, acc, B, L, L)
    return acc


def _assemble(bce, mse, kl, B, klw, pm=1000.0):
    loss, partials = torch.zeros(1, device=DEV), torch.zeros(8, device=DEV)
    zero = torch.zeros(1, dtype=torch.float64, device=DEV)
    HIP.elbo_assemble(bce if bce is not None else zero, mse if mse is not None else zero, kl, loss, partials, 1, B, klw, pm)
    return float(loss.cpu())


def test_elbo_known_answers_through_the_kernels(g):
    T = lambda k: torch.tensor(g["elbo/" + k])
    klw, B = float(g["elbo/kl_weight"]), 3
    kl = _kl(T("means"), T("log_var"))
    bv, _ = _bce(T("rv"), T("xv"))
    bt, _ = _bce(T("rt"), T("xt"))
    mp = _mse(T("rp"), T("xp"))
    assert _assemble(bv + bt, mp, kl, B, klw) == pytest.approx(float(g["elbo/mvae_vtp"]), rel=1e-5)
    assert _assemble(bv, None, kl, B, klw) == pytest.approx(float(g["elbo/mvae_v"]), rel=1e-5)
    assert _assemble(None, mp, kl, B, klw) == pytest.approx(float(g["elbo/mvae_p"]), rel=1e-5)
    assert _assemble(bv, None, kl, B, klw) == pytest.approx(float(g["elbo/vae"]), rel=1e-5)
    # loss mask [B,1,H,W] broadcast over the channels (problems.py:445-447)
    bvm, _ = _bce(T("rv"), T("xv"), T("mask"))
    btm, _ = _bce(T("rt"), T("xt"), T("mask"))
    assert _assemble(bvm, None, kl, B, klw) == pytest.approx(float(g["elbo/vae_masked"]), rel=1e-5)
    assert _assemble(bvm + btm, None, kl, B, klw) == pytest.approx(float(g["elbo/mvae_vt_masked"]), rel=1e-5)


@pytest.mark.parametrize("B,H", [(5, 64), (2, 8)])
def test_three_channel_loss_mask(B, H):
    """The dataset reader's segmentation mask is [B,3,H,W]; the reference multiplies element-wise."""
    gen = torch.Generator().manual_seed(77 + B)
    x = (torch.randn(B, 3, H, H, generator=gen) * 4).requires_grad_(True)
    t = torch.rand(B, 3, H, H, generator=gen)
    mask = (torch.rand(B, 3, H, H, generator=gen) > 0.4).float()
    ref = torch.nn.functional.binary_cross_entropy_with_logits(x * mask, t * mask, reduction="sum")
    ref.backward()
    acc, d = _bce(x.detach(), t, mask, 3)
    assert float(acc.cpu()) == pytest.approx(float(ref), rel=1e-5)
    torch.testing.assert_close(d.cpu(), x.grad, rtol=1e-4, atol=1e-6)
    # the 1-channel form of the same mask must differ (the bug this guards against read it as [B,1,H,W])
    if H == 64:
        acc1, _ = _bce(x.detach(), t, mask[:, :1].contiguous(), 1)
        assert abs(float(acc1.cpu()) - float(ref)) > 1e-3 * abs(float(ref))
    with pytest.raises(Exception):
        _bce(x.detach(), t, mask[:, :2].contiguous(), 2)


def test_three_channel_mask_through_the_module_loss():
    """BCEWithLogitsSumFn (what --mask-loss uses) with the [B,3,H,W] mask == oracle elbo_loss with that mask."""
    from mmdyn_hip.models import functional as Fn
    gen = torch.Generator().manual_seed(5)
    x = (torch.randn(4, 3, 64, 64, generator=gen) * 3)
    t = torch.rand(4, 3, 64, 64, generator=gen)
    mask = (torch.rand(4, 3, 64, 64, generator=gen) > 0.5).float()
    mu, lv = torch.randn(4, 256, generator=gen), torch.randn(4, 256, generator=gen)
    want = O.elbo_loss(x, t, mu, lv, 0.5, loss_mask=mask)
    xd = x.to(DEV).requires_grad_(True)
    bce = Fn.BCEWithLogitsSumFn.apply(xd, t.to(DEV), mask.to(DEV))
    kl = _kl(mu, lv)
    got = (float(bce.detach().cpu()) + 0.5 * float(kl.cpu())) / 4
    assert got == pytest.approx(float(want), rel=1e-5)
    bce.backward()
    xr = x.clone().requires_grad_(True)
    torch.nn.functional.binary_cross_entropy_with_logits(xr * mask, t * mask, reduction="sum").backward()
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=1e-4, atol=1e-6)
