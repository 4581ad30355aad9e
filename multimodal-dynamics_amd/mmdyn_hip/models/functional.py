"""torch.autograd glue: each Function's forward/backward is a fixed sequence of libmmdyn_hip kernels
(:mod:`mmdyn_hip.layers`).  PyTorch only threads the tensors through its graph.

Reference ops replaced (paths relative to /root/reference/mmdyn/pytorch):
  models/vae.py:224-242 Encoder.forward, :285-301 Decoder.forward, :311-318 ProductOfExperts,
  :52-61 reparametrize, :331-334 Swish; problems/problems.py:401-458 BCE / MSE / KL terms.
"""
import torch

from .. import layers, ops
from .shapes import DROPOUT_P


def _scaled(x, g):
    """x * g for a 0-dim upstream gradient g living on the device (no .item() sync)."""
    out = torch.empty_like(x)
    ops.B.scale_dev(x, g.detach().reshape(1).to(torch.float32).contiguous(), out)
    return out


def _dict(keys, tensors):
    return {k: t for k, t in zip(keys, tensors)}


class ImageEncoderTrunkFn(torch.autograd.Function):
    """conv_net + fc_net (before dropout).  ``holder`` carries the BatchNorm buffers."""

    @staticmethod
    def forward(ctx, x, holder, *params):
        keys = holder.param_keys()
        P = _dict(keys, [p.detach() for p in params])
        h, c = layers.encoder_trunk_forward(P, holder.bn_buffers(), x.detach().contiguous(), G=1)
        ctx.P, ctx.c, ctx.keys = P, c, keys
        return h

    @staticmethod
    def backward(ctx, dh):
        grads = {k: torch.empty_like(ctx.P[k]) for k in ctx.keys}
        layers.encoder_trunk_backward(ctx.P, ctx.c, dh.contiguous(), grads)
        ctx.c = None
        return (None, None) + tuple(grads[k] for k in ctx.keys)


class ImageDecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, holder, *params):
        keys = holder.param_keys()
        P = _dict(keys, [p.detach() for p in params])
        out, c = layers.decoder_forward(P, holder.bn_buffers(), z.detach().contiguous(), G=1, cond=holder._cond)
        ctx.P, ctx.c, ctx.keys = P, c, keys
        return out

    @staticmethod
    def backward(ctx, dout):
        grads = {k: torch.empty_like(ctx.P[k]) for k in ctx.keys}
        dz = layers.decoder_backward(ctx.P, ctx.c, dout.contiguous(), grads, need_dz=ctx.needs_input_grad[0])
        ctx.c = None
        return (dz, None) + tuple(grads[k] for k in ctx.keys)


class HeadsFn(torch.autograd.Function):
    """Fused linear_means | linear_log_var: returns [rows, 2L]."""

    @staticmethod
    def forward(ctx, hd, Wm, bm, Wl, bl, cond=None):
        P = _dict(layers.HEAD_KEYS, [Wm.detach(), bm.detach(), Wl.detach(), bl.detach()])
        out, c = layers.heads_forward(P, hd.detach().contiguous(), cond=cond)
        ctx.P, ctx.c = P, c
        return out

    @staticmethod
    def backward(ctx, dout):
        grads = {k: torch.empty_like(ctx.P[k]) for k in layers.HEAD_KEYS}
        dx = layers.heads_backward(ctx.c, dout.contiguous(), grads, need_dx=ctx.needs_input_grad[0])
        return (dx,) + tuple(grads[k] for k in layers.HEAD_KEYS) + (None,)


class DropoutFn(torch.autograd.Function):
    """x * keep_mask / (1 - p) with an explicit uint8 keep-mask (vae.py:213)."""

    @staticmethod
    def forward(ctx, h, mask):
        B, H = h.shape
        out = torch.empty_like(h)
        ops.B.dropout_expand(h.detach().contiguous(), mask, out, 1, B, H, DROPOUT_P)
        ctx.mask = mask
        return out

    @staticmethod
    def backward(ctx, dout):
        B, H = dout.shape
        dh = torch.empty_like(dout)
        ops.B.dropout_reduce(dout.contiguous(), ctx.mask, dh, 1, B, H, DROPOUT_P)
        return dh, None


class PoseEncoderTrunkFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pose, *params):
        P = _dict(layers.POSE_ENC_KEYS, [p.detach() for p in params])
        h, c = layers.pose_encoder_trunk_forward(P, pose.detach().contiguous())
        ctx.P, ctx.c = P, c
        return h

    @staticmethod
    def backward(ctx, dh):
        grads = {k: torch.empty_like(ctx.P[k]) for k in layers.POSE_ENC_KEYS}
        layers.pose_encoder_trunk_backward(ctx.P, ctx.c, dh.contiguous(), grads)
        return (None,) + tuple(grads[k] for k in layers.POSE_ENC_KEYS)


class PoseDecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, *params):
        P = _dict(layers.POSE_DEC_KEYS, [p.detach() for p in params])
        out, c = layers.pose_decoder_forward(P, z.detach().contiguous())
        ctx.P, ctx.c = P, c
        return out

    @staticmethod
    def backward(ctx, dout):
        grads = {k: torch.empty_like(ctx.P[k]) for k in layers.POSE_DEC_KEYS}
        dz = layers.pose_decoder_backward(ctx.P, ctx.c, dout.contiguous(), grads, need_dz=ctx.needs_input_grad[0])
        return (dz,) + tuple(grads[k] for k in layers.POSE_DEC_KEYS)


class MLPFn(torch.autograd.Function):
    """``mlp(sizes, ReLU, Identity)`` of the reference (vae.py:14-19) for any widths: Linear layers with ReLU between
    them, none after the last.  MFMA GEMMs where both widths are multiples of 32, the small kernel otherwise."""

    @staticmethod
    def forward(ctx, x, *params):
        Ws, bs = [p.detach() for p in params[0::2]], [p.detach() for p in params[1::2]]
        hs = [x.detach().contiguous()]
        for i, (W, b) in enumerate(zip(Ws, bs)):
            last = i == len(Ws) - 1
            _, h = layers.linear_forward(hs[-1], W, b, layers.ACT_NONE if last else layers.ACT_RELU)
            hs.append(h)
        ctx.Ws, ctx.hs = Ws, hs
        return hs[-1]

    @staticmethod
    def backward(ctx, dout):
        Ws, hs = ctx.Ws, ctx.hs
        grads = [None] * (2 * len(Ws))
        d = dout.contiguous()
        for i in range(len(Ws) - 1, -1, -1):
            if i < len(Ws) - 1:
                d = layers.act_backward(d, hs[i + 1], layers.ACT_RELU)      # ReLU: output sign == input sign
            gW, gb = torch.empty_like(Ws[i]), torch.empty(Ws[i].shape[0], device=d.device)
            d = layers.linear_backward(d, hs[i], Ws[i], gW, gb, need_dx=(i > 0 or ctx.needs_input_grad[0]))
            grads[2 * i], grads[2 * i + 1] = gW, gb
        return (d,) + tuple(grads)


class PoEReparamFn(torch.autograd.Function):
    """prior + up to three fused-head experts -> (means, log_var, z) in one kernel (vae.py:139-159)."""

    @staticmethod
    def forward(ctx, eps_noise, L, *heads):
        present = [h for h in heads if h is not None]
        B = present[0].shape[0]
        hs = [None if h is None else h.detach().contiguous() for h in heads]
        ref = present[0]
        mu, lv, z = (torch.empty(B, L, device=ref.device, dtype=ref.dtype) for _ in range(3))
        ops.B.poe_fwd([_pass(hs, None, L)], eps_noise, mu, lv, z, None, True, 1, B, L)
        ctx.hs, ctx.eps, ctx.L, ctx.B = hs, eps_noise, L, B
        ctx.save_for_backward(mu, lv)
        return mu, lv, z

    @staticmethod
    def backward(ctx, g_mu, g_lv, dz):
        mu, lv = ctx.saved_tensors
        ds = [None if h is None else torch.zeros_like(h) for h in ctx.hs]
        c = lambda t: None if t is None else t.contiguous()
        ops.B.poe_bwd([_pass(ctx.hs, ds, ctx.L)], ctx.eps, mu, lv, c(dz), c(g_mu), c(g_lv), 0.0, True, 1, ctx.B, ctx.L)
        return (None, None) + tuple(ds)


def _pass(hs, ds, L):
    n = len(hs)
    return {"mu": [None if h is None else h[:, :L] for h in hs],
            "lv": [None if h is None else h[:, L:] for h in hs],
            "dmu": [None if (ds is None or d is None) else d[:, :L] for d in (ds or [None] * n)],
            "dlv": [None if (ds is None or d is None) else d[:, L:] for d in (ds or [None] * n)],
            "ld": [2 * L] * n}


class ProductOfExpertsFn(torch.autograd.Function):
    """Stand-alone ProductOfExperts.forward(mu[M,B,D], logvar[M,B,D]) (vae.py:311-318); M <= 4."""

    @staticmethod
    def forward(ctx, mu, logvar):
        M, B, D = mu.shape
        if M > 4:
            raise ValueError("mmdyn_hip ProductOfExperts kernel supports at most 4 experts")
        mu_c, lv_c = mu.detach().contiguous(), logvar.detach().contiguous()
        out_mu, out_lv = torch.empty_like(mu_c[0]), torch.empty_like(mu_c[0])
        p = {"mu": [mu_c[m] for m in range(M)], "lv": [lv_c[m] for m in range(M)], "ld": [D] * M}
        ops.B.poe_fwd([p], None, out_mu, out_lv, None, None, False, 1, B, D)
        ctx.save_for_backward(mu_c, lv_c, out_mu, out_lv)
        return out_mu, out_lv

    @staticmethod
    def backward(ctx, g_mu, g_lv):
        mu_c, lv_c, out_mu, out_lv = ctx.saved_tensors
        M, B, D = mu_c.shape
        dmu, dlv = torch.zeros_like(mu_c), torch.zeros_like(lv_c)
        p = {"mu": [mu_c[m] for m in range(M)], "lv": [lv_c[m] for m in range(M)],
             "dmu": [dmu[m] for m in range(M)], "dlv": [dlv[m] for m in range(M)], "ld": [D] * M}
        c = lambda t: None if t is None else t.contiguous()
        ops.B.poe_bwd([p], None, out_mu, out_lv, None, c(g_mu), c(g_lv), 0.0, False, 1, B, D)
        return dmu, dlv


class ReparamFn(torch.autograd.Function):
    """z = eps * exp(log_var / 2) + means (vae.py:57-59)."""

    @staticmethod
    def forward(ctx, means, log_var, eps_noise):
        B, L = means.shape
        m, v = means.detach().contiguous(), log_var.detach().contiguous()
        z = torch.empty_like(m)
        ops.B.reparam_fwd(m, v, eps_noise, z, None, B, L, L)
        ctx.save_for_backward(m, v, eps_noise)
        return z

    @staticmethod
    def backward(ctx, dz):
        m, v, eps_noise = ctx.saved_tensors
        B, L = m.shape
        dm, dv = torch.empty_like(m), torch.empty_like(v)
        ops.B.reparam_bwd(m, v, eps_noise, dz.contiguous(), 0.0, dm, dv, B, L, L)
        return dm, dv, None


class KLFn(torch.autograd.Function):
    """-0.5 * sum(1 + log_var - means^2 - exp(log_var))  (problems.py:406, 429) -> 0-dim fp32 tensor."""

    @staticmethod
    def forward(ctx, means, log_var):
        B, L = means.shape
        m, v = means.detach().contiguous(), log_var.detach().contiguous()
        acc = torch.zeros(1, dtype=torch.float64, device=m.device)
        ops.B.reparam_fwd(m, v, None, None, acc, B, L, L)
        ctx.save_for_backward(m, v)
        return acc[0].to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        m, v = ctx.saved_tensors
        B, L = m.shape
        dm, dv = torch.empty_like(m), torch.empty_like(v)
        ops.B.reparam_bwd(m, v, None, None, 1.0, dm, dv, B, L, L)
        return _scaled(dm, g), _scaled(dv, g)


class BCEWithLogitsSumFn(torch.autograd.Function):
    """F.binary_cross_entropy_with_logits(x, t, reduction='sum') with the optional broadcast loss mask of
    problems.py:445-447; the gradient is produced in the same pass."""

    @staticmethod
    def forward(ctx, logits, target, mask):
        x, t = logits.detach().contiguous(), target.detach().contiguous()
        n = x.numel()
        hw = x.shape[-1] * x.shape[-2]
        chw = n // x.shape[0]
        acc = torch.zeros(1, dtype=torch.float64, device=x.device)
        d = torch.empty_like(x)
        mk, mc = None, 1
        if mask is not None:
            # torch.mul(recon_i, loss_mask) broadcasts: the synthetic mask is [B,1,H,W], the dataset's segmentation
            # mask [B,C,H,W] (datasets.py: 3-channel PNG); anything else is expanded to the logits' shape first
            mk = mask.detach().to(x.dtype)
            if mk.dim() != 4 or mk.shape[0] != x.shape[0] or mk.shape[2:] != x.shape[2:] or mk.shape[1] not in (1, x.shape[1]):
                mk = mk.expand_as(x)
            mk = mk.contiguous()
            mc = mk.shape[1]
        ops.B.bce_logits(x, t, mk, d, acc, n, chw, hw, 1.0, mc)
        ctx.save_for_backward(d)
        return acc[0].to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return _scaled(d, g), None, None


class MSESumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, r, t):
        rr, tt = r.detach().contiguous(), t.detach().contiguous()
        acc = torch.zeros(1, dtype=torch.float64, device=rr.device)
        d = torch.empty_like(rr)
        ops.B.mse(rr, tt, d, acc, rr.numel(), 1.0)
        ctx.save_for_backward(d)
        return acc[0].to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return _scaled(d, g), None


class SwishFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        xc = x.detach().contiguous()
        out = torch.empty_like(xc)
        ops.B.act_fwd(xc, out, ops.ACT_SWISH)
        ctx.save_for_backward(xc)
        return out

    @staticmethod
    def backward(ctx, g):
        (xc,) = ctx.saved_tensors
        d = torch.empty_like(xc)
        ops.B.act_bwd(g.contiguous(), xc, d, ops.ACT_SWISH)
        return d
