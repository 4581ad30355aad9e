"""TEST INFRASTRUCTURE ONLY: a CPU emulation of the C ABI *contracts* in include/mmdyn_hip.h.

It lets the `-m "not gpu"` suite exercise the product's host-side schedule (mmdyn_hip/layers.py,
models/, problems/: packing, tap geometry, NHWC bookkeeping, backward ordering) on a machine without a
GPU, by swapping this object in for mmdyn_hip.ops.B.  Each method restates what the corresponding kernel
is documented to compute, with plain torch indexing -- it shares no code with the kernels and is never
imported by the product.  GPU tests (-m gpu) run the real library and compare against the oracle.
"""
import math

import numpy as np

import torch
import torch.nn.functional as F

from mmdyn_hip import _lib

DENSE, CONV, TCONV, IM2COL3, TCONV_S1P0 = 0, 1, 2, 3, 4


def _act(x, act):
    if act == 1:
        return x * torch.sigmoid(x)
    if act == 2:
        return torch.relu(x)
    return x


def _act_grad(u, act):
    if act == 1:
        s = torch.sigmoid(u)
        return s * (1 + u * (1 - s))
    if act == 2:
        return (u > 0).to(u.dtype)
    return torch.ones_like(u)


class EmuBackend:
    name = "emu"

    # ops that may see bf16 ACTIVATION tensors (bf16-storage mode): the emulation computes in fp32 on widened copies
    # and rounds what it wrote back into the bf16 tensors (RNE), which is the kernels' contract
    BF16_OPS = ("igemm_nt", "igemm_nt_dgrad_bn", "wgrad_tn", "bn_swish_fwd", "bn_swish_bwd_reduce", "bn_swish_bwd_apply",
                "act_bwd", "tconv_out3_fwd", "pack_conv_weight", "repack2d", "repack2d_ld")

    def __init__(self):
        self.lib = _lib.load()   # host-only helpers (tile counts) come from the real library
        self.calls = []
        self.precision = "fp32"
        for name in self.BF16_OPS:
            setattr(self, name, self._with_bf16(getattr(self, name)))

    @staticmethod
    def _with_bf16(fn):
        def wrapped(*args):
            h16 = (torch.bfloat16, torch.float16)           # ("bf16s" / "fp16s": the storage type of the mode)
            conv = [a.float() if torch.is_tensor(a) and a.dtype in h16 else a for a in args]
            r = fn(*conv)
            for o, c in zip(args, conv):
                if torch.is_tensor(o) and o.dtype in h16:
                    o.copy_(c)
            return r
        return wrapped

    # host helpers
    def igemm_stat_tiles(self, *a, all16=False):
        return self.lib.mmdyn_igemm_stat_tiles(*a)

    def colstats_tiles(self, r):
        return self.lib.mmdyn_colstats_tiles(r)

    def wgrad_chunks(self, mode, rows, Cd, Cg):
        return self.lib.mmdyn_wgrad_chunks_mx(mode, rows, Cd, Cg, 7 if self.precision in ("bf16s", "fp16s") else 0)

    # ---- GEMMs ----
    @staticmethod
    def _gather(A, Bt, Hi, Wi, C, Hr, Wr, stride, offset, kh, kw):
        """rows (b, r, c) -> A[b, r*stride+offset+kh, c*stride+offset+kw, :] (zero outside)."""
        X = A.reshape(Bt, Hi, Wi, C)
        pad = 8
        Xp = F.pad(X, (0, 0, pad, pad, pad, pad))
        ys = torch.arange(Hr) * stride + offset + kh + pad
        xs = torch.arange(Wr) * stride + offset + kw + pad
        return Xp[:, ys][:, :, xs].reshape(Bt * Hr * Wr, C)

    @staticmethod
    def _unfold3(x, Bt, H, W):
        cols = F.unfold(x.reshape(Bt, 3, H, W), kernel_size=4, stride=2, padding=1)      # [Bt][48][Ho*Wo]
        out = torch.zeros(Bt * (H // 2) * (W // 2), 64)
        out[:, :48] = cols.permute(0, 2, 1).reshape(-1, 48)
        return out

    def igemm_nt(self, A, Bp, bias, C, C_act, stats, ws, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, ldc, stride, offset,
                 act, splitk):
        self.calls.append("igemm_nt")
        assert Cin % 32 == 0 and N % 32 == 0 and ldc == N
        Bt = G * Bg
        Bp = Bp.reshape(-1, N, Cin)
        # operands rounded to bf16 (RNE), fp32 accumulate -- except on the HBM-bound 3-channel layers, whose kernels
        # (conv3.hip) keep fp32 matrix cores in every precision mode
        if getattr(self, "precision", "fp32") != "fp32" and mode != IM2COL3:
            h = torch.float16 if self.precision in ("fp16", "fp16s") else torch.bfloat16
            A, Bp = A.to(h).to(torch.float32), Bp.to(h).to(torch.float32)
        if mode == IM2COL3:
            assert Cin == 64
            A = self._unfold3(A, Bt, Hi, Wi)
            mode, Hi, Wi = DENSE, Ho, Wo
        A = A.reshape(-1)[: Bt * Hi * Wi * Cin]
        if mode == DENSE:
            out = A.reshape(Bt * Hi * Wi, Cin) @ Bp[0].t()
            out = out.reshape(Bt, Ho, Wo, N)
        elif mode == CONV:
            out = torch.zeros(Bt * Ho * Wo, N)
            for t in range(16):
                out += self._gather(A, Bt, Hi, Wi, Cin, Ho, Wo, stride, offset, t >> 2, t & 3) @ Bp[t].t()
            out = out.reshape(Bt, Ho, Wo, N)
        elif mode == TCONV_S1P0:
            assert Ho == Hi + 3
            out = torch.zeros(Bt * Ho * Wo, N)
            for t in range(16):
                out += self._gather(A, Bt, Hi, Wi, Cin, Ho, Wo, 1, 0, -(t >> 2), -(t & 3)) @ Bp[t].t()
            out = out.reshape(Bt, Ho, Wo, N)
        else:
            assert Ho == 2 * Hi and Wo == 2 * Wi
            out = torch.zeros(Bt, Ho, Wo, N)
            for ph in range(2):
                for pw in range(2):
                    acc = torch.zeros(Bt * Hi * Wi, N)
                    for th in range(2):
                        for tw in range(2):
                            kh, kw = 1 - ph + 2 * th, 1 - pw + 2 * tw
                            g = self._gather(A, Bt, Hi, Wi, Cin, Hi, Wi, 1, 0, ph - th, pw - tw)
                            acc += g @ Bp[kh * 4 + kw].t()
                    out[:, ph::2, pw::2] = acc.reshape(Bt, Hi, Wi, N)
        out = out.reshape(-1, N)
        if stats is not None:
            stats.zero_()
            rpg = out.shape[0] // G
            for g in range(G):
                blk = out[g * rpg:(g + 1) * rpg]
                stats[g, 0, 0] = blk.sum(0)
                stats[g, 0, 1] = (blk * blk).sum(0)
        if splitk > 1:
            ws.zero_()
            ws[0].reshape(-1, N).copy_(out)
            return
        if bias is not None:
            out = out + bias
        C.reshape(-1, N).copy_(out)
        if C_act is not None:
            C_act.reshape(-1, N).copy_(_act(out, act))

    def igemm_nt_grouped(self, A, Bp, bias, C, C_act, u, G, rows, K, N, act):
        """mmdyn_igemm_nt_grouped: G dense GEMMs of one shape, group g on its own weights / bias."""
        self.calls.append("igemm_nt_grouped")
        h16 = (torch.bfloat16, torch.float16)
        Af, Bf = A.float().reshape(G, rows, K), Bp.float().reshape(G, N, K)
        if getattr(self, "precision", "fp32") != "fp32":
            h = torch.float16 if self.precision in ("fp16", "fp16s") else torch.bfloat16
            Af, Bf = Af.to(h).to(torch.float32), Bf.to(h).to(torch.float32)
        out = torch.einsum("grk,gnk->grn", Af, Bf)
        if u is not None:
            uf = u.float().reshape(G, rows, N)
            out = out * (_act_grad(uf, act))
            C.copy_(out.reshape(C.shape).to(C.dtype))
            return
        if bias is not None:
            out = out + bias.reshape(G, 1, N)
        C.copy_(out.reshape(C.shape).to(C.dtype))
        if C_act is not None:
            C_act.copy_(_act(out, act).reshape(C_act.shape).to(C_act.dtype))

    def wgrad_tn_grouped(self, D, Gt, partial, G, rows, Cd, Cg, chunks):
        """partial [chunks][G][Cd][Cg]; one wgrad_reduce over Cd' = G*Cd then sums the slabs of all groups."""
        Df, Gf = D.float().reshape(G, rows, Cd), Gt.float().reshape(G, rows, Cg)
        if getattr(self, "precision", "fp32") != "fp32":
            h = torch.float16 if self.precision in ("fp16", "fp16s") else torch.bfloat16
            Df, Gf = Df.to(h).to(torch.float32), Gf.to(h).to(torch.float32)
        full = torch.einsum("grd,grc->gdc", Df, Gf)
        partial.zero_()
        p = partial.reshape(chunks, G, Cd, Cg)
        p[0] = 0.5 * full
        p[chunks - 1] += 0.5 * full

    def splitk_reduce(self, ws, bias, C, C_act, splitk, rows, N, act):
        out = ws.reshape(splitk, rows, N).sum(0)
        if bias is not None:
            out = out + bias
        C.reshape(rows, N).copy_(out)
        if C_act is not None:
            C_act.reshape(rows, N).copy_(_act(out, act))

    def wgrad_tn(self, D, Gt, partial, mode, Bt, Hr, Wr, Cd, Hi, Wi, Cg, stride, offset, chunks):
        assert chunks % 4 == 0 and Cd % 32 == 0 and Cg % 32 == 0
        rows = Bt * Hr * Wr
        if getattr(self, "precision", "fp32") != "fp32" and mode != IM2COL3:
            h = torch.float16 if self.precision in ("fp16", "fp16s") else torch.bfloat16
            D, Gt = D.to(h).to(torch.float32), Gt.to(h).to(torch.float32)
        Dm = D.reshape(-1)[: rows * Cd].reshape(rows, Cd)
        partial.zero_()
        taps = 16 if mode == CONV else 1
        p = partial.reshape(chunks, taps, Cd, Cg)
        for t in range(taps):
            if mode == IM2COL3:
                g = self._unfold3(Gt, Bt, Hi, Wi)
            elif mode == CONV:
                g = self._gather(Gt.reshape(-1)[: Bt * Hi * Wi * Cg], Bt, Hi, Wi, Cg, Hr, Wr, stride, offset, t >> 2, t & 3)
            else:
                g = Gt.reshape(-1)[: rows * Cg].reshape(rows, Cg)
            full = Dm.t() @ g
            p[0, t] = 0.5 * full          # spread over two slabs: the reduce must sum them all
            p[chunks - 1, t] += 0.5 * full

    def wgrad_reduce(self, partial, canon, chunks, taps, Cd, Cg, cg_canon, perm, beta):
        s = partial.reshape(chunks, taps, Cd, Cg).sum(0)[:, :, :cg_canon]          # [taps][Cd][cgc]
        if perm == 0:
            out = s.permute(1, 2, 0).reshape(-1)
        elif perm == 1:
            out = s[0].reshape(Cd, 25, 256).permute(0, 2, 1).reshape(-1)           # [cd][ch][hw]
        else:
            out = s[0].reshape(25, 256, cg_canon).permute(1, 0, 2).reshape(-1)     # [ch][hw][cg]
        flat = canon.reshape(-1)
        assert flat.numel() == out.numel(), (flat.numel(), out.numel())
        flat.copy_(beta * flat + out if beta else out)

    # ---- packing ----
    def pack_conv_weight(self, Wc, P, d0, d1, swap):
        w = Wc.reshape(d0, d1, 16)
        P.reshape(-1).copy_((w.permute(2, 1, 0) if swap else w.permute(2, 0, 1)).reshape(-1))

    def repack2d(self, src, dst, rows_in, cols_in, rows_out, cols_out, mode):
        s = src.reshape(rows_in, cols_in)
        if mode == 0:
            m = s
        elif mode == 1:
            m = s.t()
        elif mode == 2:
            m = s.reshape(rows_in, 256, 25).permute(0, 2, 1).reshape(rows_in, cols_in)
        elif mode == 3:
            m = s.reshape(256, 25, cols_in).permute(1, 0, 2).reshape(rows_in, cols_in)
        elif mode == 4:
            m = s.reshape(rows_in, 256, 25).permute(2, 1, 0).reshape(cols_in, rows_in)
        else:
            m = s.reshape(256, 25, cols_in).permute(2, 1, 0).reshape(cols_in, rows_in)
        out = torch.zeros(rows_out, cols_out)
        r, c = min(m.shape[0], rows_out), min(m.shape[1], cols_out)      # zero-pads or crops, like the kernel
        out[:r, :c] = m[:r, :c]
        dst.reshape(-1).copy_(out.reshape(-1))

    def repack2d_ld(self, src, dst, rows_in, cols_in, rows_out, cols_out, ld_out, mode):
        tmp = torch.zeros(rows_out, cols_out)
        self.repack2d(src, tmp, rows_in, cols_in, rows_out, cols_out, mode)
        flat = dst.reshape(-1) if dst.is_contiguous() else None
        base = dst.storage_offset()
        buf = torch.as_strided(dst, (rows_out, cols_out), (ld_out, 1), base)
        buf.copy_(tmp)

    def pack_plan(self, table, n):
        raise NotImplementedError("the emulation runs pack specs one by one (layers.pack_now)")

    def im2col_nchw3(self, x, col, Bt, H, W):
        cols = F.unfold(x.reshape(Bt, 3, H, W), kernel_size=4, stride=2, padding=1)      # [Bt][48][Ho*Wo]
        out = torch.zeros(Bt * (H // 2) * (W // 2), 64)
        out[:, :48] = cols.permute(0, 2, 1).reshape(-1, 48)
        col.reshape(-1).copy_(out.reshape(-1))

    def col2im_k4(self, col, out, Bt, Hi, Wi, Ho, Wo, C, ldcol, stride, pad, tap_major):
        c = col.reshape(Bt, Hi * Wi, ldcol)[:, :, : 16 * C]
        if tap_major:
            c = c.reshape(Bt, Hi * Wi, 16, C).permute(0, 3, 2, 1)        # [Bt][C][16][L]
        else:
            c = c.reshape(Bt, Hi * Wi, C, 16).permute(0, 2, 3, 1)
        img = F.fold(c.reshape(Bt, C * 16, Hi * Wi), (Ho, Wo), kernel_size=4, stride=stride, padding=pad)
        if tap_major:
            img = img.permute(0, 2, 3, 1)
        out.reshape(-1).copy_(img.reshape(-1))

    def tconv_out3_fwd(self, a, w, out, Bt, Hi, Wi):
        x = a.reshape(Bt, Hi, Wi, 32).permute(0, 3, 1, 2)
        out.reshape(-1).copy_(F.conv_transpose2d(x, w.reshape(32, 3, 4, 4), stride=2, padding=1).reshape(-1))

    def tconv_out3_bn_fwd(self, y, mean, rstd, gamma, beta, w, out, G, Bg, Hi, Wi):
        """mmdyn_tconv_out3_bn_fwd: the activation is applied in fp32 on the (possibly 16-bit) y and never rounded to storage."""
        a = torch.empty(y.shape, dtype=torch.float32)
        EmuBackend.bn_swish_fwd(self, y.float(), mean, rstd, gamma, beta, a, G, Bg * Hi * Wi, 32)
        EmuBackend.tconv_out3_fwd(self, a, w, out, G * Bg, Hi, Wi)

    def tconv_out3_bn_bce(self, y, mean, rstd, gamma, beta, w, logits, logits_group, target, dlogit, loss_slots, slot_of_group,
                          grad_scale, G, Bg, Hi, Wi, mask=None, mask_channels=1, unmasked_slots=None):
        full = torch.empty(G * Bg, 3, 2 * Hi, 2 * Wi)
        EmuBackend.tconv_out3_bn_fwd(self, y, mean, rstd, gamma, beta, w, full, G, Bg, Hi, Wi)
        n = target.numel()
        EmuBackend.bce_logits_groups(self, full, target, dlogit, loss_slots, slot_of_group, n, grad_scale, mask=mask,
                                     chw=target[0].numel(), hw=target[0, 0].numel(), mask_channels=mask_channels,
                                     unmasked_slots=unmasked_slots)
        if logits is not None:
            src = full if logits_group < 0 else full[logits_group * Bg:(logits_group + 1) * Bg]
            logits.reshape(-1).copy_(src.reshape(-1))

    def wgrad_out3_bn(self, y, mean, rstd, gamma, beta, Gt, partial, G, Bg, Hr, chunks):
        a = torch.empty(y.shape, dtype=torch.float32)
        EmuBackend.bn_swish_fwd(self, y.float(), mean, rstd, gamma, beta, a, G, Bg * Hr * Hr, 32)
        prev, self.precision = self.precision, "fp32"            # (the 3-channel layers keep fp32 matrix cores in every mode)
        try:
            EmuBackend.wgrad_tn(self, a, Gt, partial, IM2COL3, G * Bg, Hr, Hr, 32, 2 * Hr, 2 * Hr, 64, 1, 0, chunks)
        finally:
            self.precision = prev

    def nchw_to_nhwc(self, src, dst, B, C, HW):
        dst.reshape(-1).copy_(src.reshape(B, C, HW).permute(0, 2, 1).reshape(-1))

    def nhwc_to_nchw(self, src, dst, B, C, HW):
        dst.reshape(-1).copy_(src.reshape(B, HW, C).permute(0, 2, 1).reshape(-1))

    # ---- BN ----
    def colstats(self, y, partial, G, rpg, C):
        partial.zero_()
        v = y.reshape(G, rpg, C)
        partial[:, 0, 0] = v.sum(1)
        partial[:, 0, 1] = (v * v).sum(1)

    def bn_finalize(self, partial, mean, rstd, rm, rv, nbt, scratch, G, T, C, rpg, eps, momentum, repeat):
        s = partial.reshape(G, T, 2, C).double().sum(1)
        m = s[:, 0] / rpg
        var = (s[:, 1] / rpg - m * m).clamp_min(0)
        mean.copy_(m.float())
        rstd.copy_((1 / torch.sqrt(var + eps)).float())
        for g in range(G):
            for _ in range(repeat):
                if rm is not None:
                    rm.mul_(1 - momentum).add_(momentum * m[g].float())
                if rv is not None:
                    rv.mul_(1 - momentum).add_(momentum * (var[g] * rpg / max(rpg - 1, 1)).float())
        if nbt is not None:
            nbt += G * repeat

    def bn_eval_stats(self, rm, rv, mean, rstd, G, C, eps):
        mean.copy_(rm.reshape(1, C).expand(G, C))
        rstd.copy_((1.0 / torch.sqrt(rv + eps)).reshape(1, C).expand(G, C))

    def bn_reduce_partials(self, partial, sums, scratch, G, T, C):
        sums.copy_(partial.reshape(G, T, 2, C).double().sum(1))

    def bn_finalize_sums(self, sums, mean, rstd, rm, rv, nbt, G, C, n, eps, momentum, repeat):
        self.bn_finalize(sums.reshape(G, 1, 2, C), mean, rstd, rm, rv, nbt, None, G, 1, C, n, eps, momentum, repeat)

    def bn_bwd_finalize_sums(self, sums, sums_f, dgamma, dbeta, G, C, sums_scale, beta_acc):
        s = sums.reshape(G, 2, C)
        if sums_f is not None:
            sums_f.copy_((s * sums_scale).float())
        if dbeta is not None:
            dbeta.copy_((beta_acc * dbeta if beta_acc else 0) + s[:, 0].sum(0).float())
        if dgamma is not None:
            dgamma.copy_((beta_acc * dgamma if beta_acc else 0) + s[:, 1].sum(0).float())

    @staticmethod
    def _xhat(y, mean, rstd, G, rpg, C):
        return (y.reshape(G, rpg, C) - mean[:, None]) * rstd[:, None]

    def bn_swish_fwd(self, y, mean, rstd, gamma, beta, a, G, rpg, C):
        u = self._xhat(y, mean, rstd, G, rpg, C) * gamma + beta
        a.reshape(-1).copy_(_act(u, 1).reshape(-1))

    def bn_swish_bwd_reduce(self, da, y, mean, rstd, gamma, beta, partial, G, rpg, C):
        xh = self._xhat(y, mean, rstd, G, rpg, C)
        du = da.reshape(G, rpg, C) * _act_grad(xh * gamma + beta, 1)
        partial.zero_()
        partial[:, 0, 0] = du.sum(1)
        partial[:, 0, 1] = (du * xh).sum(1)

    def bn_bwd_finalize(self, partial, sums, dgamma, dbeta, scratch, G, T, C, beta_acc):
        s = partial.reshape(G, T, 2, C).sum(1)
        sums.copy_(s)
        if dbeta is not None:
            dbeta.copy_((beta_acc * dbeta if beta_acc else 0) + s[:, 0].sum(0))
        if dgamma is not None:
            dgamma.copy_((beta_acc * dgamma if beta_acc else 0) + s[:, 1].sum(0))

    def igemm_nt_dgrad_bn(self, A, Bp, C, stats, y, mean, rstd, gamma, beta, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N,
                          stride, offset):
        self.igemm_nt(A, Bp, None, C, None, None, None, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, N, stride, offset, 0, 1)
        rpg = Bg * Ho * Wo
        xh = self._xhat(y, mean, rstd, G, rpg, N)
        du = C.reshape(G, rpg, N) * _act_grad(xh * gamma + beta, 1)
        C.reshape(-1).copy_(du.reshape(-1))
        stats.zero_()
        st = stats.reshape(G, -1, 2, N)
        st[:, 0, 0] = du.sum(1)
        st[:, 0, 1] = (du * xh).sum(1)

    def bn_swish_bwd_apply(self, da, y, mean, rstd, gamma, beta, sums, dy, G, rpg, C, da_is_du=False):
        xh = self._xhat(y, mean, rstd, G, rpg, C)
        du = da.reshape(G, rpg, C) if da_is_du else da.reshape(G, rpg, C) * _act_grad(xh * gamma + beta, 1)
        out = gamma * rstd[:, None] * (du - sums[:, 0][:, None] / rpg - xh * sums[:, 1][:, None] / rpg)
        dy.reshape(-1).copy_(out.reshape(-1))

    # ---- element-wise ----
    def act_fwd(self, u, h, act):
        h.reshape(-1).copy_(_act(u.reshape(-1), act))

    def act_bwd(self, dh, u, du, act):
        du.reshape(-1).copy_(dh.reshape(-1) * _act_grad(u.reshape(-1), act))

    def dropout_expand(self, h, masks, out, P, B, H, p):
        out.reshape(P, B, H).copy_(h.reshape(1, B, H) * (masks.reshape(P, B, H).float() / (1 - p)))

    def dropout_reduce(self, dout, masks, dh, P, B, H, p, u=None, act=0):
        s = (dout.reshape(P, B, H) * (masks.reshape(P, B, H).float() / (1 - p))).sum(0)
        dh.reshape(B, H).copy_(s if u is None else s * _act_grad(u.reshape(B, H), act))

    def igemm_nt_dgrad_act(self, A, Bp, C, u, act, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, stride, offset):
        tmp = torch.empty(C.shape, dtype=torch.float32)
        self.igemm_nt(A, Bp, None, tmp, None, None, None, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, N, stride, offset, 0, 1)
        C.reshape(-1).copy_((tmp.reshape(-1) * _act_grad(u.reshape(-1).float(), act)).to(C.dtype))

    def counter_add(self, counter, inc):
        counter += inc

    def random_masks(self, masks, p, seed, offset, offset_dev=None):
        offset += 0 if offset_dev is None else int(offset_dev)
        g = torch.Generator().manual_seed(seed + offset)
        masks.copy_((torch.rand(masks.shape, generator=g) >= p).to(torch.uint8))

    def random_normal(self, out, seed, offset, offset_dev=None):
        offset += 0 if offset_dev is None else int(offset_dev)
        g = torch.Generator().manual_seed(seed + offset)
        out.copy_(torch.randn(out.shape, generator=g))

    def colsum(self, x, out, rows, C, perm, beta):
        s = x.reshape(rows, C).sum(0)
        if perm == 2:
            s = s.reshape(25, 256).t().reshape(-1)
        out.reshape(-1).copy_(beta * out.reshape(-1) + s if beta else s)

    def scale_dev(self, x, s, out):
        out.reshape(-1).copy_(x.reshape(-1) * s.reshape(-1)[0])

    def sum_blocks(self, x, out, P, n):
        out.reshape(-1).copy_(x.reshape(P, n).sum(0))

    def copy_many(self, pairs):
        for d, s_ in pairs:
            if d.data_ptr() != s_.data_ptr():
                d.copy_(s_)

    def cast_f32_to_bf16(self, src, dst):
        dst.copy_(src.to(torch.bfloat16))

    def cast_bf16_to_f32(self, src, dst):
        dst.copy_(src.to(torch.float32))

    def linear_small_fwd(self, x, W, b, y, rows, K, N, act):
        o = x.reshape(rows, K) @ W.reshape(N, K).t()
        if b is not None:
            o = o + b
        y.reshape(rows, N).copy_(_act(o, act))

    def linear_small_bwd(self, dy, x, W, dx, dW, db, rows, K, N, beta):
        dy2, x2 = dy.reshape(rows, N), x.reshape(rows, K)
        if dx is not None:
            dx.reshape(rows, K).copy_(dy2 @ W.reshape(N, K))
        dW.reshape(N, K).copy_(dy2.t() @ x2)
        if db is not None:
            db.copy_(dy2.sum(0))

    # ---- latent / loss ----
    @staticmethod
    def _experts(p, B, L):
        mus, lvs = [], []
        for m in range(len(p["ld"])):
            t = p["mu"][m]
            if t is not None:
                mus.append(t[:, :L] if t.dim() == 2 else t)
                lvs.append(p["lv"][m][:, :L])
        return mus, lvs

    def _poe_math(self, mus, lvs, with_prior, B, L):
        eps = 1e-8
        sumT = torch.full((B, L), 1.0 / (1.0 + 2 * eps) if with_prior else 0.0)
        if with_prior:
            sumT = torch.ones(B, L) / ((torch.ones(B, L) + eps) + eps)
        sumMuT = torch.zeros(B, L)
        for mu, lv in zip(mus, lvs):
            T = 1.0 / ((torch.exp(lv) + eps) + eps)
            sumT = sumT + T
            sumMuT = sumMuT + mu * T
        pd_mu = sumMuT / sumT
        pd_lv = torch.log(1.0 / sumT + eps)
        return pd_mu, pd_lv

    def poe_fwd(self, passes, eps_noise, mu, logvar, z, kl_sum, with_prior, P, B, L):
        for i, p in enumerate(passes):
            mus, lvs = self._experts(p, B, L)
            pm, plv = self._poe_math(mus, lvs, with_prior, B, L)
            mu.reshape(P, B, L)[i] = pm
            logvar.reshape(P, B, L)[i] = plv
            if z is not None:
                z.reshape(P, B, L)[i] = eps_noise.reshape(P, B, L)[i] * torch.exp(0.5 * plv) + pm
                for t in p.get("zdst", []):
                    if t is not None:
                        t.reshape(B, L).copy_(z.reshape(P, B, L)[i])
            if kl_sum is not None:
                kl_sum[i] += (-0.5 * (1 + plv - pm * pm - plv.exp()).double().sum())

    def poe_bwd(self, passes, eps_noise, mu, logvar, dz, g_mu, g_lv, kl_scale, with_prior, P, B, L, kl_weight_dev=None):
        if kl_weight_dev is not None:
            kl_scale = kl_scale * float(kl_weight_dev[0])
        with torch.enable_grad():
            self._poe_bwd(passes, eps_noise, mu, logvar, dz, g_mu, g_lv, kl_scale, with_prior, P, B, L)

    def _poe_bwd(self, passes, eps_noise, mu, logvar, dz, g_mu, g_lv, kl_scale, with_prior, P, B, L):
        for i, p in enumerate(passes):
            idx = [m for m in range(len(p["ld"])) if p["mu"][m] is not None]
            mus = [p["mu"][m][:, :L].detach().clone().requires_grad_(True) for m in idx]
            lvs = [p["lv"][m][:, :L].detach().clone().requires_grad_(True) for m in idx]
            pm, plv = self._poe_math(mus, lvs, with_prior, B, L)
            obj = kl_scale * (-0.5 * (1 + plv - pm * pm - plv.exp()).sum())
            gz = torch.zeros(B, L)
            has = False
            if dz is not None:
                gz, has = gz + dz.reshape(P, B, L)[i], True
            for t in p.get("dz", []):
                if t is not None:
                    gz, has = gz + t.reshape(B, L), True
            if has:
                zz = eps_noise.reshape(P, B, L)[i] * torch.exp(0.5 * plv) + pm
                obj = obj + (zz * gz).sum()
            if g_mu is not None:
                obj = obj + (pm * g_mu.reshape(P, B, L)[i]).sum()
            if g_lv is not None:
                obj = obj + (plv * g_lv.reshape(P, B, L)[i]).sum()
            grads = torch.autograd.grad(obj, mus + lvs)
            for k, m in enumerate(idx):
                p["dmu"][m][:, :L] = grads[k]
                p["dlv"][m][:, :L] = grads[len(idx) + k]

    def reparam_fwd(self, mu, lv, eps_noise, z, kl_sum, B, L, ld):
        if z is not None:
            z.reshape(B, L).copy_(eps_noise.reshape(B, L) * torch.exp(0.5 * lv[:, :L]) + mu[:, :L])
        if kl_sum is not None:
            kl_sum += (-0.5 * (1 + lv[:, :L] - mu[:, :L] ** 2 - lv[:, :L].exp()).double().sum())

    def reparam_bwd(self, mu, lv, eps_noise, dz, kl_scale, dmu, dlv, B, L, ld):
        m, v = mu[:, :L], lv[:, :L]
        gm = kl_scale * m
        gv = -0.5 * kl_scale * (1 - v.exp())
        if dz is not None:
            gm = gm + dz.reshape(B, L)
            gv = gv + dz.reshape(B, L) * eps_noise.reshape(B, L) * 0.5 * torch.exp(0.5 * v)
        dmu[:, :L] = gm
        dlv[:, :L] = gv

    def bce_logits(self, logits, target, mask, dlogit, loss_sum, n, chw, hw, grad_scale, mask_channels=1):
        x, t = logits.reshape(-1)[:n], target.reshape(-1)[:n]
        if mask is not None:
            c = chw // hw
            assert mask_channels in (1, c) and mask.numel() == (n // chw) * mask_channels * hw
            mk = mask.reshape(-1, mask_channels, hw).expand(-1, c, hw).reshape(-1)
            x, t = x * mk, t * mk
        loss_sum += F.binary_cross_entropy_with_logits(x, t, reduction="sum").double()
        if dlogit is not None:
            d = (torch.sigmoid(x) - t) * grad_scale
            if mask is not None:
                d = d * mk
            dlogit.reshape(-1)[:n] = d

    def bce_logits_groups(self, logits, target, dlogit, loss_slots, slot_of_group, n, grad_scale, mask=None, chw=0, hw=0,
                          mask_channels=1, unmasked_slots=None):
        lg = logits.reshape(len(slot_of_group), -1)
        dl = None if dlogit is None else dlogit.reshape(len(slot_of_group), -1)
        for g, slot in enumerate(slot_of_group):
            if slot < 0:
                if dl is not None:
                    dl[g].zero_()
                continue
            self.bce_logits(lg[g], target, mask, None if dl is None else dl[g], loss_slots[slot:slot + 1], n, chw, hw,
                            grad_scale, mask_channels)
            if mask is not None and unmasked_slots is not None:
                self.bce_logits(lg[g], target, None, None, unmasked_slots[slot:slot + 1], n, 0, 0, grad_scale)

    def mse(self, r, t, dr, loss_sum, n, grad_scale):
        d = r.reshape(-1)[:n] - t.reshape(-1)[:n]
        loss_sum += (d * d).double().sum()
        if dr is not None:
            dr.reshape(-1)[:n] = 2 * d * grad_scale

    def mse_groups(self, r, t, dr, loss_slots, slot_of_group, n, grad_scale):
        for g, slot in enumerate(slot_of_group):
            self.mse(r.reshape(-1)[g * n:(g + 1) * n], t, None if dr is None else dr.reshape(-1)[g * n:(g + 1) * n],
                     loss_slots[slot:slot + 1], n, grad_scale)

    def elbo_assemble(self, bce, mse, kl, loss, partials, P, B, kl_weight, pose_multiplier, kl_weight_dev=None):
        if kl_weight_dev is not None:
            kl_weight = kl_weight * float(kl_weight_dev[0])
        tot = 0.0
        for p in range(P):
            v = ((float(bce[p]) if bce is not None else 0.0) + pose_multiplier * (float(mse[p]) if mse is not None else 0.0)
                 + kl_weight * (float(kl[p]) if kl is not None else 0.0)) / B
            if partials is not None:
                partials[p] = v
            tot += v
        loss[0] = tot

    def adam_step(self, p, g, m, v, state, lr, beta1, beta2, eps, grad_scale, guarded=False):
        if guarded and not bool(torch.isfinite(g).all()):
            state[4] += 1
            state[5] = 1
            return
        if guarded:
            state[5] = 0
        state[0] += 1
        t = float(state[0])
        state[1] = lr / (1 - beta1 ** t)
        state[2] = math.sqrt(1 - beta2 ** t)
        gg = g * grad_scale
        m.add_((gg - m) * (1 - beta1))
        v.mul_(beta2).add_((1 - beta2) * gg * gg)
        p.sub_(float(state[1]) * (m / (v.sqrt() / float(state[2]) + eps)))

    def sgd_step(self, p, g, buf, lr, momentum, weight_decay, grad_scale, first):
        d = g * grad_scale + weight_decay * p
        buf.copy_(d if first else momentum * buf + d)
        p.sub_(lr * buf)

    # ---- image decode ----
    def resize_plan(self, in_size, out_size, device):
        ks = self.lib.mmdyn_resize_ksize(in_size, out_size)
        bounds = torch.empty(out_size, 2, dtype=torch.int32)
        coeffs = torch.empty(out_size, ks, dtype=torch.int32)
        assert self.lib.mmdyn_resize_plan(in_size, out_size, bounds.data_ptr(), coeffs.data_ptr()) == ks
        return bounds, coeffs

    def resize_u8_to_chw_f32(self, src, index, dst, n_out, Hin, Win, Hout, Wout, xb, xk, yb, yk):
        from oracle import resize_oracle as RO
        frames = src.reshape(-1, Hin, Win, 3).numpy()
        out = dst.reshape(n_out, 3, Hout, Wout)
        for b in range(n_out):
            img = frames[int(index[b]) if index is not None else b]
            r = RO.resize_bilinear_u8(img, Hout, Wout)
            out[b] = torch.from_numpy(np.ascontiguousarray(r.transpose(2, 0, 1)).astype(np.float32) / np.float32(255.0))

