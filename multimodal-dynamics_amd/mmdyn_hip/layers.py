"""Layer stacks of the cnn VAE / MVAE as explicit forward / backward kernel sequences.

This is the host-side schedule of the hot path: which ``mmdyn_*`` kernel runs on which buffer.  The
arithmetic itself is entirely in libmmdyn_hip.so (see :mod:`mmdyn_hip.ops`); torch supplies device
memory (``torch.empty``) only.

Layout: activations between layers are channels-last matrices ``[rows][C]`` with rows ordered
(sample, y, x).  A batch may consist of ``G`` groups of ``Bg`` samples (one group per modality-subset
pass of the reference's ``_evaluate_mvae``); train-mode BatchNorm statistics are per group, which makes
the grouped batch numerically identical to ``G`` separate reference forward calls.

Reference structure restated here (file:line relative to /root/reference/mmdyn/pytorch/models/vae.py):
  image encoder  :197-216, 224-242      image decoder  :263-279, 285-301
  pose MLPs      :117-123, 14-19        the flatten order c*25+hw of :227/:295 is absorbed into the
  packed FC weights (hw*256+c), so no activation is ever transposed.
"""
import weakref

import torch

from . import ops
from .ops import ACT_NONE, ACT_SWISH, ACT_RELU, DENSE, CONV, TCONV_S2P1, IM2COL3, TCONV_S1P0
from .models.shapes import BN_EPS, BN_MOMENTUM, FEAT


def _new(like, *shape, dtype=torch.float32):
    if isinstance(like, ops.Planes):
        like = like.t
    return torch.empty(shape, device=like.device, dtype=dtype)


# Storage type of the activations BETWEEN the convolution layers (pre-BatchNorm outputs, post-Swish activations and
# their gradients).  torch.float32 everywhere except in the bf16-storage mode of the engine (precision="bf16s",
# BASELINE configs[2]), which sets torch.bfloat16 for the duration of its calls.  FC-level tensors, logits, statistics
# and weights are always fp32.
ACT_DTYPE = torch.float32
# Storage type of the PACKED GEMM operands (the [tap][n][k] weight copies the pack kernels write every step): bf16 in
# both bf16 precision modes -- the matrix cores round them to bf16 anyway, and packing them so halves the weight bytes
# each implicit-GEMM block pulls through L2.  Master weights, biases and gradients stay fp32.
W_DTYPE = torch.float32


def _act(like, *shape):
    return torch.empty(shape, device=like.device, dtype=ACT_DTYPE)


def _cdiv(a, b):
    return (a + b - 1) // b


# ------------------------------------------------------------------------------------------------
# primitive helpers
# ------------------------------------------------------------------------------------------------
# fp32x3, round 6: the decoder's Linear forward (z and its activated output as planes) and the encoder FC layer's input gradient on
# the DENSE mode of the plane-ring kernel.  Built and tested (tests/test_kernels_aten_gpu.py::test_planes_dense_*); alone on the chip
# the two launches run 45.5 / 27.5 us against 51.5 / 26.5 us for the fp32-operand kernels and two stand-alone split launches
# disappear, but the two-lane step measured 0.6 % SLOWER with them, same box, alternating runs, twice (53.10 against 53.43 k, 52.9
# against 53.3 k samples/s: docs/LAB_NOTES.md H.b) -- a one-block-per-CU persistent kernel keeps the other lane's kernels off its
# CUs where the 64 KB blocks of the register-staged kernel shared them.  Off by default; bench.py --fc-planes turns it on.
FC_PLANES = False


def dense_planes_served(rows, K, N):
    """fp32x3: does the plane-ring kernel take this Linear-level launch on operands that arrive split (host-side query)?"""
    return FC_PLANES and planes_served(DENSE, 1, rows, 1, K, 1, N)


def dense(A, Bp, bias, rows, K, N, act=ACT_NONE, want_act=False, out_dtype=torch.float32, act_dtype=None, out=None,
          A_planes=None, act_planes=0):
    """C = A[rows][K] . Bp[N][K]^T (+bias) on the MFMA GEMM; picks split-K for short, wide-K problems.
    Returns (pre_activation, activated or None).  ``act_dtype``: storage type of the activated output alone (bf16 when it
    feeds a convolution of the bf16-storage mode; only without split-K).  ``out``: write the pre-activation there (a
    contiguous [rows][N] view, e.g. one group's rows of a grouped launch's operand) instead of a fresh tensor.
    ``A_planes`` (fp32x3, round 6): A as an ops.Planes written by its producer -- with the packed weight's plane twin the launch then
    runs on the persistent plane-ring kernel (stream-K over all CUs, split tiles finished inside the launch) where that serves the
    shape (dense_planes_served).  ``act_planes`` = c > 0: the activated output is returned as an ops.Planes of c-channel rows
    ([rows * N / c][c]: the operand of the plane launch that consumes it) instead of a tensor."""
    if A_planes is not None and out_dtype == torch.float32 and act_dtype is None and dense_planes_served(rows, K, N):
        wp = _plane_twin(Bp)
        if wp is not None and wp.C == K and wp.rows == N:
            C = _new(Bp, rows, N) if out is None else out
            Ca = None
            if want_act:
                Ca = ops.Planes(rows * N // act_planes, act_planes, Bp.device) if act_planes else _new(Bp, rows, N)
            ops.B.igemm_nt(A_planes, wp, bias, C, Ca, None, None, DENSE, 1, rows, 1, 1, K, 1, 1, N, N, 1, 0, act, 1)
            return C, Ca
    if act_planes:
        raise ValueError("mmdyn_hip: dense(act_planes=...) needs a launch the plane-ring kernel serves (ask dense_planes_served)")
    C = _new(A, rows, N, dtype=out_dtype) if out is None else out
    Ca = _new(A, rows, N, dtype=out_dtype if act_dtype is None else act_dtype) if want_act else None
    # split-K only when the 64x64 tiling leaves most of the 256 CUs idle AND K is long enough to amortise the
    # partial-sum pass and its second launch.  Re-measured with the wave-specialised kernels (tests/microbench/
    # splitk_probe.py, profiles/r3/splitk_probe.txt): K = 512 problems (heads, pose MLPs: 128 tiles) are faster unsplit
    # (15.7 vs 16.8 us incl. the reduce launch); 256 x 6400 -> 512 wants 8 slices (24 us; 16: 27), 1024 x 6400 -> 256 four.
    tiles = _cdiv(rows, 64) * _cdiv(N, 64)
    steps = K // 32
    splitk = max(1, min(256 // tiles, steps // 16)) if (N % 64 == 0 and out_dtype == torch.float32 and act_dtype is None) else 1
    if getattr(ops.B, "precision", "fp32") != "fp32" and (N % 64 == 0 and out_dtype == torch.float32 and act_dtype is None):
        # 16-bit matrix-core modes: a K-step is latency, not arithmetic (a 512-row K = 512 GEMM unsplit: 26 us at 64 blocks),
        # so the finer split of rounds 1-2 stays (without it bf16s bs 128 measured 2.45 vs 2.31 ms per step)
        splitk = max(1, min(512 // tiles, steps // 8))
    # (kept in the 16-bit matrix-core modes too: without it bf16s bs 128 measured 2.45 vs 2.31 ms per step)
    if splitk > 1:
        ws = _new(A, splitk, rows, N)
        ops.B.igemm_nt(A, Bp, None, C, None, None, ws, DENSE, 1, rows, 1, 1, K, 1, 1, N, N, 1, 0, ACT_NONE, splitk)
        ops.B.splitk_reduce(ws, bias, C, Ca, splitk, rows, N, act)
    else:
        ops.B.igemm_nt(A, Bp, bias, C, Ca, None, None, DENSE, 1, rows, 1, 1, K, 1, 1, N, N, 1, 0, act, 1)
    return C, Ca


# fp32x3: hand the plane-ring kernel its operands already split where it serves the launch (False: A/B measurements only --
# every launch then splits inside the kernel, the round-4 structure)
PLANES = True
# packed fp32 weight (by data pointer) -> (weak reference to that tensor, the Planes twin a PackPlan writes next to it every step).
# The plan's outputs never move; the weak reference guards against a dead plan's address being handed to another tensor.
PLANE_TWIN = {}


def _plane_twin(Wp):
    ent = PLANE_TWIN.get(Wp.data_ptr())
    if ent is None:
        return None
    if ent[0]() is not Wp:
        if ent[0]() is None:
            del PLANE_TWIN[Wp.data_ptr()]
        return None
    return ent[1]


PATCH_PLANES = True      # (False: the 32-channel up-sampling layers keep fp32 operands -- A/B measurements only)


def planes_served(mode, G, Bg, Hi, Cin, Ho, N):
    """True when the fp32x3 launch of this shape takes its operands already split (plane-ring kernel / the patch-resident kernel of
    the 32-channel up-sampling layers, host-side query)."""
    served = getattr(ops.B, "igemm_planes_served", None) if PLANES else None
    if N == 32 and not PATCH_PLANES:
        return False
    return bool(served is not None and ACT_DTYPE == torch.float32 and served(mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N))


def split_operands(x, Wp, mode, G, Bg, Hi, Cin, Ho, N):
    """fp32x3: where the plane-ring kernel serves the launch (the large N % 128 == 0 convolution-level GEMMs), hand it its two
    operands ALREADY SPLIT -- (Planes(x), Planes(Wp)) -- so that the GEMM contains no split; everything else gets (x, Wp) back.
    An activation that arrives as Planes was written so by its producer (bn_swish_*(planes=...)); the packed weight's twin comes
    from the pack plan (PLANE_TWIN); whatever is still fp32 is split here by its own launch (mmdyn_split_planes)."""
    if isinstance(Wp, ops.Planes):
        return x, Wp
    xin = x.t if isinstance(x, ops.Planes) else x
    if xin.dtype not in (torch.float32, torch.bfloat16) or Wp.dtype != torch.float32 or not planes_served(mode, G, Bg, Hi, Cin, Ho, N):
        if isinstance(x, ops.Planes):
            raise ValueError("mmdyn_hip: a plane operand for a launch the plane-ring kernel does not serve")
        return x, Wp
    if not isinstance(x, ops.Planes):
        xp = ops.Planes(x.numel() // Cin, Cin, x.device)
        ops.B.split_planes(x, xp)
        x = xp
    wp = _plane_twin(Wp)
    if wp is None or wp.C != Cin or wp.rows * Cin != Wp.numel():
        wp = ops.Planes(Wp.numel() // Cin, Cin, Wp.device)
        ops.B.split_planes(Wp, wp)
    return x, wp


def as_planes(x, C):
    """The fp32 activation x ([rows][C], or any contiguous tensor whose rows are C channels) as an ops.Planes: one mmdyn_split_planes
    launch (for tensors whose producer cannot write planes itself)."""
    xp = ops.Planes(x.numel() // C, C, x.device)
    ops.B.split_planes(x, xp)
    return xp


def conv_like(x, Wp, mode, G, Bg, Hi, Cin, Ho, N, stride=1, offset=0, stats=False, out_dtype=None):
    """Implicit-GEMM conv / transposed conv on NHWC rows; optional per-tile BatchNorm partial sums."""
    Bt = G * Bg
    if out_dtype in (None, torch.float32) and ACT_DTYPE == torch.float32:
        x, Wp = split_operands(x, Wp, mode, G, Bg, Hi, Cin, Ho, N)
    y = _new(x, Bt * Ho * Ho, N, dtype=ACT_DTYPE if out_dtype is None else out_dtype)
    st, T = None, 0
    if stats:
        if isinstance(x, ops.Planes):                # (operands that arrive split: the plane launch's tile count)
            T = ops.B.igemm_stat_tiles(mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, planes=True)
        else:
            T = ops.B.igemm_stat_tiles(mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, all16=ops._is16(x) and ops._is16(Wp))
        st = _new(x, G, T, 2, N)
    ops.B.igemm_nt(x, Wp, None, y, None, st, None, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, N, stride, offset,
                   ACT_NONE, 1)
    return y, st, T


def dgrad_act(x, Wp, mode, G, Bg, Hi, Cin, Ho, N, u, act, stride=1, offset=0):
    """Input-gradient GEMM with the backward of the activation whose pre-activation is ``u`` in its epilogue:
    returns dL/du = (x (*) Wp) * act'(u), stored like ``u``."""
    du = torch.empty_like(u)
    if u.dtype == torch.float32:
        x, Wp = split_operands(x, Wp, mode, G, Bg, Hi, Cin, Ho, N)
    ops.B.igemm_nt_dgrad_act(x, Wp, du, u, act, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, stride, offset)
    return du


def tconv_s1p0(x, Wsp, G, Bg, Cin, N, stats=False):
    """k4 s1 p0 transposed conv 5x5 -> 8x8 on NHWC rows, swap-packed weights [16][N][Cin].
    Large batches: the tap-skipping implicit GEMM (exactly the useful MACs, balanced pixel quads).  Small ones
    (too few blocks to fill 256 CUs): dense GEMM to a [rows][16*N] column matrix + col2im gather."""
    Bt = G * Bg
    # (from 256 blocks on: the per-GPU share of a 4-group decoder at bs 128 -- the column-matrix route costs 2.5x the
    # FLOPs plus a col2im pass: 0.55 vs ~0.15 ms there)
    if (G * 16 * _cdiv(Bg, 64) * (N // 64) >= 256 or ACT_DTYPE != torch.float32      # (bf16 storage: no column matrix)
            or isinstance(x, ops.Planes) or planes_served(TCONV_S1P0, G, Bg, 5, Cin, 8, N)):
        # (fp32x3: the plane-ring kernel's stream-K schedule serves one-group launches too -- 51 us against 141 for the
        #  register-staged quad walk and ~100 for the column-matrix route on the encoder's input gradient at bs 256)
        return conv_like(x, Wsp, TCONV_S1P0, G, Bg, 5, Cin, 8, N, stats=stats)
    col, _ = dense(x, Wsp, None, Bt * 25, Cin, 16 * N)
    y = _new(x, Bt * 64, N)
    ops.B.col2im_k4(col, y, Bt, 5, 5, 8, 8, N, 16 * N, 1, 0, True)
    st, T = None, 0
    if stats:
        T = ops.B.colstats_tiles(Bg * 64)
        st = _new(x, G, T, 2, N)
        ops.B.colstats(y, st, G, Bg * 64, N)
    return y, st, T


class BNState:
    """Handles of one BatchNorm2d: affine parameters + running buffers (may be None to skip updates).
    ``eval_stats``: (mean, rstd) [1][C] of the running estimates, computed ahead by a caller whose weights do not change between
    calls (engine.MVAEInference.refresh) -- eval-mode passes with one group then skip the per-call mmdyn_bn_eval_stats launch."""
    __slots__ = ("gamma", "beta", "rm", "rv", "nbt", "eval_stats")

    def __init__(self, gamma, beta, rm=None, rv=None, nbt=None, eval_stats=None):
        self.gamma, self.beta, self.rm, self.rv, self.nbt, self.eval_stats = gamma, beta, rm, rv, nbt, eval_stats


def _eval_stats(y, bn, G, C):
    """mean / rstd [G][C] of an eval-mode BatchNorm: the running estimates (nn.BatchNorm2d, training=False)."""
    if G == 1 and bn.eval_stats is not None:
        return bn.eval_stats
    mean, rstd = _new(y, G, C), _new(y, G, C)
    ops.B.bn_eval_stats(bn.rm, bn.rv, mean, rstd, G, C, BN_EPS)
    return mean, rstd


def precompute_eval_stats(buf, prev=None):
    """For every BatchNorm2d with running buffers in ``buf``: mean / rstd of the running estimates, stored in a copy of ``buf`` under
    "<layer>.eval_stats" (picked up by _bn_of).  Valid until the buffers change: the caller recomputes -- into the SAME tensors when it
    passes its previous result as ``prev`` (captured graphs keep reading them)."""
    out = dict(buf)
    for k in buf:
        if k.endswith(".running_mean"):
            pre = k[:-len(".running_mean")]
            rm, rv = buf[k], buf[pre + ".running_var"]
            C = rm.numel()
            old = (prev or {}).get(pre + ".eval_stats")
            if old is not None and old[0].numel() == C and old[0].device == rm.device:
                mean, rstd = old
            else:
                mean = torch.empty(1, C, device=rm.device, dtype=torch.float32)
                rstd = torch.empty(1, C, device=rm.device, dtype=torch.float32)
            ops.B.bn_eval_stats(rm, rv, mean, rstd, 1, C, BN_EPS)
            out[pre + ".eval_stats"] = (mean, rstd)
    return out


class SyncBN:
    """Synchronised BatchNorm over a data-parallel process group: set ``layers.SYNC = SyncBN(pg, world)`` and every
    BatchNorm of this module uses the statistics of the GLOBAL batch (one tiny fp64 all-reduce per layer and
    direction).  ``None`` (default) = local statistics, the semantics of DistributedDataParallel and of bench.py."""

    def __init__(self, group, world, lane_groups=None):
        """``lane_groups``: one process group per lane of the engine's schedule (same ranks as ``group``).  The visual
        and the tactile lane issue their statistics all-reduces from two streams -- and, in graph mode, from two
        concurrently replayed graphs -- whose relative order is not fixed across ranks; collectives of ONE communicator
        must be issued in the same order everywhere, so each lane gets a communicator of its own."""
        self.group, self.world = group, int(world)
        self.lane_groups = lane_groups
        self.pending = None      # list: Work handles of eagerly issued collectives are kept here (engine: the warm-up step
        #                          in front of a graph capture waits on every one of them before the capture starts)

    def all_reduce(self, t):
        import torch.distributed as dist
        g = self.group
        if self.lane_groups is not None and CUR_LANE is not None and CUR_LANE < len(self.lane_groups):
            g = self.lane_groups[CUR_LANE]
        if self.pending is None:
            dist.all_reduce(t, group=g)
        else:
            w = dist.all_reduce(t, group=g, async_op=True)
            w.wait()                                  # stream-ordered, like the blocking form
            self.pending.append(w)


SYNC = None
CUR_LANE = None          # index of the lane whose kernels are being enqueued (set by engine._Lanes / the graph capture)


def _bn_forward_stats(y, partial, T, bn, G, rows_per_group, C, repeat):
    mean, rstd = _new(y, G, C), _new(y, G, C)
    scratch = _new(y, 32, G, 2, C, dtype=torch.float64)
    if SYNC is None:
        ops.B.bn_finalize(partial, mean, rstd, bn.rm, bn.rv, bn.nbt, scratch, G, T, C, rows_per_group, BN_EPS,
                          BN_MOMENTUM, repeat)
    else:
        sums = _new(y, G, 2, C, dtype=torch.float64)
        ops.B.bn_reduce_partials(partial, sums, scratch, G, T, C)
        SYNC.all_reduce(sums)
        ops.B.bn_finalize_sums(sums, mean, rstd, bn.rm, bn.rv, bn.nbt, G, C, rows_per_group * SYNC.world, BN_EPS,
                               BN_MOMENTUM, repeat)
    return mean, rstd


def _bn_backward_sums(y, partial, T, dgamma, dbeta, G, C):
    """Parameter gradients (local) and the [G][2][C] sums the apply kernel needs (global under SyncBN)."""
    sums = _new(y, G, 2, C)
    scratch = _new(y, 32, G, 2, C, dtype=torch.float64)
    if SYNC is None:
        ops.B.bn_bwd_finalize(partial, sums, dgamma, dbeta, scratch, G, T, C, 0.0)
    else:
        s64 = _new(y, G, 2, C, dtype=torch.float64)
        ops.B.bn_reduce_partials(partial, s64, scratch, G, T, C)
        ops.B.bn_bwd_finalize_sums(s64, None, dgamma, dbeta, G, C, 1.0, 0.0)
        SYNC.all_reduce(s64)
        ops.B.bn_bwd_finalize_sums(s64, sums, None, None, G, C, 1.0 / SYNC.world, 0.0)
    return sums


def bn_swish_from_partials(y, partial, T, bn, G, rows_per_group, C, repeat=1, planes=False):
    """Returns (a, mean, rstd).  ``planes`` (fp32x3: the consumer GEMMs take their operand already split): the activated tensor is
    written ONLY as an ops.Planes -- the split rides on this pass, no fp32 copy exists -- and returned in place of ``a``."""
    if partial is None:             # eval mode: running estimates, no update (nn.BatchNorm2d, training=False)
        mean, rstd = _eval_stats(y, bn, G, C)
    else:
        mean, rstd = _bn_forward_stats(y, partial, T, bn, G, rows_per_group, C, repeat)
    if planes and y.dtype == torch.float32:
        ap = ops.Planes(y.shape[0], C, y.device)
        ops.B.bn_swish_fwd(y, mean, rstd, bn.gamma, bn.beta, None, G, rows_per_group, C, planes=ap)
        return ap, mean, rstd
    a = torch.empty_like(y)
    ops.B.bn_swish_fwd(y, mean, rstd, bn.gamma, bn.beta, a, G, rows_per_group, C)
    return a, mean, rstd


def _apply_out(y, planes_out):
    """Destination of a BatchNorm backward apply pass: (dy fp32 or None, dy as Planes or None) -- exactly one of the two."""
    if planes_out and y.dtype == torch.float32:
        return None, ops.Planes(y.shape[0], y.shape[1], y.device)
    return torch.empty_like(y), None


def bn_swish_backward(da, y, mean, rstd, bn, dgamma, dbeta, G, rows_per_group, C, planes_out=False):
    """Returns dL/dy -- as an ops.Planes (and only so) with ``planes_out``: the GEMMs that consume it take plane operands."""
    T = ops.B.colstats_tiles(rows_per_group)
    partial = _new(y, G, T, 2, C)
    ops.B.bn_swish_bwd_reduce(da, y, mean, rstd, bn.gamma, bn.beta, partial, G, rows_per_group, C)
    sums = _bn_backward_sums(y, partial, T, dgamma, dbeta, G, C)
    dy, dyp = _apply_out(y, planes_out)
    if dyp is not None:
        ops.B.bn_swish_bwd_apply(da, y, mean, rstd, bn.gamma, bn.beta, sums, None, G, rows_per_group, C, planes=dyp)
        return dyp
    ops.B.bn_swish_bwd_apply(da, y, mean, rstd, bn.gamma, bn.beta, sums, dy, G, rows_per_group, C)
    return dy


def dgrad_bn_swish_backward(x, Wp, mode, G, Bg, Hi, Cin, Ho, N, stride, offset, y, mean, rstd, bn, dgamma, dbeta, planes_out=False):
    """Input-gradient GEMM of the layer ABOVE fused with this layer's BatchNorm+Swish backward: the GEMM epilogue
    turns dL/da into du = dL/da * swish'(.) and emits the per-tile sums, so only finalize + apply remain.
    Returns dL/dy (gradient w.r.t. this layer's conv output) -- as an ops.Planes (and only so) with ``planes_out``: the apply pass
    writes the operand of the plane launches that consume it already split."""
    rows_per_group = Bg * Ho * Ho
    if y.dtype == torch.float32 and mode != IM2COL3:
        x, Wp = split_operands(x, Wp, mode, G, Bg, Hi, Cin, Ho, N)
    if isinstance(x, ops.Planes):
        T = ops.B.igemm_stat_tiles(mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, planes=True)
    else:
        T = ops.B.igemm_stat_tiles(mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, all16=ops._is16(x) and ops._is16(Wp))
    du = torch.empty_like(y)
    partial = _new(y, G, T, 2, N)
    ops.B.igemm_nt_dgrad_bn(x, Wp, du, partial, y, mean, rstd, bn.gamma, bn.beta, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N,
                            stride, offset)
    sums = _bn_backward_sums(y, partial, T, dgamma, dbeta, G, N)
    dy, dyp = _apply_out(y, planes_out)
    if dyp is not None:
        ops.B.bn_swish_bwd_apply(du, y, mean, rstd, bn.gamma, bn.beta, sums, None, G, rows_per_group, N, True, planes=dyp)
        return dyp
    ops.B.bn_swish_bwd_apply(du, y, mean, rstd, bn.gamma, bn.beta, sums, dy, G, rows_per_group, N, True)
    return dy


def wgrad(D, Gt, canon, mode, Bt, Hr, Cd, Hi, Cg, stride=1, offset=0, cg_canon=None, perm=0, defer=None):
    """canon[cd][cg][taps] = sum_rows D[row][cd] * G_tap[row][cg]  (reference weight-gradient layout).
    ``defer`` (a list): do not launch -- queue the call (nothing on the backward chain depends on a weight gradient); the
    engine runs the queue on another stream next to the following phase (run_deferred_wgrads)."""
    if defer is not None:
        defer.append((D, Gt, canon, mode, Bt, Hr, Cd, Hi, Cg, stride, offset, cg_canon, perm))
        return
    rows = Bt * Hr * Hr
    taps = 16 if mode == CONV else 1
    if isinstance(D, ops.Planes) or isinstance(Gt, ops.Planes):
        chunks = ops.B.wgrad_chunks(mode, rows, Cd, Cg, planes=(isinstance(D, ops.Planes), isinstance(Gt, ops.Planes)))
    else:
        chunks = ops.B.wgrad_chunks(mode, rows, Cd, Cg)
    partial = _new(D, chunks, taps, Cd, Cg)
    ops.B.wgrad_tn(D, Gt, partial, mode, Bt, Hr, Hr, Cd, Hi, Hi, Cg, stride, offset, chunks)
    ops.B.wgrad_reduce(partial, canon, chunks, taps, Cd, Cg, Cg if cg_canon is None else cg_canon, perm, 0.0)


def run_deferred_wgrads(queue):
    """Launch (and forget) the weight-gradient GEMMs queued with ``wgrad(..., defer=queue)`` on the current stream."""
    while queue:
        item = queue.pop(0)
        if item and callable(item[0]):          # (fn, *args): a weight gradient with its own launcher (wgrad_out3_bn)
            item[0](*item[1:])
        else:
            wgrad(*item)


def bn_stats_only(y, partial, T, bn, G, rows_per_group, C, repeat=1):
    """mean / rstd of a BatchNorm layer WITHOUT the apply pass (the consumer applies BatchNorm + Swish on its operand fetch)."""
    if partial is None:             # eval mode: running estimates
        return _eval_stats(y, bn, G, C)
    return _bn_forward_stats(y, partial, T, bn, G, rows_per_group, C, repeat)


def fuses_last_bn(S):
    """The decoder's last BatchNorm + Swish rides on its two consumers (tconv_out3.hip FUSED, conv3_wgrad_kernel BNACT) for the
    image sizes those direct kernels serve (local and synchronised statistics alike: the consumers only need mean / rstd)."""
    return S in (64, 128, 256)


def wgrad_out3_bn(y, mean, rstd, bn, dlogits, canon, G, Bg, Hr, defer=None):
    """Weight gradient of the last decoder layer (canonical [32][3][4][4]) from the pre-BatchNorm tensor y of the stage below:
    swish(BatchNorm(y)) is recomputed on the operand fetch (the activated tensor was never stored)."""
    if defer is not None:
        defer.append((wgrad_out3_bn, y, mean, rstd, bn, dlogits, canon, G, Bg, Hr))
        return
    rows = G * Bg * Hr * Hr
    chunks = ops.B.wgrad_chunks(IM2COL3, rows, 32, 64)
    partial = _new(y, chunks, 1, 32, 64)
    ops.B.wgrad_out3_bn(y, mean, rstd, bn.gamma, bn.beta, dlogits, partial, G, Bg, Hr, chunks)
    ops.B.wgrad_reduce(partial, canon, chunks, 1, 32, 64, 48, 0, 0.0)


def pack_conv(W, swap):
    d0, d1 = W.shape[0], W.shape[1]
    P = _new(W, 16, d1 if swap else d0, d0 if swap else d1)
    ops.B.pack_conv_weight(W, P, d0, d1, swap)
    return P


def repack(W, rows_in, cols_in, rows_out, cols_out, mode):
    out = _new(W, rows_out, cols_out)
    ops.B.repack2d(W, out, rows_in, cols_in, rows_out, cols_out, mode)
    return out


def _pad32(n):
    return (n + 31) // 32 * 32


def concat_condition(x, cond, width):
    """[x | cond | 0] as one [rows, width] matrix (width = K padded to the MFMA K-step): the reference's
    torch.cat((x, c.float()), dim=-1) of the conditional models (vae.py:231-237, 286-291)."""
    rows, K = x.shape
    cd = cond.shape[1]
    out = torch.zeros(rows, width, device=x.device, dtype=torch.float32)
    ops.B.repack2d_ld(x, out, rows, K, rows, K, width, 0)
    ops.B.repack2d_ld(cond.to(torch.float32).contiguous(), out.view(-1)[K:], rows, cd, rows, cd, width, 0)
    return out


def crop_columns(x, width_in, cols):
    """First ``cols`` columns of a [rows, width_in] matrix as a contiguous tensor."""
    rows = x.numel() // width_in
    out = _new(x, rows, cols)
    ops.B.repack2d(x, out, rows, width_in, rows, cols, 0)
    return out


# ------------------------------------------------------------------------------------------------
# weight packing: specs -> packed GEMM operands, either one kernel per entry or one launch for a whole plan
# ------------------------------------------------------------------------------------------------
K_KEEP, K_SWAP = 100, 101      # mmdyn_pack_entry.kind for conv weights; 0..5 = mmdyn_repack2d modes


def _spec(name, src, kind, rin, cin, rout, cout, shape, part=None):
    """part = (row_offset, col_offset, ld_out) when this entry fills a sub-block of the packed tensor `name`."""
    return dict(name=name, src=src, kind=kind, rin=rin, cin=cin, rout=rout, cout=cout, shape=tuple(shape), part=part)


def enc_layout(P):
    """Sequential indices of the convolutions that follow conv_net.0 (each has its BatchNorm at index + 1) and the
    input size they imply: 64 * 2**extra (models/shapes.py: the 128 / 256 pixel extensions add 32 -> 32 stages)."""
    idx = sorted(int(k.split(".")[1]) for k in P if k.startswith("conv_net.") and k.endswith(".weight") and P[k].dim() == 4)
    assert idx[0] == 0 and len(idx) >= 4, idx
    return idx[1:], 64 << (len(idx) - 4)


def dec_layout(P):
    """Sequential indices of the transposed convolutions in front of the last one, the index of the last one, and the
    output size."""
    idx = sorted(int(k.split(".")[1]) for k in P if k.startswith("hallucinate.") and k.endswith(".weight") and P[k].dim() == 4)
    assert len(idx) >= 4, idx
    return idx[:-1], idx[-1], 64 << (len(idx) - 4)


def enc_keys(extra=0):
    ks = ["conv_net.0.weight"]
    for j in range(3 + extra):
        ks += [f"conv_net.{2 + 3 * j}.weight", f"conv_net.{3 + 3 * j}.weight", f"conv_net.{3 + 3 * j}.bias"]
    return ks + ["fc_net.0.weight", "fc_net.0.bias"]


def dec_keys(extra=0):
    ks = ["upsample.0.weight", "upsample.0.bias"]
    for j in range(3 + extra):
        ks += [f"hallucinate.{3 * j}.weight", f"hallucinate.{3 * j + 1}.weight", f"hallucinate.{3 * j + 1}.bias"]
    return ks + [f"hallucinate.{3 * (3 + extra)}.weight"]


def encoder_pack_specs(P):
    """Packed GEMM operands of an image encoder: W1p (first layer, im2col form), W{j}k / W{j}s (tap-major forward /
    swapped input-gradient form of the j-th convolution, j = 2..), Wf / WfT (FC, flatten order absorbed)."""
    def c(name, key, kind):
        d0, d1 = P[key].shape[0], P[key].shape[1]
        return _spec(name, P[key], kind, d0, d1, 0, 0, (16, d1, d0) if kind == K_SWAP else (16, d0, d1))
    convs, _ = enc_layout(P)
    fwd = [c(f"W{j + 2}k", f"conv_net.{i}.weight", K_KEEP) for j, i in enumerate(convs)]
    bwd = [c(f"W{j + 2}s", f"conv_net.{i}.weight", K_SWAP) for j, i in reversed(list(enumerate(convs)))]
    return ([_spec("W1p", P["conv_net.0.weight"], 0, 32, 48, 32, 64, (32, 64))] + fwd +
            [_spec("Wf", P["fc_net.0.weight"], 2, 512, FEAT, 512, FEAT, (512, FEAT)),
             _spec("WfT", P["fc_net.0.weight"], 4, 512, FEAT, FEAT, 512, (FEAT, 512))] + bwd)


def decoder_pack_specs(P):
    """Wu / bu / WuT (FC), W{j}s (forward form of the j-th transposed convolution, j = 1..), W{n}p (last layer's
    input-gradient operand, im2col form), W{j}k (input-gradient form)."""
    L = P["upsample.0.weight"].shape[1]          # latent (+ condition_dim)
    Lp = _pad32(L)

    def c(name, key, kind):
        d0, d1 = P[key].shape[0], P[key].shape[1]
        return _spec(name, P[key], kind, d0, d1, 0, 0, (16, d1, d0) if kind == K_SWAP else (16, d0, d1))
    convs, last, _ = dec_layout(P)
    fwd = [c(f"W{j + 1}s", f"hallucinate.{i}.weight", K_SWAP) for j, i in enumerate(convs)]
    bwd = [c(f"W{j + 1}k", f"hallucinate.{i}.weight", K_KEEP) for j, i in reversed(list(enumerate(convs)))]
    return ([_spec("Wu", P["upsample.0.weight"], 3, FEAT, L, FEAT, Lp, (FEAT, Lp)),
             _spec("bu", P["upsample.0.bias"], 3, FEAT, 1, FEAT, 1, (FEAT,))] + fwd +
            [_spec(f"W{len(convs) + 1}p", P[f"hallucinate.{last}.weight"], 0, 32, 48, 32, 64, (32, 64))] + bwd +
            [_spec("WuT", P["upsample.0.weight"], 5, FEAT, L, Lp, FEAT, (Lp, FEAT))])


def heads_pack_specs(P):
    """K = 512 (+ condition_dim, zero-padded to a multiple of 32 for the conditional models)."""
    Wm, Wl = P["linear_means.weight"], P["linear_log_var.weight"]
    L, K = Wm.shape
    Kp = _pad32(K)
    return [_spec("Wh", Wm, 0, L, K, L, Kp, (2 * L, Kp), (0, 0, Kp)), _spec("Wh", Wl, 0, L, K, L, Kp, (2 * L, Kp), (L, 0, Kp)),
            _spec("bh", P["linear_means.bias"], 0, L, 1, L, 1, (2 * L,), (0, 0, 1)),
            _spec("bh", P["linear_log_var.bias"], 0, L, 1, L, 1, (2 * L,), (L, 0, 1)),
            _spec("WhT", Wm, 1, L, K, Kp, L, (Kp, 2 * L), (0, 0, 2 * L)),
            _spec("WhT", Wl, 1, L, K, Kp, L, (Kp, 2 * L), (0, L, 2 * L))]


def _alloc_packed(specs, like, w_dtype=None, pre=None):
    """``pre``: {name: tensor} destinations that already exist (views into a grouped launch's contiguous operands)."""
    wd = W_DTYPE if w_dtype is None else w_dtype
    out = dict(pre or {})
    for s in specs:
        if s["name"] not in out:       # matrices are GEMM operands (W_DTYPE); vectors are biases (always fp32)
            out[s["name"]] = torch.zeros(s["shape"], device=like.device, dtype=wd if len(s["shape"]) >= 2 else torch.float32)
    return out


PLANE_TWIN_2D = ("Wu", "WfT")


def wants_plane_twin(s):
    """Conv-weight packs [16][N][Cin] whose launches a plane kernel can serve (N % 64 == 0: the plane-ring kernel; N == 32: the
    patch-resident up-sampling kernel; Cin % 32 == 0) also get a Planes twin from the plan in the fp32x3 arithmetic (whether a
    given batch's launch takes it: planes_served)."""
    if s["kind"] < K_KEEP and not FC_PLANES:
        return False
    if s["kind"] < K_KEEP:
        # FC-level operands whose Linear launch the plane-ring kernel can serve (round 6): the decoder's Wu [6400][L] and the
        # encoder's WfT [6400][512] -- whole, unpadded 2-D packs with K % 32 == 0 and N % 128 == 0
        return (s["name"] in PLANE_TWIN_2D and s["part"] is None and len(s["shape"]) == 2 and s["shape"][1] % 32 == 0
                and s["shape"][0] % 128 == 0 and s["cout"] == s["shape"][1] and s["rout"] == s["shape"][0])
    return (s["kind"] >= K_KEEP and len(s["shape"]) == 3 and (s["shape"][1] % 64 == 0 or s["shape"][1] == 32)
            and s["shape"][2] % 32 == 0)


def pack_now(specs, pre=None):
    """One kernel per entry (module-API path)."""
    out = _alloc_packed(specs, specs[0]["src"], pre=pre)
    for s in specs:
        dst = out[s["name"]]
        if s["kind"] >= K_KEEP:
            ops.B.pack_conv_weight(s["src"], dst, s["rin"], s["cin"], s["kind"] - K_KEEP)
        elif s["part"] is None:
            ops.B.repack2d_ld(s["src"], dst, s["rin"], s["cin"], s["rout"], s["cout"], s["cout"], s["kind"])
        else:
            r0, c0, ld = s["part"]
            corner = dst.view(-1)[r0 * ld + c0:]
            ops.B.repack2d_ld(s["src"], corner, s["rin"], s["cin"], s["rout"], s["cout"], ld, s["kind"])
    return out


class PackPlan:
    """All repacks of the given spec lists as ONE kernel launch (mmdyn_pack_plan).  Source and destination
    storage must not move afterwards (the fused engine's flat parameter buffer and these outputs never do)."""

    def __init__(self, named_specs, early=(), w_dtype=None, prealloc=None, plane_twins=False):
        """``early``: names of the packed tensors the first phase of the step needs; they go to the front of the
        table so that :meth:`run_early` / :meth:`run_late` can launch the two halves at different points.
        ``plane_twins`` (fp32x3): the conv-weight packs the plane-ring kernel can use are ALSO written as Planes by the same
        launch (one more table entry each, dst_bf16 = 3) and registered in PLANE_TWIN under their fp32 pack."""
        import ctypes
        from ._lib import PackEntry
        self.packed, entries = {}, []
        order = []
        for key, specs in named_specs.items():
            outs = _alloc_packed(specs, specs[0]["src"], w_dtype, (prealloc or {}).get(key))
            self.packed[key] = outs
            order += [(0 if s["name"] in early else 1, len(order) + i, outs, s) for i, s in enumerate(specs)]
        order.sort(key=lambda t: (t[0], t[1]))
        self.n_early = sum(1 for t in order if t[0] == 0)
        self.twins = []
        for _, _, outs, s in order:
            for twin in ((False, True) if (plane_twins and wants_plane_twin(s) and outs[s["name"]].dtype == torch.float32) else (False,)):
                dst = outs[s["name"]]
                e = PackEntry()
                e.src = s["src"].data_ptr()
                e.kind, e.rows_in, e.cols_in = s["kind"], s["rin"], s["cin"]
                if twin:
                    conv = s["kind"] >= K_KEEP
                    pl = ops.Planes(16 * s["shape"][1] if conv else s["shape"][0], s["shape"][2] if conv else s["shape"][1], dst.device)
                    PLANE_TWIN[dst.data_ptr()] = (weakref.ref(dst), pl)
                    self.twins.append((dst.data_ptr(), pl))
                    e.dst, e.rows_out, e.cols_out, e.ld_out, e.dst_bf16 = pl.t.data_ptr(), 0, 0, 0, 3
                    if not conv:           # (2-D kinds: the same permutation as the fp32 pack, rows of [plane][cols_out])
                        e.rows_out, e.cols_out, e.ld_out = s["rout"], s["cout"], s["cout"]
                    if s["name"] in early:
                        self.n_early += 1
                    entries.append(e)
                    continue
                if s["kind"] >= K_KEEP:
                    e.dst, e.rows_out, e.cols_out, e.ld_out = dst.data_ptr(), 0, 0, 0
                else:
                    r0, c0, ld = s["part"] if s["part"] is not None else (0, 0, s["cout"])
                    e.dst = dst.data_ptr() + dst.element_size() * (r0 * ld + c0)
                    e.rows_out, e.cols_out, e.ld_out = s["rout"], s["cout"], ld
                e.dst_bf16 = {torch.bfloat16: 1, torch.float16: 2}.get(dst.dtype, 0)
                entries.append(e)
        self.n = len(entries)
        arr = (PackEntry * self.n)(*entries)
        raw = bytes(memoryview(arr))
        dev = next(iter(next(iter(self.packed.values())).values())).device
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        self._ptrs = [(e.src, e.dst) for e in entries]

        self.entry_bytes = len(raw) // max(self.n, 1)

    def run(self):
        ops.B.pack_plan(self.table, self.n)

    def run_early(self):
        if self.n_early:
            ops.B.pack_plan(self.table[: self.n_early * self.entry_bytes], self.n_early)

    def run_late(self):
        if self.n > self.n_early:
            ops.B.pack_plan(self.table[self.n_early * self.entry_bytes:], self.n - self.n_early)


# ------------------------------------------------------------------------------------------------
# generic Linear (MFMA when both dims are multiples of 32, the small kernel for the 7-wide pose ends)
# ------------------------------------------------------------------------------------------------
def linear_forward(x, W, b, act=ACT_NONE, out=None):
    """y = act(x W^T + b).  Returns (pre_activation or None, y).  ``out`` (act == ACT_NONE, MFMA path): destination of y."""
    rows, K = x.shape
    N = W.shape[0]
    if K % 32 == 0 and N % 32 == 0:
        if act == ACT_NONE:
            u, _ = dense(x, W, b, rows, K, N, out=out)
            return None, u
        u, y = dense(x, W, b, rows, K, N, act, want_act=True)
        return u, y
    y = _new(x, rows, N)
    ops.B.linear_small_fwd(x, W, b, y, rows, K, N, act)
    return None, y


def linear_backward(dy, x, W, gW, gb, need_dx=True, x_act=ACT_NONE, Wt=None):
    """Gradients of y = x W^T + b given dy (already through the activation).  Returns dx or None.
    ``x_act``: x is the OUTPUT of that activation (ReLU: act'(u) = [x > 0]) and dx is returned through it: dL/du.
    ``Wt``: W^T [K][N] where the caller's pack plan has written it (else transposed here, one launch)."""
    rows, K = x.shape
    N = W.shape[0]
    if K % 32 == 0 and N % 32 == 0:
        wgrad(dy, x, gW, DENSE, rows, 1, N, 1, K)
        if gb is not None:
            ops.B.colsum(dy, gb, rows, N, 0, 0.0)
        if not need_dx:
            return None
        if Wt is None:
            Wt = repack(W, N, K, K, N, 1)
        if x_act != ACT_NONE:
            return dgrad_act(dy, Wt, DENSE, 1, rows, 1, N, 1, K, x, x_act)
        dx, _ = dense(dy, Wt, None, rows, N, K)
        return dx
    dx = _new(x, rows, K) if need_dx else None
    ops.B.linear_small_bwd(dy, x, W, dx, gW, gb, rows, K, N, 0.0)
    return act_backward(dx, x, x_act) if (need_dx and x_act != ACT_NONE) else dx


def act_backward(dh, u, act):
    du = torch.empty_like(u)
    ops.B.act_bwd(dh, u, du, act)
    return du


# ------------------------------------------------------------------------------------------------
# image encoder trunk: conv_net + fc_net (up to, not including, the dropout)
# ------------------------------------------------------------------------------------------------
ENC_KEYS = enc_keys(0)                      # the reference's 64 x 64 encoder


def _bn_of(P, buf, pre):
    b = buf or {}
    return BNState(P[pre + ".weight"], P[pre + ".bias"], b.get(pre + ".running_mean"),
                   b.get(pre + ".running_var"), b.get(pre + ".num_batches_tracked"), b.get(pre + ".eval_stats"))


def run(gen):
    """Drive a stack generator to completion and return its result."""
    try:
        while True:
            next(gen)
    except StopIteration as e:
        return e.value


def interleave(jobs):
    """jobs: list of (stream_context_factory, generator).  Advances the generators round-robin, one layer at a
    time, each under its own stream context, so that the kernels of independent stacks are *enqueued* alternately
    (host launches -- and the node order of a captured HIP graph -- then feed both lanes evenly instead of one
    stack after the other).  Returns the generators' results in order."""
    results = [None] * len(jobs)
    live = list(range(len(jobs)))
    while live:
        for i in list(live):
            ctx, gen = jobs[i]
            with ctx():
                try:
                    next(gen)
                except StopIteration as e:
                    results[i] = e.value
                    live.remove(i)
    return results


def encoder_trunk_forward(*a, **k):
    return run(encoder_trunk_forward_steps(*a, **k))


def encoder_trunk_forward_steps(P, buf, x, G=1, repeat=1, packed=None, training=True):
    """x: NCHW [Bt,3,S,S] (S = 64, or 128 / 256 for the extended stacks) -> h = Swish(fc(conv stack)) [Bt,512];
    returns (h, ctx).  Generator: yields after every layer (see :func:`interleave`).
    ``repeat``: how many reference forward calls this one stands for (running-stat EMA updates).
    ``packed``: pre-packed weights (PackPlan); packed here, one kernel each, when absent."""
    Bt = x.shape[0]
    Bg = Bt // G
    convs, S = enc_layout(P)
    if tuple(x.shape[1:]) != (3, S, S):
        raise ValueError(f"mmdyn_hip: this encoder takes [B,3,{S},{S}] images, got {tuple(x.shape)}")
    pk = packed if packed is not None else pack_now(encoder_pack_specs(P))
    c = {"Bt": Bt, "G": G, "Bg": Bg, "pk": pk, "S": S}
    H = S // 2
    W1p = pk["W1p"]                                                          # [32][64], cols 48.. zero
    u1, a1 = _act(x, Bt * H * H, 32), _act(x, Bt * H * H, 32)
    # first layer: the k4 s2 p1 window of the NCHW image is gathered on the fly (no im2col matrix in HBM)
    ops.B.igemm_nt(x, W1p, None, u1, a1, None, None, IM2COL3, 1, Bt, S, S, 64, H, H, 32, 32, 1, 0,
                   ACT_SWISH, 1)
    yield
    stages, a, cin = [], a1, 32
    # fp32x3: an activation whose consumer GEMM takes plane operands exists ONLY as ops.Planes from here on (the convolution
    # that reads it and the weight gradient that reads it as `a_in` both take it so); conv_net.0's output comes out of its kernel
    # as fp32 and is split by its own launch
    # (eval mode too since round 6: the inference engine's launches take the same plane operands as the train step's)
    if len(convs) and planes_served(CONV, G, Bg, H, 32, H // 2, P[f"conv_net.{convs[0]}.weight"].shape[0]):
        a = as_planes(a1, 32)
    for j, i in enumerate(convs):
        cout = P[f"conv_net.{i}.weight"].shape[0]
        last = j == len(convs) - 1                                           # Conv2d(128,256,4,1,0): 8 -> 5
        Ho, stride, offset = (H - 3, 1, 0) if last else (H // 2, 2, -1)
        bn = _bn_of(P, buf, f"conv_net.{i + 1}")
        y, st, T = conv_like(a, pk[f"W{j + 2}k"], CONV, G, Bg, H, cin, Ho, cout, stride, offset, training)
        want = False           # does the NEXT convolution take its operand already split?  Then BatchNorm + Swish writes it so.
        if not last:
            nlast = j + 1 == len(convs) - 1
            want = planes_served(CONV, G, Bg, Ho, cout, Ho - 3 if nlast else Ho // 2, P[f"conv_net.{convs[j + 1]}.weight"].shape[0])
        an, m, r = bn_swish_from_partials(y, st, T, bn, G, Bg * Ho * Ho, cout, repeat, planes=want)
        stages.append(dict(i=i, Hi=H, Ho=Ho, cin=cin, cout=cout, stride=stride, offset=offset, a_in=a, y=y, a=an, m=m, r=r,
                           bn=bn))
        a, cin, H = an, cout, Ho
        yield
    u5, h = dense(a, pk["Wf"], P["fc_net.0.bias"], Bt, FEAT, 512, ACT_SWISH, want_act=True)   # columns hw*256+c
    c.update(x=x, u1=u1, a1=a1, u5=u5, stages=stages)
    return h, c


def encoder_trunk_backward(*a, **k):
    return run(encoder_trunk_backward_steps(*a, **k))


def encoder_trunk_backward_steps(P, c, dh, grads, dh_is_du=False, dh_planes=None):
    """dh: [Bt,512]; writes every weight gradient of the trunk into ``grads[key]`` (canonical layout).
    ``dh_is_du``: the caller has already taken dh through the FC layer's Swish (c["u5"]), e.g. in its dropout backward.
    ``dh_planes`` (with dh_is_du): the same tensor as an ops.Planes -- the FC layer's input gradient then runs on the plane-ring
    kernel where it serves the launch."""
    Bt, G, Bg, pk, S, st = c["Bt"], c["G"], c["Bg"], c["pk"], c["S"], c["stages"]
    n = len(st)
    du5 = act_backward(dh, c["u5"], ACT_SWISH) if not dh_is_du else dh
    wgrad(du5, st[-1]["a"], grads["fc_net.0.weight"], DENSE, Bt, 1, 512, 1, FEAT, perm=1)
    ops.B.colsum(du5, grads["fc_net.0.bias"], Bt, 512, 0, 0.0)
    da, _ = dense(du5, pk["WfT"], None, Bt, 512, FEAT, out_dtype=ACT_DTYPE,  # WfT: [hw*256+c][512]
                  A_planes=dh_planes if dh_is_du else None)
    yield

    def bn_keys(t):
        return grads[f"conv_net.{t['i'] + 1}.weight"], grads[f"conv_net.{t['i'] + 1}.bias"]

    # the k4 s1 p0 stage (8 -> 5): its input gradient is the tap-skipping transposed convolution
    # (fp32x3: a dL/dy whose input-gradient launch takes plane operands is written by its apply pass ONLY as ops.Planes; the weight
    #  gradient takes it so as well)
    t = st[n - 1]
    dy = bn_swish_backward(da, t["y"], t["m"], t["r"], t["bn"], *bn_keys(t), G, Bg * t["Ho"] ** 2, t["cout"],
                           planes_out=planes_served(TCONV_S1P0, 1, Bt, 5, t["cout"], 8, t["cin"]))
    wgrad(dy, t["a_in"], grads[f"conv_net.{t['i']}.weight"], CONV, Bt, t["Ho"], t["cout"], t["Hi"], t["cin"], 1, 0)
    da = tconv_s1p0(dy, pk[f"W{n + 1}s"], 1, Bt, t["cout"], t["cin"])[0]     # W{n+1}s: [16][Cin][Cout]
    yield
    t = st[n - 2]
    dy = bn_swish_backward(da, t["y"], t["m"], t["r"], t["bn"], *bn_keys(t), G, Bg * t["Ho"] ** 2, t["cout"],
                           planes_out=planes_served(TCONV_S2P1, G if n >= 3 else 1, Bg if n >= 3 else Bt, t["Ho"], t["cout"], t["Hi"],
                                                    t["cin"]))
    wgrad(dy, t["a_in"], grads[f"conv_net.{t['i']}.weight"], CONV, Bt, t["Ho"], t["cout"], t["Hi"], t["cin"], 2, -1)
    yield
    for k in range(n - 3, -1, -1):
        # input gradient of stage k+1 with stage k's BatchNorm+Swish backward in its epilogue
        up, t = st[k + 1], st[k]
        dy = dgrad_bn_swish_backward(dy, pk[f"W{k + 3}s"], TCONV_S2P1, G, Bg, up["Ho"], up["cout"], up["Hi"], up["cin"], 1, 0,
                                     t["y"], t["m"], t["r"], t["bn"], *bn_keys(t),
                                     planes_out=planes_served(TCONV_S2P1, G if k > 0 else 1, Bg if k > 0 else Bt, t["Ho"], t["cout"],
                                                              t["Hi"], t["cin"]))
        wgrad(dy, t["a_in"], grads[f"conv_net.{t['i']}.weight"], CONV, Bt, t["Ho"], t["cout"], t["Hi"], t["cin"], 2, -1)
        if k > 0:
            yield
    t = st[0]
    du1 = dgrad_act(dy, pk["W2s"], TCONV_S2P1, 1, Bt, t["Ho"], t["cout"], t["Hi"], t["cin"], c["u1"], ACT_SWISH)
    yield
    wgrad(du1, c["x"], grads["conv_net.0.weight"], IM2COL3, Bt, S // 2, 32, S, 64, cg_canon=48)


# ------------------------------------------------------------------------------------------------
# image decoder: upsample FC + hallucinate stack -> logits (NCHW)
# ------------------------------------------------------------------------------------------------
DEC_KEYS = dec_keys(0)                      # the reference's 64 x 64 decoder


def decoder_forward(*a, **k):
    return run(decoder_forward_steps(*a, **k))


def decoder_forward_steps(P, buf, z, G=1, repeat=1, logits=True, packed=None, cond=None, training=True, loss=None, z_planes=None):
    """z: [Bt, L] -> logits NCHW [Bt,3,S,S]; returns (logits, ctx).  ``logits=False`` stops after the last
    BatchNorm (used only to reproduce the running statistics of the reference's unused decoder passes).
    ``loss`` (fused engine): dict(target [Bg,3,S,S], slots [G], acc (fp64 loss slots), grad_scale, want_grad, keep (group whose logits
    are published, or None: all), mask, mask_channels, acc_u) -- where the last layer is the direct fused kernel, the BCE term of
    problems.py:433-437 rides in its epilogue: ctx["dl"] = dlogits (or None), ctx["loss_fused"] = True, and the returned logits hold
    group ``keep`` only ([Bg,3,S,S]).  Otherwise ctx["loss_fused"] is False and the caller runs the loss kernel."""
    Bt, L0 = z.shape
    Bg = Bt // G
    convs, last, S = dec_layout(P)
    pk = packed if packed is not None else pack_now(decoder_pack_specs(P))
    Lc = P["upsample.0.weight"].shape[1]         # latent + condition_dim
    L = _pad32(Lc)
    if cond is not None or L != L0:
        z = concat_condition(z, cond, L)          # [z | c | 0]  (vae.py:286-291)
    c = {"Bt": Bt, "G": G, "Bg": Bg, "L": L, "L0": L0, "Lc": Lc, "z": z, "pk": pk, "S": S, "last": last}
    # rows -> hw*256+c.  bf16 storage mode: the activated output is the first transposed convolution's operand and is stored
    # as such (the matrix cores round it to bf16 either way); the pre-activation stays fp32 for the backward
    c0 = P[f"hallucinate.{convs[0]}.weight"].shape[0]
    want0 = planes_served(TCONV_S1P0, G, Bg, 5, c0, 8, P[f"hallucinate.{convs[0]}.weight"].shape[1])
    # fp32x3 (round 6): with z arriving split (``z_planes``: written by the product-of-experts launch) and the packed weight's plane
    # twin the Linear layer runs on the plane-ring kernel and hands its activated output to the first transposed convolution as
    # plane rows of its 256 channels -- no stand-alone split launch, no fp32 copy of h0 (the backward reads u0)
    fc_planes = (want0 and z_planes is not None and cond is None and L == L0 and ACT_DTYPE == torch.float32
                 and FEAT % c0 == 0 and dense_planes_served(Bt, L, FEAT) and _plane_twin(pk["Wu"]) is not None)
    if fc_planes:
        u0, h0 = dense(z, pk["Wu"], pk["bu"], Bt, L, FEAT, ACT_SWISH, want_act=True, A_planes=z_planes, act_planes=c0)
    else:
        u0, h0 = dense(z, pk["Wu"], pk["bu"], Bt, L, FEAT, ACT_SWISH, want_act=True,
                       act_dtype=ACT_DTYPE if ACT_DTYPE != torch.float32 else None)
    yield
    stages, a, H = [], h0, 5
    if want0 and not fc_planes:
        a = as_planes(h0, c0)       # (the FC kernel's fp32 output, split by its own launch)
    for j, i in enumerate(convs):
        cin, cout = P[f"hallucinate.{i}.weight"].shape[0], P[f"hallucinate.{i}.weight"].shape[1]
        bn = _bn_of(P, buf, f"hallucinate.{i + 1}")
        if j == 0:                                                           # ConvTranspose2d(256,128,4,1,0): 5 -> 8
            Ho = H + 3
            y, st, T = tconv_s1p0(a, pk["W1s"], G, Bg, cin, cout, stats=training)       # W1s: [16][Cout][Cin]
        else:
            Ho = 2 * H
            y, st, T = conv_like(a, pk[f"W{j + 1}s"], TCONV_S2P1, G, Bg, H, cin, Ho, cout, stats=training)
        if j == len(convs) - 1 and fuses_last_bn(S) and cout == 32:
            # last BatchNorm + Swish: statistics only -- the two consumers of the activated tensor (the 3-channel output layer
            # below, the last layer's weight gradient) apply it on their operand fetch, the tensor itself is never written
            m, r = bn_stats_only(y, st, T, bn, G, Bg * Ho * Ho, cout, repeat)
            an = None
        else:
            # (fp32x3: written ONLY as ops.Planes when the next transposed convolution takes plane operands)
            want = j + 1 < len(convs) and planes_served(TCONV_S2P1, G, Bg, Ho, cout, 2 * Ho,
                                                                     P[f"hallucinate.{convs[j + 1]}.weight"].shape[1])
            an, m, r = bn_swish_from_partials(y, st, T, bn, G, Bg * Ho * Ho, cout, repeat, planes=want)
        stages.append(dict(i=i, Hi=H, Ho=Ho, cin=cin, cout=cout, a_in=a, y=y, a=an, m=m, r=r, bn=bn))
        a, H = an, Ho
        yield
    out = None
    c["loss_fused"] = False
    if logits:
        t = stages[-1]
        if t["a"] is None and loss is not None:
            keep = loss.get("keep")
            out = _new(z, Bt if keep is None else Bg, 3, S, S)
            c["dl"] = _new(z, Bt, 3, S, S) if loss["want_grad"] else None
            ops.B.tconv_out3_bn_bce(t["y"], t["m"], t["r"], t["bn"].gamma, t["bn"].beta, P[f"hallucinate.{last}.weight"], out,
                                    -1 if keep is None else keep, loss["target"], c["dl"], loss["acc"], loss["slots"],
                                    loss["grad_scale"], G, Bg, H, H, mask=loss.get("mask"),
                                    mask_channels=loss.get("mask_channels", 1), unmasked_slots=loss.get("acc_u"))
            c["loss_fused"] = True
            c.update(u0=u0, h0=h0, stages=stages)
            return out, c
        out = _new(z, Bt, 3, S, S)
        if t["a"] is None:
            ops.B.tconv_out3_bn_fwd(t["y"], t["m"], t["r"], t["bn"].gamma, t["bn"].beta, P[f"hallucinate.{last}.weight"], out, G, Bg,
                                    H, H)
        else:
            ops.B.tconv_out3_fwd(a, P[f"hallucinate.{last}.weight"], out, Bt, H, H)     # direct kernel, canonical weights
    c.update(u0=u0, h0=h0, stages=stages)
    return out, c


def decoder_backward(*a, **k):
    return run(decoder_backward_steps(*a, **k))


def decoder_backward_steps(P, c, dlogits, grads, need_dz=True, defer=None):
    """dlogits: NCHW [Bt,3,S,S] -> dz [Bt, L]; weight gradients into ``grads`` (``defer``: queued, see wgrad)."""
    Bt, G, Bg, L, pk, S, st = c["Bt"], c["G"], c["Bg"], c["L"], c["pk"], c["S"], c["stages"]
    n = len(st)

    def bn_keys(t):
        return grads[f"hallucinate.{t['i'] + 1}.weight"], grads[f"hallucinate.{t['i'] + 1}.bias"]

    # last layer backward: both GEMMs gather the k4 s2 p1 window of the NCHW logit gradient on the fly
    t = st[n - 1]
    if t["a"] is None:            # (the activated tensor of the last BatchNorm was never stored: recomputed on the fetch)
        wgrad_out3_bn(t["y"], t["m"], t["r"], t["bn"], dlogits, grads[f"hallucinate.{c['last']}.weight"], G, Bg, S // 2, defer=defer)
    else:
        wgrad(t["a"], dlogits, grads[f"hallucinate.{c['last']}.weight"], IM2COL3, Bt, S // 2, 32, S, 64, cg_canon=48, defer=defer)
    # the input-gradient GEMM of every layer carries the BatchNorm+Swish backward of the layer below in its epilogue
    def next_takes_planes(k):
        """the input-gradient launch that consumes stage k's dL/dy: of stage k (k >= 1), or the k4 s1 p0 layer's (k = 0)"""
        if k >= 1:
            u = st[k]
            return planes_served(CONV, G, Bg, u["Ho"], u["cout"], u["Hi"], u["cin"])
        return planes_served(CONV, 1, Bt, 8, st[0]["cout"], 5, st[0]["cin"])

    dy = dgrad_bn_swish_backward(dlogits, pk[f"W{n + 1}p"], IM2COL3, G, Bg, S, 64, S // 2, 32, 1, 0, t["y"], t["m"], t["r"],
                                 t["bn"], *bn_keys(t), planes_out=next_takes_planes(n - 1))
    yield
    for k in range(n - 1, 0, -1):
        up, t = st[k], st[k - 1]
        wgrad(up["a_in"], dy, grads[f"hallucinate.{up['i']}.weight"], CONV, Bt, up["Hi"], up["cin"], up["Ho"], up["cout"], 2, -1,
              defer=defer)
        dy = dgrad_bn_swish_backward(dy, pk[f"W{k + 1}k"], CONV, G, Bg, up["Ho"], up["cout"], up["Hi"], up["cin"], 2, -1,
                                     t["y"], t["m"], t["r"], t["bn"], *bn_keys(t), planes_out=next_takes_planes(k - 1))
        yield
    t = st[0]
    wgrad(t["a_in"], dy, grads[f"hallucinate.{t['i']}.weight"], CONV, Bt, 5, t["cin"], 8, t["cout"], 1, 0, defer=defer)
    # input gradient of the k4 s1 p0 layer with the FC layer's Swish backward in its epilogue (FC level: fp32)
    du0 = dgrad_act(dy, pk["W1k"], CONV, 1, Bt, 8, t["cout"], 5, t["cin"], c["u0"], ACT_SWISH, 1, 0)
    yield
    wgrad(du0, c["z"], grads["upsample.0.weight"], DENSE, Bt, 1, FEAT, 1, L, cg_canon=c["Lc"], perm=2, defer=defer)
    ops.B.colsum(du0, grads["upsample.0.bias"], Bt, FEAT, 2, 0.0)
    if not need_dz:
        return None
    dz, _ = dense(du0, pk["WuT"], None, Bt, FEAT, L)                         # WuT: [L][hw*256+c]
    return dz if L == c["L0"] else crop_columns(dz, L, c["L0"])


# ------------------------------------------------------------------------------------------------
# fused heads (linear_means | linear_log_var) and the pose MLPs
# ------------------------------------------------------------------------------------------------
def heads_forward(P, hd, packed=None, cond=None):
    """hd: [rows, 512] (+ cond [rows, cd] for the conditional models) -> out [rows, 2L]: columns [0,L) = means,
    [L,2L) = log-variances."""
    L, K = P["linear_means.weight"].shape
    Kp = _pad32(K)
    pk = packed if packed is not None else pack_now(heads_pack_specs(P))
    K0 = hd.shape[1]                                   # width of the features proper (512 for the image encoders)
    if cond is not None or Kp != K0:
        hd = concat_condition(hd, cond, Kp)
    out, _ = dense(hd, pk["Wh"], pk["bh"], hd.shape[0], Kp, 2 * L)
    return out, {"hd": hd, "pk": pk, "L": L, "K": K, "Kp": Kp, "K0": K0}


def heads_backward(c, dout, grads, need_dx=True, fused=None):
    """``fused``: (gW [2L][K], gb [2L]) views covering both heads' gradients where the caller keeps them adjacent (the
    fused engine's flat buffer): the fused GEMM's gradients are then written in place, no split copies."""
    hd, L, K, Kp = c["hd"], c["L"], c["K"], c["Kp"]
    rows = hd.shape[0]
    if fused is not None and K == Kp:
        wgrad(dout, hd, fused[0], DENSE, rows, 1, 2 * L, 1, Kp)
        ops.B.colsum(dout, fused[1], rows, 2 * L, 0, 0.0)
    else:
        gW, gb = _new(hd, 2 * L, Kp), _new(hd, 2 * L)
        wgrad(dout, hd, gW, DENSE, rows, 1, 2 * L, 1, Kp)
        ops.B.colsum(dout, gb, rows, 2 * L, 0, 0.0)
        ops.B.repack2d(gW[:L], grads["linear_means.weight"], L, Kp, L, K, 0)      # drops the zero-padded columns
        ops.B.repack2d(gW[L:], grads["linear_log_var.weight"], L, Kp, L, K, 0)
        ops.B.repack2d(gb[:L], grads["linear_means.bias"], L, 1, L, 1, 0)
        ops.B.repack2d(gb[L:], grads["linear_log_var.bias"], L, 1, L, 1, 0)
    if not need_dx:
        return None
    dx, _ = dense(dout, c["pk"]["WhT"], None, rows, 2 * L, Kp)
    return dx if Kp == c["K0"] else crop_columns(dx, Kp, c["K0"])


def heads_group_buffers(Ps, like, w_dtype=None):
    """Contiguous packed operands of the fused heads of ``len(Ps)`` encoders with equal head shapes: ({"Wh": [G][2L][Kp],
    "bh": [G][2L], "WhT": [G][Kp][2L]}, per-group views in the layout of heads_pack_specs) -- the weights of a grouped
    launch (mmdyn_igemm_nt_grouped) follow one another at a fixed stride."""
    L, K = Ps[0]["linear_means.weight"].shape
    Kp, G = _pad32(K), len(Ps)
    wd = W_DTYPE if w_dtype is None else w_dtype
    allb = {"Wh": torch.zeros(G, 2 * L, Kp, device=like.device, dtype=wd), "bh": torch.zeros(G, 2 * L, device=like.device),
            "WhT": torch.zeros(G, Kp, 2 * L, device=like.device, dtype=wd)}
    return allb, [{k: v[g] for k, v in allb.items()} for g in range(G)]


def heads_forward_grouped(hd_all, pk_all, G, rows):
    """hd_all [G*rows][Kp] (group g = rows [g*rows, (g+1)*rows)) -> out [G*rows][2L], every group on its own heads: the
    linear_means | linear_log_var pairs of the visual, tactile and pose encoders (vae.py:211-216, 239-240) as ONE launch."""
    _, N, Kp = pk_all["Wh"].shape
    out = _new(hd_all, G * rows, N)
    ops.B.igemm_nt_grouped(hd_all, pk_all["Wh"], pk_all["bh"], out, None, None, G, rows, Kp, N, ACT_NONE)
    return out


def heads_backward_grouped(dout_all, pk_all, G, rows):
    """Input gradients of :func:`heads_forward_grouped`, [G*rows][Kp], as one grouped GEMM (what the backward chain waits for)."""
    _, N, Kp = pk_all["Wh"].shape
    dx = _new(dout_all, G * rows, Kp)
    ops.B.igemm_nt_grouped(dout_all, pk_all["WhT"], None, dx, None, None, G, rows, N, Kp, ACT_NONE)
    return dx


def heads_wgrad_grouped(hd_all, dout_all, pk_all, gW_all, gb, G, rows):
    """Parameter gradients of :func:`heads_forward_grouped` (nothing on the backward chain reads them): the G weight
    gradients as one grouped GEMM + one slab reduction into ``gW_all`` ([G*2L][Kp], the adjacent heads' gradients of the flat
    buffer), the bias gradients into ``gb[g]`` ([2L] each)."""
    _, N, Kp = pk_all["Wh"].shape
    chunks = ops.B.wgrad_chunks(DENSE, rows, N, Kp)
    partial = _new(hd_all, chunks, G, N, Kp)
    ops.B.wgrad_tn_grouped(dout_all, hd_all, partial, G, rows, N, Kp, chunks)
    ops.B.wgrad_reduce(partial, gW_all, chunks, 1, G * N, Kp, Kp, 0, 0.0)
    for g in range(G):
        ops.B.colsum(dout_all[g * rows:(g + 1) * rows], gb[g], rows, N, 0, 0.0)


HEAD_KEYS = ["linear_means.weight", "linear_means.bias", "linear_log_var.weight", "linear_log_var.bias"]
POSE_ENC_KEYS = ["fc_net.0.weight", "fc_net.0.bias", "fc_net.2.weight", "fc_net.2.bias"]
POSE_DEC_KEYS = ["deconv_net.0.weight", "deconv_net.0.bias", "deconv_net.2.weight", "deconv_net.2.bias",
                 "deconv_net.4.weight", "deconv_net.4.bias"]


def pose_encoder_trunk_forward(P, pose, out=None):
    """Linear(7,512) ReLU Linear(512,512) (Identity): vae.py:218-222 with layer_sizes [512,512]."""
    _, h1 = linear_forward(pose, P["fc_net.0.weight"], P["fc_net.0.bias"], ACT_RELU)
    _, h2 = linear_forward(h1, P["fc_net.2.weight"], P["fc_net.2.bias"], ACT_NONE, out=out)
    return h2, {"x": pose, "h1": h1}


def pose_mlp_pack_specs(P, keys):
    """W^T of the pose MLPs' MFMA-sized Linear layers (the input-gradient GEMMs' operand), written by the step's pack plan:
    one entry "<key>T" per weight [N][K] -> [K][N]."""
    out = []
    for k in keys:
        N, K = P[k].shape
        if K % 32 == 0 and N % 32 == 0:
            out.append(_spec(k + "T", P[k], 1, N, K, K, N, (K, N)))
    return out


def pose_encoder_trunk_backward(P, c, dh2, grads, packed=None):
    # (ReLU: the sign of the output equals the sign of the input, so h1 stands in for the pre-activation)
    pk = packed or {}
    du1 = linear_backward(dh2, c["h1"], P["fc_net.2.weight"], grads["fc_net.2.weight"], grads["fc_net.2.bias"],
                          x_act=ACT_RELU, Wt=pk.get("fc_net.2.weightT"))
    linear_backward(du1, c["x"], P["fc_net.0.weight"], grads["fc_net.0.weight"], grads["fc_net.0.bias"],
                    need_dx=False)


def pose_decoder_forward(P, z):
    """Linear(L,512) ReLU Linear(512,512) ReLU Linear(512,7): vae.py:281-283."""
    _, h1 = linear_forward(z, P["deconv_net.0.weight"], P["deconv_net.0.bias"], ACT_RELU)
    _, h2 = linear_forward(h1, P["deconv_net.2.weight"], P["deconv_net.2.bias"], ACT_RELU)
    _, out = linear_forward(h2, P["deconv_net.4.weight"], P["deconv_net.4.bias"], ACT_NONE)
    return out, {"z": z, "h1": h1, "h2": h2}


def pose_decoder_backward(P, c, dout, grads, need_dz=True, packed=None):
    pk = packed or {}
    du2 = linear_backward(dout, c["h2"], P["deconv_net.4.weight"], grads["deconv_net.4.weight"],
                          grads["deconv_net.4.bias"], x_act=ACT_RELU)
    du1 = linear_backward(du2, c["h1"], P["deconv_net.2.weight"], grads["deconv_net.2.weight"],
                          grads["deconv_net.2.bias"], x_act=ACT_RELU, Wt=pk.get("deconv_net.2.weightT"))
    return linear_backward(du1, c["z"], P["deconv_net.0.weight"], grads["deconv_net.0.weight"],
                           grads["deconv_net.0.bias"], need_dx=need_dz, Wt=pk.get("deconv_net.0.weightT"))
