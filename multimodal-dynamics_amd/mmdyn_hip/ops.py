"""Tensor-level wrappers over the C ABI (include/mmdyn_hip.h).

Every method of :class:`HipBackend` maps 1:1 onto one ``mmdyn_*`` entry point: it validates device /
dtype / contiguity, passes raw device pointers and launches on torch's *current* HIP stream.  PyTorch is
used only for memory, streams and autograd plumbing -- all arithmetic of the hot path happens in
libmmdyn_hip.so.  There is no CPU / ATen fallback: a CPU tensor raises.

``B`` is the active backend.  The test-suite may swap in an emulation object (tests/emu_backend.py) to
exercise the host-side orchestration on a machine without a GPU; nothing in the product does.
"""
import ctypes

import os

import torch

from . import _lib
from ._lib import PassExperts, MAX_PASSES, MAX_EXPERTS, check

ACT_NONE, ACT_SWISH, ACT_RELU = 0, 1, 2
DENSE, CONV, TCONV_S2P1, IM2COL3, TCONV_S1P0 = 0, 1, 2, 3, 4


def _ptr(t, dtype=torch.float32):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("mmdyn_hip: the HIP path needs GPU tensors (got a CPU tensor); there is no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"mmdyn_hip: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError("mmdyn_hip: tensor must be contiguous")
    return t.data_ptr()


def _aptr(t):
    """Pointer of an ACTIVATION (or packed-weight) tensor: fp32, or 16-bit in the storage modes.  Returns (pointer, fmt):
    fmt 0 = fp32, 1 = bf16 ("bf16s"), 2 = IEEE half ("fp16s") -- truthy exactly for the 16-bit formats."""
    if t is None:
        return None, 0
    if t.dtype == torch.bfloat16:
        return _ptr(t, torch.bfloat16), 1
    if t.dtype == torch.float16:
        return _ptr(t, torch.float16), 2
    return _ptr(t), 0


def _is16(t):
    return t.dtype in (torch.bfloat16, torch.float16)


def _half(*fmts):
    """1 when the 16-bit tensors of one call are IEEE half, 0 when bf16 (or none); the two never mix in one call."""
    kinds = {f for f in fmts if f}
    if len(kinds) > 1:
        raise TypeError("mmdyn_hip: bf16 and fp16 tensors in one call")
    return int(2 in kinds)


def _stream():
    return torch.cuda.current_stream().cuda_stream


class Planes:
    """An fp32 matrix [rows][C] kept as its exact three-term bf16 split, the operand format of the fp32x3 GEMMs that take their
    operands ALREADY SPLIT (include/mmdyn_hip.h, flag bits 7 + 8): ``t[row][plane][c]`` (torch.bfloat16), plane 0 = hi = bf16(x),
    1 = mid = bf16(x - hi), 2 = lo = x - hi - mid; hi + mid + lo == x bit for bit.  Written by ``B.split_planes`` (or by the
    kernels that produce the tensor); a channels-last activation has one row per pixel, packed weights one per (tap, n)."""
    __slots__ = ("t",)

    def __init__(self, rows, C, device):
        if C % 8:
            raise ValueError("mmdyn_hip: plane rows are moved in 8-channel granules (C % 8 == 0)")
        self.t = torch.empty(rows, 3, C, dtype=torch.bfloat16, device=device)

    @property
    def rows(self):
        return self.t.shape[0]

    @property
    def C(self):
        return self.t.shape[2]

    def float(self):
        """The fp32 matrix back (exact: hi + mid has at most 16 significant bits, + lo is x)."""
        f = self.t.to(torch.float32)
        return (f[:, 0] + f[:, 1]) + f[:, 2]


# The "last block finishes" single-launch forms of mmdyn_bn_finalize / mmdyn_bn_bwd_finalize / mmdyn_colsum are not taken by
# EVERY launch: measured on the two-lane step that cost 0.12 ms (fp32 bs 256: 6.78 -> 6.90 ms; bf16s bs 128: 2.165 -> 2.19 ms) --
# up to 1024 blocks arrive on one counter (~12 ns per arrival) and every block drains its stores before it may leave
# (docs/LAB_NOTES.md D).  The BatchNorm finalize launches with SMALL partial tables take it (HipBackend.ticket_max_work, round 6:
# LAB_NOTES H.g); MMDYN_TICKET=1 switches it on everywhere (kernel tests, experiments).
_USE_TICKET = bool(os.environ.get("MMDYN_TICKET"))


class HipBackend:
    name = "hip"

    def __init__(self, lib_path=None):
        self._l, self._lib_path = None, lib_path       # lib_path: the LAB build (tests / microbenchmarks only)
        self._tickets = {}                             # device -> [zeroed int32 pool, next eager slot, next graph slot, free graph slots]
        self._drawn = None                             # (device, slot) pairs drawn by captured launches since ticket_mark()
        self.use_flags = False                         # (the product library ignores arrival words: see _flags)
        self._flagpool = {}                            # device -> [zeroed int32 pool, next eager block, next graph block, free graph blocks]
        self.force_ticket = False
        # BatchNorm finalize launches whose partial-sum table has at most this many (group, tile) rows take the single-launch
        # "last block finishes" form (0: none, rounds 3-5 -- the form lost on the whole step when EVERY finalize took it).  Same
        # box, interleaved (profiles/r6/ab_bn_finalize_small_tables.txt, bench.py --ticket-max-work): 1024 = nothing on the
        # power-bound configs[1] step, +1.1 % on bf16s bs 128, 18 launches fewer per step; bit-identical results either way
        self.ticket_max_work = 1024
        # "fp32": v_mfma_f32_32x32x2_f32 (the reference's arithmetic, the default and the BASELINE configs[1] path);
        # "bf16": operands rounded to bf16 on their way into the matrix cores, fp32 accumulate (configs[2]);
        # "fp16": the same with IEEE-half operands (configs[4]); "bf16s": bf16 + bf16 activation storage;
        # "fp16s": fp16 + fp16 activation storage
        self.precision = "fp32"
        # With precision "fp32": let the fp32 GEMM launches that gain from it run on the bf16 matrix cores through the EXACT
        # three-term split of their fp32 operands (x = hi + mid + lo, six of nine products, fp32 accumulate: csrc/igemm_nt.hip
        # X3; error against fp64 no larger than the native fp32 matrix cores').  The engines' precision "fp32x3" sets it.
        self.fp32_split = False

    TICKET_SLOTS, TICKET_STRIDE = 4096, 16             # one 64-byte line per slot

    def _ticket(self, like, work=None):
        """Address of a zero-initialised arrival counter for ONE launch (the "last block finishes" kernels:
        mmdyn_bn_finalize, mmdyn_bn_bwd_finalize, mmdyn_colsum); the kernel that used a slot leaves it zero.
        A launch recorded into a HIP graph keeps its slot for every replay, so captured launches draw from the lower half
        of the pool, and a slot handed out there belongs to that graph until its owner gives it back: the engines bracket a
        capture with ticket_mark() / ticket_take() and call ticket_release() with the slots of a graph they drop (a
        recapture on a new batch shape), so a long run that recaptures every epoch does not run out.  Eager
        launches cycle through the upper half: a slot comes up again only after 2048 eager launches of this kind -- far more
        than a train step issues -- so no two launches in flight share one, and none can meet a slot that a replaying graph
        owns."""
        if not (_USE_TICKET or self.force_ticket or (work is not None and work <= self.ticket_max_work)):
            return None
        dev = like.device
        ent = self._tickets.get(dev)
        if ent is None:
            ent = [torch.zeros(self.TICKET_SLOTS * self.TICKET_STRIDE, dtype=torch.int32, device=dev), 0, 0, []]
            self._tickets[dev] = ent
        pool, nxt_eager, nxt_graph, free = ent
        half = self.TICKET_SLOTS // 2
        if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
            if free:
                slot = free.pop()
            elif nxt_graph >= half:
                raise RuntimeError("mmdyn_hip: ticket slots for captured launches exhausted (2048 live per device)")
            else:
                ent[2] = nxt_graph + 1
                slot = nxt_graph
            if self._drawn is not None:
                self._drawn.append((dev, slot))
        else:
            ent[1] = (nxt_eager + 1) % half
            slot = half + nxt_eager
        return pool.data_ptr() + 4 * self.TICKET_STRIDE * slot

    # arrival words of the persistent stream-K GEMMs (igemm_wsp.hip, round 6): one block of FLAG_WORDS zeroed words per launch, the
    # same ownership rules as the tickets (captured launches own theirs until ticket_release, eager launches cycle through the upper
    # half); the kernel leaves every word it set back at zero.  Exhausted (hundreds of live captures): None -- the launch then takes
    # the two-launch form (slabs + fix-up kernel).  ``use_flags`` is False by default: finishing split tiles inside the launch is
    # built, tested bit-identical and measured SLOWER on the step (docs/LAB_NOTES.md H.a); the product library compiles it out and
    # only the LAB library honours the words (tests, bench.py --inkernel-finish with MMDYN_HIP_LIB pointing at the LAB build).
    FLAG_BLOCKS, FLAG_WORDS = 1024, 8192

    def _flags(self, like):
        if not self.use_flags:       # (A/B measurements and the bit-equality test against the two-launch form)
            return None
        dev = like.device
        ent = self._flagpool.get(dev)
        if ent is None:
            ent = [torch.zeros(self.FLAG_BLOCKS * self.FLAG_WORDS, dtype=torch.int32, device=dev), 0, 0, []]
            self._flagpool[dev] = ent
        pool, nxt_eager, nxt_graph, free = ent
        half = self.FLAG_BLOCKS // 2
        if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
            if free:
                blk = free.pop()
            elif nxt_graph >= half:
                return None
            else:
                ent[2] = nxt_graph + 1
                blk = nxt_graph
            if self._drawn is not None:
                self._drawn.append((dev, -1 - blk))            # (negative: a flag block, not a ticket slot)
        else:
            ent[1] = (nxt_eager + 1) % half
            blk = half + nxt_eager
        return pool.data_ptr() + 4 * self.FLAG_WORDS * blk

    def ticket_mark(self):
        """Start recording the slots that captured launches draw (one capture at a time)."""
        self._drawn = []

    def ticket_take(self):
        """The slots drawn since ticket_mark(): they belong to the graph(s) just captured."""
        drawn, self._drawn = self._drawn or [], None
        return drawn

    def ticket_release(self, slots):
        """Give back the slots of graphs that will never be replayed again.  The caller has synchronised the device (a replay
        still in flight would otherwise share its counter with the next capture's launch)."""
        for dev, slot in slots or ():
            if slot < 0:
                self._flagpool[dev][3].append(-1 - slot)
            else:
                self._tickets[dev][3].append(slot)

    @property
    def lib(self):
        if self._l is None:
            self._l = _lib.load(self._lib_path)
        return self._l

    # ---- host-only helpers (no GPU needed) ----
    def _x3(self):
        """Flag bit 7 of the GEMM entry points: this fp32 launch may take the three-term split."""
        return 128 if (self.fp32_split and self.precision == "fp32") else 0

    def _all16_flags(self):
        """Flags of a launch whose two operands are both 16-bit in HBM, in the current storage mode."""
        return 1 | 2 | 16 | (32 if self.precision == "fp16s" else 0)

    def igemm_stat_tiles(self, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, all16=False, planes=False):
        """T of the per-tile BatchNorm partial sums the implicit GEMM of the current precision mode writes.  ``all16``: both
        operands of the launch are 16-bit in HBM (the convolution-level launches of the "bf16s" / "fp16s" modes); ``planes``: the
        fp32x3 launch takes its operands already split (ops.Planes)."""
        if planes:
            return self.lib.mmdyn_igemm_stat_tiles_mx(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, 384)
        if self.precision == "fp32":
            if self._x3():
                return self.lib.mmdyn_igemm_stat_tiles_mx(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, 128)
            return self.lib.mmdyn_igemm_stat_tiles(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N)
        if all16 and self.precision in ("bf16s", "fp16s"):
            return self.lib.mmdyn_igemm_stat_tiles_mx(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, self._all16_flags())
        return self.lib.mmdyn_igemm_stat_tiles_bf16(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N)

    def colstats_tiles(self, rows_per_group):
        return self.lib.mmdyn_colstats_tiles(rows_per_group)

    def wgrad_chunks(self, mode, rows, Cd, Cg, planes=(False, False)):
        """``planes``: which of (D, Gt) arrive split (ops.Planes): the cut follows the kernel that serves the launch."""
        # (16-bit storage modes: the all-16-bit weight-gradient kernels keep their own tile rule)
        pre = (256 if planes[0] else 0) | (512 if planes[1] else 0)
        r = self.lib.mmdyn_wgrad_chunks_mx(mode, rows, Cd, Cg, 7 if self.precision in ("bf16s", "fp16s") else (self._x3() | (pre if self._x3() else 0)))
        if r < 0:
            check(r, "mmdyn_wgrad_chunks_mx")
        return r

    # ---- GEMMs ----
    def _slabs(self, like, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, all16=False, planes=False):
        """Workspace for the pieces of the persistent ring kernel's split tiles (fp32 launches not split over K), or None."""
        if planes:
            n = self.lib.mmdyn_igemm_slab_floats_mx(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, 384)
        elif self.precision == "fp32" and self._x3():
            n = self.lib.mmdyn_igemm_slab_floats_mx(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, 128)
        elif self.precision == "fp32":
            n = self.lib.mmdyn_igemm_slab_floats(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N)
        elif all16 and self.precision in ("bf16s", "fp16s"):
            n = self.lib.mmdyn_igemm_slab_floats_mx(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, self._all16_flags())
        else:
            return None
        return torch.empty(n, device=like.device, dtype=torch.float32) if n > 0 else None

    def _mx(self, *fmts):
        """Base flags of the mixed-storage GEMM entry points: bit 0 (16-bit matrix cores), bit 5 when the 16-bit tensors of
        the call are IEEE half.  The storage format has to be the precision mode's: bf16 tensors with "bf16" / "bf16s",
        half tensors with "fp16s"."""
        half = _half(*fmts)
        ok = ("fp16s",) if half else ("bf16", "bf16s")
        if self.precision not in ok:
            raise ValueError(f"mmdyn_hip: {'fp16' if half else 'bf16'} tensors need precision {' / '.join(ok)} "
                             f"(current: {self.precision})")
        return 1 | (32 if half else 0)

    def _check_stat_tiles(self, stats, all16, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, planes=False):
        """The partial-sum buffer must have the tile count the kernel that serves THIS launch writes (it depends on the
        storage types of the operands: igemm_stat_tiles(..., all16)) -- a mismatch would be an out-of-bounds write."""
        if stats is None:
            return
        T = self.igemm_stat_tiles(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, all16=all16, planes=planes)
        if stats.dim() != 4 or tuple(stats.shape) != (G, T, 2, N):
            raise ValueError(f"mmdyn_hip: stats must be [G={G}][T={T}][2][N={N}] for this launch (igemm_stat_tiles with "
                             f"all16={all16}), got {tuple(stats.shape)}")

    def _planes_pair(self, A, Bp, Cin):
        """Both operands of a GEMM arrive split (or neither): their pointers, after the checks the kernel cannot make."""
        if not (isinstance(A, Planes) and isinstance(Bp, Planes)):
            raise TypeError("mmdyn_hip: a GEMM takes both operands as Planes or neither")
        if not self._x3():
            raise ValueError("mmdyn_hip: plane operands belong to the fp32x3 arithmetic (engine precision 'fp32x3')")
        if A.C != Cin or Bp.C != Cin:
            raise ValueError(f"mmdyn_hip: plane rows of {A.C} / {Bp.C} channels for a GEMM over Cin = {Cin}")
        for t in (A.t, Bp.t):
            if not t.is_cuda or not t.is_contiguous():
                raise RuntimeError("mmdyn_hip: plane tensors must be contiguous GPU tensors")
        return A.t.data_ptr(), Bp.t.data_ptr()

    def split_planes(self, x, planes):
        """planes <- the exact three-term bf16 split of the fp32 matrix x ([rows][C] contiguous)."""
        if x.numel() != planes.rows * planes.C:
            raise ValueError("mmdyn_hip: split_planes: shapes differ")
        check(self.lib.mmdyn_split_planes(_ptr(x), planes.t.data_ptr(), planes.rows, planes.C, _stream()), "mmdyn_split_planes")

    def igemm_planes_served(self, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N):
        """True when the fp32x3 launch of this shape can take its operands already split (host query, no GPU needed)."""
        return bool(self._x3()) and self.lib.mmdyn_igemm_planes_served(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N) == 1

    def igemm_nt(self, A, Bp, bias, C, C_act, stats, ws, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, ldc, stride,
                 offset, act, splitk):
        if isinstance(A, Planes) or isinstance(Bp, Planes):
            pa, pb = self._planes_pair(A, Bp, Cin)
            self._check_stat_tiles(stats, False, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, planes=True)
            if ws is None and splitk == 1:
                ws = self._slabs(C, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, planes=True)
            flags, pca = 384, None
            if isinstance(C_act, Planes):            # the activated output as a plane tensor (flag bit 10: DENSE launches)
                if N % C_act.C or C_act.rows * C_act.C != G * Bg * Ho * Wo * N or ldc != N:
                    raise ValueError("mmdyn_hip: igemm_nt: plane output of the wrong shape")
                flags, pca = 384 | 1024 | ((C_act.C // 8) << 16), C_act.t.data_ptr()
            elif C_act is not None:
                pca = _ptr(C_act)
            check(self.lib.mmdyn_igemm_nt_mx(pa, pb, _ptr(bias), _ptr(C), pca, _ptr(stats), _ptr(ws), None, None, None, None,
                                             None, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, ldc, stride, offset, act, splitk, flags,
                                             self._flags(C) if ws is not None and splitk == 1 else None, _stream()), "mmdyn_igemm_nt_mx")
            return
        (pa, a16), (pc, c16), (pca, ca16), (pb, b16) = _aptr(A), _aptr(C), _aptr(C_act), _aptr(Bp)
        self._check_stat_tiles(stats, bool(a16 and b16), mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N)
        if a16 or c16 or ca16 or b16:      # bf16 activation storage / bf16 packed weights: the mixed-storage entry point
            mixed_out = C_act is not None and ca16 and not c16       # fp32 pre-activation + bf16 activated output
            if ((c16 or mixed_out) and splitk != 1) or (C_act is not None and c16 and not ca16):
                raise ValueError("mmdyn_hip: no split-K into a 16-bit output, and a 16-bit C only with a 16-bit C_act")
            if ws is None and splitk == 1:
                ws = self._slabs(A, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, all16=bool(a16 and b16))
            check(self.lib.mmdyn_igemm_nt_mx(pa, pb, _ptr(bias), pc, pca, _ptr(stats), _ptr(ws), None, None, None,
                                             None, None, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, ldc, stride, offset, act,
                                             splitk, self._mx(a16, c16, ca16, b16) | (2 if a16 else 0) | (4 if c16 else 0) |
                                             (16 if b16 else 0) | (64 if mixed_out else 0),
                                             self._flags(C) if ws is not None and splitk == 1 else None, _stream()), "mmdyn_igemm_nt_mx")
            return
        fn = {"fp32": self.lib.mmdyn_igemm_nt, "fp16": self.lib.mmdyn_igemm_nt_f16,
              "fp16s": self.lib.mmdyn_igemm_nt_f16}.get(self.precision, self.lib.mmdyn_igemm_nt_bf16)
        if ws is None and splitk == 1:
            ws = self._slabs(A, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N)
        if self._x3():
            check(self.lib.mmdyn_igemm_nt_mx(pa, pb, _ptr(bias), pc, pca, _ptr(stats), _ptr(ws), None, None, None, None, None,
                                             mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, ldc, stride, offset, act, splitk, 128,
                                             self._flags(C) if ws is not None and splitk == 1 else None, _stream()), "mmdyn_igemm_nt_mx")
            return
        check(fn(pa, _ptr(Bp), _ptr(bias), pc, pca, _ptr(stats), _ptr(ws),
                 mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, ldc, stride, offset, act, splitk, _stream()), "mmdyn_igemm_nt")

    def igemm_nt_dgrad_bn(self, A, Bp, C, stats, y, mean, rstd, gamma, beta, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N,
                          stride, offset):
        if isinstance(A, Planes) or isinstance(Bp, Planes):
            pa, pb = self._planes_pair(A, Bp, Cin)
            self._check_stat_tiles(stats, False, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, planes=True)
            ws = self._slabs(C, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, planes=True)
            check(self.lib.mmdyn_igemm_nt_dgrad_bn(pa, pb, _ptr(C), _ptr(stats), _ptr(y), _ptr(mean), _ptr(rstd), _ptr(gamma),
                                                   _ptr(beta), mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, stride, offset, 4, _ptr(ws),
                                                   self._flags(C) if ws is not None else None, _stream()), "mmdyn_igemm_nt_dgrad_bn")
            return
        (pa, a16), (pc, c16), (py, y16), (pb, b16) = _aptr(A), _aptr(C), _aptr(y), _aptr(Bp)
        self._check_stat_tiles(stats, bool(a16 and b16), mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N)
        if a16 or c16 or y16 or b16:
            ws = self._slabs(A, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, all16=bool(a16 and b16))
            check(self.lib.mmdyn_igemm_nt_mx(pa, pb, None, pc, None, _ptr(stats), _ptr(ws), py, _ptr(mean), _ptr(rstd),
                                             _ptr(gamma), _ptr(beta), mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, N, stride,
                                             offset, ACT_NONE, 1, self._mx(a16, c16, y16, b16) | (2 if a16 else 0) | (4 if c16 else 0) |
                                             (8 if y16 else 0) | (16 if b16 else 0), self._flags(C) if ws is not None else None,
                                             _stream()), "mmdyn_igemm_nt_mx")
            return
        ws = self._slabs(A, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N)
        check(self.lib.mmdyn_igemm_nt_dgrad_bn(pa, _ptr(Bp), pc, _ptr(stats), py, _ptr(mean), _ptr(rstd),
                                               _ptr(gamma), _ptr(beta), mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, stride,
                                               offset, 3 if self._x3() else {"fp32": 0, "fp16": 2, "fp16s": 2}.get(self.precision, 1), _ptr(ws),
                                               self._flags(C) if ws is not None else None, _stream()),
              "mmdyn_igemm_nt_dgrad_bn")

    def igemm_nt_dgrad_act(self, A, Bp, C, u, act, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, stride, offset):
        """C = (A x Bp) * act'(u): input-gradient GEMM with the activation backward in its epilogue."""
        if isinstance(A, Planes) or isinstance(Bp, Planes):
            pa, pb = self._planes_pair(A, Bp, Cin)
            ws = self._slabs(C, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, planes=True)
            check(self.lib.mmdyn_igemm_nt_dgrad_act(pa, pb, _ptr(C), _ptr(u), int(act), mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, stride,
                                                    offset, 384, _ptr(ws), self._flags(C) if ws is not None else None, _stream()),
                  "mmdyn_igemm_nt_dgrad_act")
            return
        (pa, a16), (pc, c16), (pu, u16), (pb, b16) = _aptr(A), _aptr(C), _aptr(u), _aptr(Bp)
        flags = {"fp32": self._x3(), "fp16": 32, "fp16s": 32}.get(self.precision, 1)
        if a16 or c16 or u16 or b16:
            flags = self._mx(a16, c16, u16, b16)
        flags |= (2 if a16 else 0) | (4 if c16 else 0) | (8 if u16 else 0) | (16 if b16 else 0)
        ws = self._slabs(A, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, all16=bool(a16 and b16)) if (flags in (0, 128) or (a16 and b16)) else None
        check(self.lib.mmdyn_igemm_nt_dgrad_act(pa, pb, pc, pu, int(act), mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, stride, offset,
                                                flags, _ptr(ws), self._flags(C) if ws is not None else None, _stream()),
              "mmdyn_igemm_nt_dgrad_act")

    def igemm_nt_grouped(self, A, Bp, bias, C, C_act, u, G, rows, K, N, act):
        """G dense GEMMs of one shape in one launch: A [G*rows][K], Bp [G][N][K], bias [G][N] | None, C [G*rows][N];
        u (same shape as C): C = (A . Bp^T) * act'(u)."""
        (pa, a16), (pc, c16), (pca, ca16), (pb, b16), (pu, u16) = _aptr(A), _aptr(C), _aptr(C_act), _aptr(Bp), _aptr(u)
        flags = {"fp32": self._x3(), "fp16": 33, "fp16s": 33}.get(self.precision, 1)
        if a16 or c16 or ca16 or b16 or u16:
            flags = self._mx(a16, c16, ca16, b16, u16)
        flags |= (2 if a16 else 0) | (4 if c16 else 0) | (8 if u16 else 0) | (16 if b16 else 0)
        check(self.lib.mmdyn_igemm_nt_grouped(pa, pb, _ptr(bias), pc, pca, pu, G, rows, K, N, int(act), flags, _stream()),
              "mmdyn_igemm_nt_grouped")

    def wgrad_tn_grouped(self, D, Gt, partial, G, rows, Cd, Cg, chunks):
        """partial [chunks][G][Cd][Cg] of G weight-gradient problems whose rows follow one another in D / Gt."""
        (pd, d16), (pg, g16) = _aptr(D), _aptr(Gt)
        flags = {"fp32": self._x3(), "fp16": 33, "fp16s": 33}.get(self.precision, 1)
        if d16 or g16:
            flags = self._mx(d16, g16)
        flags |= (2 if d16 else 0) | (4 if g16 else 0)
        check(self.lib.mmdyn_wgrad_tn_grouped(pd, pg, _ptr(partial), G, rows, Cd, Cg, chunks, flags, _stream()),
              "mmdyn_wgrad_tn_grouped")

    def splitk_reduce(self, ws, bias, C, C_act, splitk, rows, N, act):
        check(self.lib.mmdyn_splitk_reduce(_ptr(ws), _ptr(bias), _ptr(C), _ptr(C_act), splitk, rows, N, act,
                                           _stream()), "mmdyn_splitk_reduce")

    def wgrad_tn(self, D, Gt, partial, mode, Bt, Hr, Wr, Cd, Hi, Wi, Cg, stride, offset, chunks):
        if isinstance(D, Planes) or isinstance(Gt, Planes):
            # fp32x3 with one or both operands arriving split (flag bits 8 / 9): that operand goes to the LDS planes as it is
            if not self._x3() or mode != CONV:
                raise ValueError("mmdyn_hip: plane operands belong to the convolution-level weight gradients of the fp32x3 arithmetic")
            flags, ptrs = 128, []
            for k, (t, C) in enumerate(((D, Cd), (Gt, Cg))):
                if isinstance(t, Planes):
                    if t.C != C or not t.t.is_cuda or not t.t.is_contiguous():
                        raise ValueError("mmdyn_hip: wgrad_tn: plane operand of the wrong shape")
                    flags |= 256 << k
                    ptrs.append(t.t.data_ptr())
                else:
                    ptrs.append(_ptr(t))
            check(self.lib.mmdyn_wgrad_tn_mx(ptrs[0], ptrs[1], _ptr(partial), mode, Bt, Hr, Wr, Cd, Hi, Wi, Cg, stride, offset, chunks,
                                             flags, _stream()), "mmdyn_wgrad_tn_mx")
            return
        (pd, d16), (pg, g16) = _aptr(D), _aptr(Gt)
        if d16 or g16:
            check(self.lib.mmdyn_wgrad_tn_mx(pd, pg, _ptr(partial), mode, Bt, Hr, Wr, Cd, Hi, Wi, Cg, stride, offset, chunks,
                                             self._mx(d16, g16) | (2 if d16 else 0) | (4 if g16 else 0), _stream()),
                  "mmdyn_wgrad_tn_mx")
            return
        if self._x3():
            check(self.lib.mmdyn_wgrad_tn_mx(pd, pg, _ptr(partial), mode, Bt, Hr, Wr, Cd, Hi, Wi, Cg, stride, offset, chunks, 128,
                                             _stream()), "mmdyn_wgrad_tn_mx")
            return
        fn = {"fp32": self.lib.mmdyn_wgrad_tn, "fp16": self.lib.mmdyn_wgrad_tn_f16,
              "fp16s": self.lib.mmdyn_wgrad_tn_f16}.get(self.precision, self.lib.mmdyn_wgrad_tn_bf16)
        check(fn(pd, pg, _ptr(partial), mode, Bt, Hr, Wr, Cd, Hi, Wi, Cg, stride, offset, chunks,
                 _stream()), "mmdyn_wgrad_tn")

    def wgrad_reduce(self, partial, canon, chunks, taps, Cd, Cg, cg_canon, perm, beta):
        check(self.lib.mmdyn_wgrad_reduce(_ptr(partial), _ptr(canon), chunks, taps, Cd, Cg, cg_canon, perm,
                                          float(beta), _stream()), "mmdyn_wgrad_reduce")

    # ---- packing / layout ----
    def pack_conv_weight(self, Wc, P, d0, d1, swap):
        pp, p16 = _aptr(P)                 # a 16-bit destination: the GEMM operand type of the 16-bit precision modes
        if p16:
            check(self.lib.mmdyn_pack_conv_weight_b16(_ptr(Wc), pp, d0, d1, int(swap), _half(p16), _stream()),
                  "mmdyn_pack_conv_weight_b16")
            return
        check(self.lib.mmdyn_pack_conv_weight(_ptr(Wc), pp, d0, d1, int(swap), _stream()), "mmdyn_pack_conv_weight")

    def repack2d(self, src, dst, rows_in, cols_in, rows_out, cols_out, mode):
        if _is16(dst):
            return self.repack2d_ld(src, dst, rows_in, cols_in, rows_out, cols_out, cols_out, mode)
        check(self.lib.mmdyn_repack2d(_ptr(src), _ptr(dst), rows_in, cols_in, rows_out, cols_out, mode, _stream()),
              "mmdyn_repack2d")

    def repack2d_ld(self, src, dst, rows_in, cols_in, rows_out, cols_out, ld_out, mode):
        """dst: any fp32 (or 16-bit) view whose first element is the block's top-left corner (row stride ld_out)."""
        _, d16 = _aptr(dst)                # device / contiguity / dtype check
        if d16:
            check(self.lib.mmdyn_repack2d_ld_b16(_ptr(src), dst.data_ptr(), rows_in, cols_in, rows_out, cols_out, ld_out, mode,
                                                 _half(d16), _stream()), "mmdyn_repack2d_ld_b16")
            return
        check(self.lib.mmdyn_repack2d_ld(_ptr(src), dst.data_ptr(), rows_in, cols_in, rows_out, cols_out, ld_out, mode,
                                         _stream()), "mmdyn_repack2d_ld")

    def pack_plan(self, plan_dev, n):
        """plan_dev: uint8 device tensor holding n mmdyn_pack_entry structs (see layers.PackPlan)."""
        check(self.lib.mmdyn_pack_plan(_ptr(plan_dev, torch.uint8), n, _stream()), "mmdyn_pack_plan")

    def im2col_nchw3(self, x, col, Bt, H, W):
        check(self.lib.mmdyn_im2col_nchw3(_ptr(x), _ptr(col), Bt, H, W, _stream()), "mmdyn_im2col_nchw3")

    def col2im_k4(self, col, out, Bt, Hi, Wi, Ho, Wo, C, ldcol, stride, pad, tap_major):
        check(self.lib.mmdyn_col2im_k4(_ptr(col), _ptr(out), Bt, Hi, Wi, Ho, Wo, C, ldcol, stride, pad,
                                       int(tap_major), _stream()), "mmdyn_col2im_k4")

    def tconv_out3_fwd(self, a, w, out, Bt, Hi, Wi):
        pa, a16 = _aptr(a)
        if a16:
            check(self.lib.mmdyn_tconv_out3_fwd_b16(pa, _ptr(w), _ptr(out), Bt, Hi, Wi, _half(a16), _stream()),
                  "mmdyn_tconv_out3_fwd_b16")
            return
        check(self.lib.mmdyn_tconv_out3_fwd(pa, _ptr(w), _ptr(out), Bt, Hi, Wi, _stream()), "mmdyn_tconv_out3_fwd")

    def tconv_out3_bn_fwd(self, y, mean, rstd, gamma, beta, w, out, G, Bg, Hi, Wi):
        """The last decoder layer on the PRE-BatchNorm tensor y: BatchNorm2d + Swish applied while the input tile is staged."""
        py, y16 = _aptr(y)
        check(self.lib.mmdyn_tconv_out3_bn_fwd(py, _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), _ptr(w), _ptr(out), G, Bg, Hi, Wi,
                                               y16, _stream()), "mmdyn_tconv_out3_bn_fwd")

    def tconv_out3_bn_bce(self, y, mean, rstd, gamma, beta, w, logits, logits_group, target, dlogit, loss_slots, slot_of_group,
                          grad_scale, G, Bg, Hi, Wi, mask=None, mask_channels=1, unmasked_slots=None):
        """tconv_out3_bn_fwd with the BCE-with-logits term (sums into loss_slots[slot_of_group[g]], dlogit) in its epilogue; the
        logits are written for group ``logits_group`` only (-1: all groups; ``logits`` None: none)."""
        py, y16 = _aptr(y)
        if len(slot_of_group) != G or tuple(target.shape) != (Bg, 3, 2 * Hi, 2 * Wi):
            raise ValueError("mmdyn_hip: tconv_out3_bn_bce: one slot per group and a [Bg][3][2Hi][2Wi] target")
        if logits is not None and logits.numel() != (G if logits_group < 0 else 1) * Bg * 3 * 4 * Hi * Wi:
            raise ValueError("mmdyn_hip: tconv_out3_bn_bce: logits buffer of the wrong size")
        if dlogit is not None and dlogit.numel() != G * Bg * 3 * 4 * Hi * Wi:
            raise ValueError("mmdyn_hip: tconv_out3_bn_bce: dlogit buffer of the wrong size")
        if mask is not None and tuple(mask.shape) != (Bg, mask_channels, 2 * Hi, 2 * Wi):
            raise ValueError(f"mmdyn_hip: tconv_out3_bn_bce: mask {tuple(mask.shape)} is not [Bg][{mask_channels}][2Hi][2Wi]")
        slots = (ctypes.c_int * G)(*[int(s) for s in slot_of_group])
        check(self.lib.mmdyn_tconv_out3_bn_bce(py, _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), _ptr(w), _ptr(logits),
                                               int(logits_group), _ptr(target), _ptr(mask), int(mask_channels), _ptr(dlogit),
                                               loss_slots.data_ptr(), None if unmasked_slots is None else unmasked_slots.data_ptr(),
                                               ctypes.addressof(slots), float(grad_scale), G, Bg, Hi, Wi, y16, _stream()),
              "mmdyn_tconv_out3_bn_bce")

    def wgrad_out3_bn(self, y, mean, rstd, gamma, beta, Gt, partial, G, Bg, Hr, chunks):
        """Weight gradient of the last decoder layer with swish(BatchNorm(y)) recomputed on the operand fetch."""
        py, y16 = _aptr(y)
        check(self.lib.mmdyn_wgrad_out3_bn(py, _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), _ptr(Gt), _ptr(partial), G, Bg, Hr,
                                           chunks, y16, _stream()), "mmdyn_wgrad_out3_bn")

    def nchw_to_nhwc(self, src, dst, B, C, HW):
        check(self.lib.mmdyn_nchw_to_nhwc(_ptr(src), _ptr(dst), B, C, HW, _stream()), "mmdyn_nchw_to_nhwc")

    def nhwc_to_nchw(self, src, dst, B, C, HW):
        check(self.lib.mmdyn_nhwc_to_nchw(_ptr(src), _ptr(dst), B, C, HW, _stream()), "mmdyn_nhwc_to_nchw")

    # ---- BatchNorm + Swish ----
    def colstats(self, y, partial, G, rows_per_group, C):
        check(self.lib.mmdyn_colstats(_ptr(y), _ptr(partial), G, rows_per_group, C, _stream()), "mmdyn_colstats")

    def bn_finalize(self, partial, mean, rstd, running_mean, running_var, nbt, scratch, G, T, C, rows_per_group,
                    eps, momentum, repeat):
        check(self.lib.mmdyn_bn_finalize(_ptr(partial), _ptr(mean), _ptr(rstd), _ptr(running_mean),
                                         _ptr(running_var), _ptr(nbt, torch.int64), _ptr(scratch, torch.float64),
                                         G, T, C, rows_per_group, eps, momentum, repeat, self._ticket(partial, G * T), _stream()),
              "mmdyn_bn_finalize")

    def bn_swish_fwd(self, y, mean, rstd, gamma, beta, a, G, rows_per_group, C, planes=None):
        """``planes`` (a Planes): the activated tensor is ALSO written already split (``a`` may then be None: planes only)."""
        if planes is not None:
            if planes.rows != G * rows_per_group or planes.C != C:
                raise ValueError("mmdyn_hip: bn_swish_fwd: plane tensor of the wrong shape")
            check(self.lib.mmdyn_bn_swish_fwd_planes(_ptr(y), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), _ptr(a),
                                                     planes.t.data_ptr(), G, rows_per_group, C, _stream()), "mmdyn_bn_swish_fwd_planes")
            return
        (py, y16), (pa, a16) = _aptr(y), _aptr(a)
        if y16 != a16:
            raise TypeError("mmdyn_hip: bn_swish_fwd input and output must share the storage type")
        if y16:
            check(self.lib.mmdyn_bn_swish_fwd_b16(py, _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), pa, G, rows_per_group, C,
                                                  _half(y16), _stream()), "mmdyn_bn_swish_fwd_b16")
            return
        check(self.lib.mmdyn_bn_swish_fwd(py, _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), pa, G, rows_per_group, C,
                                          _stream()), "mmdyn_bn_swish_fwd")

    def bn_swish_bwd_reduce(self, da, y, mean, rstd, gamma, beta, partial, G, rows_per_group, C):
        (pd, d16), (py, y16) = _aptr(da), _aptr(y)
        if d16 != y16:
            raise TypeError("mmdyn_hip: bn_swish_bwd_reduce operands must share the storage type")
        if y16:
            check(self.lib.mmdyn_bn_swish_bwd_reduce_b16(pd, py, _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), _ptr(partial),
                                                         G, rows_per_group, C, _half(y16), _stream()),
                  "mmdyn_bn_swish_bwd_reduce_b16")
            return
        check(self.lib.mmdyn_bn_swish_bwd_reduce(pd, py, _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), _ptr(partial), G,
                                                 rows_per_group, C, _stream()), "mmdyn_bn_swish_bwd_reduce")

    def bn_bwd_finalize(self, partial, sums, dgamma, dbeta, scratch, G, T, C, beta_acc):
        check(self.lib.mmdyn_bn_bwd_finalize(_ptr(partial), _ptr(sums), _ptr(dgamma), _ptr(dbeta),
                                             _ptr(scratch, torch.float64), G, T, C, float(beta_acc), self._ticket(partial, G * T),
                                             _stream()), "mmdyn_bn_bwd_finalize")

    def bn_eval_stats(self, running_mean, running_var, mean, rstd, G, C, eps):
        check(self.lib.mmdyn_bn_eval_stats(_ptr(running_mean), _ptr(running_var), _ptr(mean), _ptr(rstd), G, C, eps,
                                           _stream()), "mmdyn_bn_eval_stats")

    def bn_reduce_partials(self, partial, sums, scratch, G, T, C):
        F64 = torch.float64
        check(self.lib.mmdyn_bn_reduce_partials(_ptr(partial), _ptr(sums, F64), _ptr(scratch, F64), G, T, C, _stream()),
              "mmdyn_bn_reduce_partials")

    def bn_finalize_sums(self, sums, mean, rstd, running_mean, running_var, nbt, G, C, n, eps, momentum, repeat):
        check(self.lib.mmdyn_bn_finalize_sums(_ptr(sums, torch.float64), _ptr(mean), _ptr(rstd), _ptr(running_mean),
                                              _ptr(running_var), _ptr(nbt, torch.int64), G, C, n, eps, momentum, repeat,
                                              _stream()), "mmdyn_bn_finalize_sums")

    def bn_bwd_finalize_sums(self, sums, sums_f, dgamma, dbeta, G, C, sums_scale, beta_acc):
        check(self.lib.mmdyn_bn_bwd_finalize_sums(_ptr(sums, torch.float64), _ptr(sums_f), _ptr(dgamma), _ptr(dbeta), G, C,
                                                  float(sums_scale), float(beta_acc), _stream()),
              "mmdyn_bn_bwd_finalize_sums")

    def bn_swish_bwd_apply(self, da, y, mean, rstd, gamma, beta, sums, dy, G, rows_per_group, C, da_is_du=False, planes=None):
        """``planes`` (a Planes): dL/dy is ALSO written already split (``dy`` may then be None: planes only)."""
        if planes is not None:
            if planes.rows != G * rows_per_group or planes.C != C:
                raise ValueError("mmdyn_hip: bn_swish_bwd_apply: plane tensor of the wrong shape")
            check(self.lib.mmdyn_bn_swish_bwd_apply_planes(_ptr(da), _ptr(y), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), _ptr(sums),
                                                           _ptr(dy), planes.t.data_ptr(), G, rows_per_group, C, int(da_is_du), _stream()),
                  "mmdyn_bn_swish_bwd_apply_planes")
            return
        (pd, d16), (py, y16), (po, o16) = _aptr(da), _aptr(y), _aptr(dy)
        if not (d16 == y16 == o16):
            raise TypeError("mmdyn_hip: bn_swish_bwd_apply tensors must share the storage type")
        if y16:
            check(self.lib.mmdyn_bn_swish_bwd_apply_b16(pd, py, _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), _ptr(sums), po,
                                                        G, rows_per_group, C, int(da_is_du), _half(y16), _stream()),
                  "mmdyn_bn_swish_bwd_apply_b16")
            return
        check(self.lib.mmdyn_bn_swish_bwd_apply(pd, py, _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), _ptr(sums), po, G,
                                                rows_per_group, C, int(da_is_du), _stream()), "mmdyn_bn_swish_bwd_apply")

    # ---- element-wise ----
    def act_fwd(self, u, h, act):
        check(self.lib.mmdyn_act_fwd(_ptr(u), _ptr(h), u.numel(), act, _stream()), "mmdyn_act_fwd")

    def act_bwd(self, dh, u, du, act):
        (pd, d16), (pu, u16), (po, o16) = _aptr(dh), _aptr(u), _aptr(du)
        if not (d16 == u16 == o16):
            raise TypeError("mmdyn_hip: act_bwd tensors must share the storage type")
        if u16:
            check(self.lib.mmdyn_act_bwd_b16(pd, pu, po, u.numel(), act, _half(u16), _stream()), "mmdyn_act_bwd_b16")
            return
        check(self.lib.mmdyn_act_bwd(pd, pu, po, u.numel(), act, _stream()), "mmdyn_act_bwd")

    def dropout_expand(self, h, masks, out, P, B, H, p_drop):
        check(self.lib.mmdyn_dropout_expand(_ptr(h), _ptr(masks, torch.uint8), _ptr(out), P, B, H, p_drop,
                                            _stream()), "mmdyn_dropout_expand")

    def dropout_reduce(self, dout, masks, dh, P, B, H, p_drop, u=None, act=ACT_NONE, planes=None):
        """u (optional, [B][H] fp32): dh = (sum over the passes) * act'(u) -- the activation backward in the same launch.
        planes (optional ops.Planes [B][H]): dh is written there as well, already split."""
        if planes is not None and (planes.rows != B or planes.C != H):
            raise ValueError("mmdyn_hip: dropout_reduce: plane output of the wrong shape")
        check(self.lib.mmdyn_dropout_reduce(_ptr(dout), _ptr(masks, torch.uint8), _ptr(dh), P, B, H, p_drop,
                                            _ptr(u), int(act), None if planes is None else planes.t.data_ptr(), _stream()),
              "mmdyn_dropout_reduce")

    def random_masks(self, masks, p_drop, seed, offset, offset_dev=None):
        check(self.lib.mmdyn_random_masks(_ptr(masks, torch.uint8), masks.numel(), p_drop, seed, offset,
                                          _ptr(offset_dev, torch.int64), _stream()), "mmdyn_random_masks")

    def random_normal(self, out, seed, offset, offset_dev=None):
        check(self.lib.mmdyn_random_normal(_ptr(out), out.numel(), seed, offset, _ptr(offset_dev, torch.int64),
                                           _stream()), "mmdyn_random_normal")

    def counter_add(self, counter, inc):
        check(self.lib.mmdyn_counter_add(_ptr(counter, torch.int64), inc, _stream()), "mmdyn_counter_add")

    def colsum(self, x, out, rows, C, perm, beta):
        scratch = torch.empty(self.lib.mmdyn_colsum_chunks(rows) * C, device=x.device, dtype=torch.float32)
        check(self.lib.mmdyn_colsum(_ptr(x), _ptr(out), _ptr(scratch), rows, C, perm, float(beta), self._ticket(x),
                                    _stream()), "mmdyn_colsum")

    def scale_dev(self, x, s, out):
        check(self.lib.mmdyn_scale_dev(_ptr(x), _ptr(s), _ptr(out), x.numel(), _stream()), "mmdyn_scale_dev")

    def sum_blocks(self, x, out, P, n):
        check(self.lib.mmdyn_sum_blocks(_ptr(x), _ptr(out), P, n, _stream()), "mmdyn_sum_blocks")

    def cast_f32_to_bf16(self, src, dst):
        """dst (bf16, same element count) = RNE(src): the gradient bucket on its way to the all-reduce."""
        check(self.lib.mmdyn_cast_f32_to_bf16(_ptr(src), _ptr(dst, torch.bfloat16), src.numel(), _stream()),
              "mmdyn_cast_f32_to_bf16")

    def copy_many(self, pairs):
        """[(dst, src)]: contiguous GPU tensors of equal byte size, copied by ONE launch per 8 pairs."""
        pairs = [(d, s_) for d, s_ in pairs if d.data_ptr() != s_.data_ptr()]
        for d, s_ in pairs:
            if not (d.is_cuda and s_.is_cuda and d.is_contiguous() and s_.is_contiguous()) or \
                    d.numel() * d.element_size() != s_.numel() * s_.element_size() or d.dtype != s_.dtype:
                raise ValueError("mmdyn_hip: copy_many takes contiguous GPU tensors of equal type and size")
        for i in range(0, len(pairs), 8):
            part = pairs[i:i + 8]
            n = len(part)
            src = (ctypes.c_void_p * n)(*[s_.data_ptr() for _, s_ in part])
            dst = (ctypes.c_void_p * n)(*[d.data_ptr() for d, _ in part])
            nb = (ctypes.c_int64 * n)(*[d.numel() * d.element_size() for d, _ in part])
            check(self.lib.mmdyn_copy_many(ctypes.addressof(src), ctypes.addressof(dst), ctypes.addressof(nb), n, _stream()),
                  "mmdyn_copy_many")

    def cast_bf16_to_f32(self, src, dst):
        check(self.lib.mmdyn_cast_bf16_to_f32(_ptr(src, torch.bfloat16), _ptr(dst), src.numel(), _stream()),
              "mmdyn_cast_bf16_to_f32")

    def linear_small_fwd(self, x, W, b, y, rows, K, N, act):
        check(self.lib.mmdyn_linear_small_fwd(_ptr(x), _ptr(W), _ptr(b), _ptr(y), rows, K, N, act, _stream()),
              "mmdyn_linear_small_fwd")

    def linear_small_bwd(self, dy, x, W, dx, dW, db, rows, K, N, beta):
        check(self.lib.mmdyn_linear_small_bwd(_ptr(dy), _ptr(x), _ptr(W), _ptr(dx), _ptr(dW), _ptr(db), rows, K, N,
                                              float(beta), _stream()), "mmdyn_linear_small_bwd")

    # ---- latent / loss ----
    @staticmethod
    def _passes(passes):
        """passes: list of dicts {'mu': [t|None]*n, 'lv': [...], 'dmu': [...], 'dlv': [...], 'ld': [int]*n}, n<=4;
        tensors may be views (row stride ld) into the fused heads output, so raw data_ptr is used."""
        arr = (PassExperts * MAX_PASSES)()
        for i, p in enumerate(passes):
            for m in range(len(p["ld"])):
                for key in ("mu", "lv", "dmu", "dlv"):
                    t = p.get(key, [None] * MAX_EXPERTS)[m]
                    if t is not None:
                        if not t.is_cuda or t.dtype != torch.float32 or t.stride(-1) != 1:
                            raise ValueError("mmdyn_hip: expert tensors must be fp32 GPU tensors with unit inner stride")
                        getattr(arr[i], key)[m] = t.data_ptr()
                arr[i].ld[m] = int(p["ld"][m])
            for k, t in enumerate(p.get("dz", [])):
                if t is not None:
                    arr[i].dz[k] = _ptr(t)
            for k, t in enumerate(p.get("zdst", [])):      # forward: further [B][L] destinations of the pass's z
                if t is not None:
                    arr[i].zdst[k] = _ptr(t)
            for k, t in enumerate(p.get("zpl", [])):       # ... and plane row blocks (raw pointers into an ops.Planes)
                if t is not None:
                    arr[i].zpl[k] = int(t)
        return arr

    def poe_fwd(self, passes, eps_noise, mu, logvar, z, kl_sum, with_prior, P, B, L):
        arr = self._passes(passes)
        check(self.lib.mmdyn_poe_fwd(ctypes.cast(arr, ctypes.c_void_p), _ptr(eps_noise), _ptr(mu), _ptr(logvar),
                                     _ptr(z), _ptr(kl_sum, torch.float64), int(with_prior), P, B, L, _stream()),
              "mmdyn_poe_fwd")

    def poe_bwd(self, passes, eps_noise, mu, logvar, dz, g_mu, g_lv, kl_scale, with_prior, P, B, L, kl_weight_dev=None):
        """kl_weight_dev: optional 1-element device tensor multiplying kl_scale (the annealed KL weight kept on the GPU)."""
        arr = self._passes(passes)
        check(self.lib.mmdyn_poe_bwd(ctypes.cast(arr, ctypes.c_void_p), _ptr(eps_noise), _ptr(mu), _ptr(logvar),
                                     _ptr(dz), _ptr(g_mu), _ptr(g_lv), float(kl_scale), int(with_prior), P, B, L,
                                     _ptr(kl_weight_dev), _stream()), "mmdyn_poe_bwd")

    def reparam_fwd(self, mu, lv, eps_noise, z, kl_sum, B, L, ld):
        """mu/lv may be column views (row stride ld) of the fused heads output."""
        check(self.lib.mmdyn_reparam_fwd(mu.data_ptr(), lv.data_ptr(), _ptr(eps_noise), _ptr(z),
                                         None if kl_sum is None else kl_sum.data_ptr(), B, L, ld, _stream()),
              "mmdyn_reparam_fwd")

    def reparam_bwd(self, mu, lv, eps_noise, dz, kl_scale, dmu, dlv, B, L, ld):
        check(self.lib.mmdyn_reparam_bwd(mu.data_ptr(), lv.data_ptr(), _ptr(eps_noise), _ptr(dz), float(kl_scale),
                                         dmu.data_ptr(), dlv.data_ptr(), B, L, ld, _stream()), "mmdyn_reparam_bwd")

    def bce_logits(self, logits, target, mask, dlogit, loss_sum, n, chw, hw, grad_scale, mask_channels=1):
        if mask is not None and mask.numel() * chw != n * mask_channels * hw:
            raise ValueError(f"mmdyn_bce_logits: mask of {mask.numel()} elements does not match [B={n // chw}]"
                             f"[{mask_channels}][hw={hw}]")
        check(self.lib.mmdyn_bce_logits(_ptr(logits), _ptr(target), _ptr(mask), _ptr(dlogit),
                                        loss_sum.data_ptr(), n, chw, hw, int(mask_channels), float(grad_scale), _stream()),
              "mmdyn_bce_logits")

    def bce_logits_groups(self, logits, target, dlogit, loss_slots, slot_of_group, n, grad_scale, mask=None, chw=0, hw=0,
                          mask_channels=1, unmasked_slots=None):
        """logits / dlogit: [G*n]; target: [n]; loss_slots: fp64 vector; slot_of_group[g] < 0 = discarded pass.
        mask ([B][mask_channels][hw], n = B*chw): the --mask-loss form; unmasked_slots then also gets the plain sums."""
        G = len(slot_of_group)
        slots = (ctypes.c_int * G)(*[int(s) for s in slot_of_group])
        if mask is None:
            check(self.lib.mmdyn_bce_logits_groups(_ptr(logits), _ptr(target), _ptr(dlogit), loss_slots.data_ptr(),
                                                   ctypes.addressof(slots), G, n, float(grad_scale), _stream()),
                  "mmdyn_bce_logits_groups")
            return
        if chw <= 0 or hw <= 0 or mask.numel() * chw != n * mask_channels * hw:
            raise ValueError(f"mmdyn_bce_logits_groups_masked: mask of {mask.numel()} elements does not match "
                             f"[B={n // max(chw, 1)}][{mask_channels}][hw={hw}]")
        check(self.lib.mmdyn_bce_logits_groups_masked(_ptr(logits), _ptr(target), _ptr(mask), _ptr(dlogit),
                                                      loss_slots.data_ptr(),
                                                      None if unmasked_slots is None else unmasked_slots.data_ptr(),
                                                      ctypes.addressof(slots), G, n, chw, hw, int(mask_channels),
                                                      float(grad_scale), _stream()), "mmdyn_bce_logits_groups_masked")

    def mse(self, r, t, dr, loss_sum, n, grad_scale):
        check(self.lib.mmdyn_mse(_ptr(r), _ptr(t), _ptr(dr), loss_sum.data_ptr(), n, float(grad_scale), _stream()),
              "mmdyn_mse")

    def mse_groups(self, r, t, dr, loss_slots, slot_of_group, n, grad_scale):
        """r / dr: [G*n]; t: [n]; loss_slots: fp64 vector, pass g adds its sum to loss_slots[slot_of_group[g]].  One launch."""
        G = len(slot_of_group)
        slots = (ctypes.c_int * G)(*[int(s) for s in slot_of_group])
        check(self.lib.mmdyn_mse_groups(_ptr(r), _ptr(t), _ptr(dr), loss_slots.data_ptr(), ctypes.addressof(slots), G, n,
                                        float(grad_scale), _stream()), "mmdyn_mse_groups")

    def elbo_assemble(self, bce, mse, kl, loss, partials, P, B, kl_weight, pose_multiplier, kl_weight_dev=None):
        check(self.lib.mmdyn_elbo_assemble(_ptr(bce, torch.float64), _ptr(mse, torch.float64),
                                           _ptr(kl, torch.float64), _ptr(loss), _ptr(partials), P, B,
                                           float(kl_weight), float(pose_multiplier), _ptr(kl_weight_dev), _stream()),
              "mmdyn_elbo_assemble")

    def adam_step(self, p, g, m, v, state, lr, beta1, beta2, eps, grad_scale, guarded=False):
        """guarded: ``state`` has six doubles and a gradient holding inf / NaN skips the step (counted in state[4])."""
        if guarded and state.numel() < 6:
            raise ValueError("mmdyn_hip: the guarded Adam step keeps six doubles of state")
        fn = self.lib.mmdyn_adam_step_guarded if guarded else self.lib.mmdyn_adam_step
        check(fn(_ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(state, torch.float64), p.numel(), lr, beta1, beta2, eps, grad_scale,
                 _stream()), "mmdyn_adam_step")


    def sgd_step(self, p, g, buf, lr, momentum, weight_decay, grad_scale, first):
        check(self.lib.mmdyn_sgd_step(_ptr(p), _ptr(g), _ptr(buf), p.numel(), lr, momentum, weight_decay, grad_scale,
                                      int(first), _stream()), "mmdyn_sgd_step")


    # ---- image decode (uint8 HWC frames in HBM -> float32 CHW) ----
    def resize_plan(self, in_size, out_size, device):
        """(bounds int32 [out][2], coeffs int32 [out][ksize]) of Pillow's 8-bit bilinear resampler, on ``device``."""
        ks = self.lib.mmdyn_resize_ksize(in_size, out_size)
        check(min(ks, 0), "mmdyn_resize_ksize")
        bounds = torch.empty(out_size, 2, dtype=torch.int32)
        coeffs = torch.empty(out_size, ks, dtype=torch.int32)
        check(min(self.lib.mmdyn_resize_plan(in_size, out_size, bounds.data_ptr(), coeffs.data_ptr()), 0),
              "mmdyn_resize_plan")
        return bounds.to(device), coeffs.to(device)

    def resize_u8_to_chw_f32(self, src, index, dst, n_out, Hin, Win, Hout, Wout, xb, xk, yb, yk):
        I32 = torch.int32
        check(self.lib.mmdyn_resize_u8_to_chw_f32(_ptr(src, torch.uint8), _ptr(index, I32), _ptr(dst), n_out, Hin, Win,
                                                  Hout, Wout, _ptr(xb, I32), _ptr(xk, I32), _ptr(yb, I32),
                                                  _ptr(yk, I32), _stream()), "mmdyn_resize_u8_to_chw_f32")


B = HipBackend()


def set_backend(b):
    """Test hook only (tests/emu_backend.py): swap the object that executes the ops."""
    global B
    old, B = B, b
    return old
