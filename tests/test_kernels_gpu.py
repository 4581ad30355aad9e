"""Per-kernel parity on a real MI355X: every C-ABI entry point of libmmdyn_hip.so is run on the GPU and
compared with the CPU contract emulation (tests/emu_backend.py) / the oracle on identical seeded inputs.
fp32 MFMA is an exact fp32 fma chain, so tolerances only cover summation-order differences."""
import math

import pytest
import torch

from mmdyn_hip import ops
from mmdyn_hip.ops import DENSE, CONV, TCONV_S2P1, IM2COL3, TCONV_S1P0
from emu_backend import EmuBackend

pytestmark = pytest.mark.gpu

HIP = ops.HipBackend()
# the LAB build of the same sources (make lab): reads the MMDYN_* experiment variables and contains the opt-in
# direct-fragment kernels.  Tests that FORCE a kernel variant on a small shape swap it in for the duration of the test.
from mmdyn_hip import _lib as _libmod
HIP_LAB = ops.HipBackend(lib_path=_libmod.LAB_LIB_PATH)
EMU = EmuBackend()


@pytest.fixture()
def lab(monkeypatch):
    import sys
    monkeypatch.setattr(sys.modules[__name__], "HIP", HIP_LAB)
    yield HIP_LAB
DEV = "cuda"


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def both(name, args, outs, post=None, tol=2e-5):
    """Call HIP.<name> on GPU copies and EMU.<name> on CPU copies of `args`; compare the tensors named in
    `outs` (indices into args)."""
    gpu_args = [a.to(DEV) if torch.is_tensor(a) else a for a in args]
    cpu_args = [a.clone() if torch.is_tensor(a) else a for a in args]
    getattr(HIP, name)(*gpu_args)
    torch.cuda.synchronize()
    getattr(EMU, name)(*cpu_args)
    for i in outs:
        g, c = gpu_args[i].cpu(), cpu_args[i]
        if post:
            g, c = post(i, g), post(i, c)
        assert torch.isfinite(g).all(), (name, i)
        assert rel(g, c) <= tol, (name, i, rel(g, c))
    return gpu_args, cpu_args


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


IGEMM_CASES = [
    # mode, G, Bg, Hi, Cin, Ho, N, stride, offset
    (DENSE, 1, 256, 1, 64, 1, 32, 1, 0),
    (DENSE, 1, 100, 1, 256, 1, 2048, 1, 0),
    (DENSE, 1, 37, 1, 512, 1, 512, 1, 0),
    (DENSE, 2, 50, 1, 32, 1, 64, 1, 0),
    (CONV, 1, 4, 32, 32, 16, 64, 2, -1),
    (CONV, 2, 3, 16, 64, 8, 128, 2, -1),
    (CONV, 1, 5, 8, 128, 5, 256, 1, 0),
    (CONV, 1, 3, 32, 32, 16, 64, 2, -1),
    (TCONV_S2P1, 1, 4, 8, 128, 16, 64, 1, 0),
    (TCONV_S2P1, 2, 3, 16, 64, 32, 32, 1, 0),
    (TCONV_S2P1, 1, 2, 16, 64, 32, 32, 1, 0),
    (CONV, 4, 64, 8, 128, 5, 256, 1, 0),       # 4 groups, rows not a tile multiple (64*25 = 1600)
    (TCONV_S1P0, 1, 5, 5, 256, 8, 128, 1, 0),
    (TCONV_S1P0, 4, 70, 5, 256, 8, 128, 1, 0),  # 4 groups, samples per group not a tile multiple
    (TCONV_S1P0, 1, 256, 5, 256, 8, 128, 1, 0),
]


@pytest.mark.parametrize("case", IGEMM_CASES)
def test_igemm_nt(case):
    mode, G, Bg, Hi, Cin, Ho, N, stride, offset = case
    Bt = G * Bg
    taps = 16 if mode != DENSE else 1
    A = rnd(Bt * Hi * Hi, Cin, seed=1)
    Bp = rnd(taps, N, Cin, seed=2, scale=0.2)
    bias = rnd(N, seed=3)
    C = torch.zeros(Bt * Ho * Ho, N)
    Ca = torch.zeros(Bt * Ho * Ho, N)
    T = HIP.igemm_stat_tiles(mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N)
    stats = torch.zeros(G, T, 2, N)
    post = lambda i, t: t.sum(1) if i == 5 else t
    # plain output + stats, no bias
    both("igemm_nt", [A, Bp, None, C, None, stats, None, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, N, stride, offset, 0, 1],
         [3, 5], post)
    # bias + swish second output
    both("igemm_nt", [A, Bp, bias, C, Ca, None, None, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, N, stride, offset, 1, 1],
         [3, 4])


@pytest.mark.parametrize("case", [c for c in IGEMM_CASES if c[0] != DENSE][:6] + [IGEMM_CASES[0]])
def test_igemm_dgrad_bn_epilogue(case):
    """Input-gradient GEMM with the BatchNorm+Swish backward of the layer below in its epilogue: C = du and the
    per-tile (sum du, sum du*xhat) partials, against GEMM -> bn_swish_bwd_reduce composed from the emulation."""
    mode, G, Bg, Hi, Cin, Ho, N, stride, offset = case
    Bt = G * Bg
    taps = 16 if mode != DENSE else 1
    A = rnd(Bt * Hi * Hi, Cin, seed=31)
    Bp = rnd(taps, N, Cin, seed=32, scale=0.2)
    rows = Bt * Ho * Ho
    y = rnd(rows, N, seed=33) * 1.5 + 0.2
    mean, rstd = rnd(G, N, seed=34) * 0.3, rnd(G, N, seed=35).abs() + 0.5
    gamma, beta = rnd(N, seed=36) + 1.2, rnd(N, seed=37)
    T = HIP.igemm_stat_tiles(mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N)
    C, stats = torch.zeros(rows, N), torch.zeros(G, T, 2, N)
    post = lambda i, t: t.sum(1) if i == 3 else t
    both("igemm_nt_dgrad_bn", [A, Bp, C, stats, y, mean, rstd, gamma, beta, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, stride,
                               offset], [2, 3], post, tol=5e-5)


@pytest.mark.parametrize("case", [IGEMM_CASES[2], IGEMM_CASES[3], IGEMM_CASES[5], IGEMM_CASES[6], IGEMM_CASES[8],
                                  (TCONV_S2P1, 1, 3, 16, 64, 32, 32, 1, 0), (CONV, 1, 37, 8, 128, 5, 256, 1, 0)])
@pytest.mark.parametrize("act", [1, 2])
def test_igemm_dgrad_act_epilogue(case, act):
    """Input-gradient GEMM with a plain activation backward in its epilogue (C = acc * act'(u)): the wave-specialised
    kernels, the register-staged ones (N = 32) and the patch-resident transposed convolution."""
    mode, G, Bg, Hi, Cin, Ho, N, stride, offset = case
    Bt = G * Bg
    taps = 1 if mode == DENSE else 16
    A = rnd(Bt * Hi * Hi, Cin, seed=41)
    Bp = rnd(taps, N, Cin, seed=42, scale=0.2)
    rows = Bt * Ho * Ho
    u = rnd(rows, N, seed=43) * 2.0
    both("igemm_nt_dgrad_act", [A, Bp, torch.zeros(rows, N), u, act, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, stride, offset], [2],
         tol=5e-5)


def test_single_launch_finalize_equals_two_launches():
    """The "last block finishes" forms (ticket counter; taken for small partial tables only, see ops.HipBackend.ticket_max_work) of
    the BatchNorm finalize, the BatchNorm-backward finalize and the column sum are bit-identical to the two-launch forms, also when
    the same counter slot is used again (the last block resets it)."""
    G, T, C, rpg = 4, 300, 64, 4096
    part = rnd(G, T, 2, C, seed=61).to(DEV)
    outs = []
    rule = HIP.ticket_max_work
    for tick in (False, True, True):
        HIP.force_ticket, HIP.ticket_max_work = tick, 0
        try:
            mean, rstd = torch.zeros(G, C, device=DEV), torch.zeros(G, C, device=DEV)
            rm, rv, nbt = torch.zeros(C, device=DEV), torch.ones(C, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)
            scratch = torch.empty(32, G, 2, C, dtype=torch.float64, device=DEV)
            HIP.bn_finalize(part, mean, rstd, rm, rv, nbt, scratch, G, T, C, rpg, 1e-5, 0.1, 2)
            sums, dg, db = torch.zeros(G, 2, C, device=DEV), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
            HIP.bn_bwd_finalize(part, sums, dg, db, scratch, G, T, C, 0.0)
            x, cs = rnd(1000, 512, seed=62).to(DEV), torch.zeros(512, device=DEV)
            HIP.colsum(x, cs, 1000, 512, 0, 0.0)
            torch.cuda.synchronize()
            outs.append([t.clone() for t in (mean, rstd, rm, rv, nbt, sums, dg, db, cs)])
        finally:
            HIP.force_ticket, HIP.ticket_max_work = False, rule
    for i, (a, b, c) in enumerate(zip(*outs)):
        if i == len(outs[0]) - 1:
            # column sums: without a ticket the FC-level row counts take the single-launch kernel (one block per 32 channels over all
            # rows, round 6) -- another summation order than the chunked last-block-finishes form
            assert torch.equal(b, c) and torch.allclose(a, b, rtol=1e-5, atol=1e-4)
            continue
        assert torch.equal(a, b) and torch.equal(a, c)


def test_dropout_reduce_with_activation_backward():
    P, B, H = 4, 37, 512
    dout, u = rnd(P * B, H, seed=51), rnd(B, H, seed=52) * 2
    masks = (torch.rand(P, B, H, generator=torch.Generator().manual_seed(5)) > 0.1).to(torch.uint8)
    both("dropout_reduce", [dout, masks, torch.zeros(B, H), P, B, H, 0.1, u, 1], [2])


@pytest.fixture()
def d16_tile(request, monkeypatch, lab):
    """Force one wave-tile shape of the direct-fragment fp32 kernels (igemm_d16.hip); without this the small test shapes
    fall below its work threshold and run the LDS-tiled kernel."""
    monkeypatch.setenv("MMDYN_D16_TILE", request.param)
    yield request.param
    monkeypatch.delenv("MMDYN_D16_TILE")


D16_TILES = ["4,4", "8,2", "4,2", "2,2"]


@pytest.fixture()
def mfma16(monkeypatch, lab):
    """The LDS-tiled fp32 kernels on v_mfma_f32_16x16x4_f32 (taken by default only for large launches)."""
    monkeypatch.setenv("MMDYN_IGEMM_M16", "1")
    monkeypatch.setenv("MMDYN_IGEMM_WS", "0")      # (the register-staged kernel: most fp32 shapes go to igemm_ws.hip by default)
    yield
    monkeypatch.delenv("MMDYN_IGEMM_M16")


@pytest.mark.parametrize("case", IGEMM_CASES)
def test_igemm_nt_mfma16(case, mfma16):
    test_igemm_nt(case)


@pytest.fixture()
def regstage(monkeypatch, lab):
    """The register-staged LDS-tiled fp32 kernels (igemm_nt.hip) on shapes the wave-specialised kernels would take."""
    monkeypatch.setenv("MMDYN_IGEMM_WS", "0")
    yield


@pytest.mark.parametrize("case", IGEMM_CASES)
def test_igemm_nt_regstage(case, regstage):
    test_igemm_nt(case)


@pytest.fixture(params=["128,64", "128,128"])
def wsp(request, monkeypatch, lab):
    """The persistent, stream-K-scheduled ring kernel (igemm_wsp.hip) forced onto the small test shapes: with a few hundred
    (tile, K-step) units spread over 256 / 512 resident blocks every block gets one or two K-steps, so EVERY tile is split
    into many pieces and goes through the slab + fix-up path; rows are not tile multiples."""
    monkeypatch.setenv("MMDYN_WSP_TILE", request.param)
    yield request.param


WSP_CASES = [c for c in IGEMM_CASES if c[0] in (DENSE, CONV, TCONV_S2P1) and c[6] % 64 == 0] + [
    (CONV, 4, 40, 16, 64, 8, 128, 2, -1), (TCONV_S2P1, 4, 33, 8, 128, 16, 64, 1, 0), (CONV, 1, 300, 8, 128, 5, 256, 1, 0)]


@pytest.mark.parametrize("case", WSP_CASES)
def test_igemm_nt_persistent(case, wsp):
    if case[6] % int(wsp.split(",")[1]):
        pytest.skip("N is not a multiple of the forced tile width")
    test_igemm_nt(case)


@pytest.mark.parametrize("case", [c for c in WSP_CASES if c[0] != DENSE])
def test_igemm_dgrad_bn_epilogue_persistent(case, wsp):
    if case[6] % int(wsp.split(",")[1]):
        pytest.skip("N is not a multiple of the forced tile width")
    test_igemm_dgrad_bn_epilogue(case)


@pytest.mark.parametrize("case", [WSP_CASES[2], WSP_CASES[4], WSP_CASES[-1], WSP_CASES[-2]])
@pytest.mark.parametrize("act", [1, 2])
def test_igemm_dgrad_act_epilogue_persistent(case, act, wsp):
    if case[6] % int(wsp.split(",")[1]):
        pytest.skip("N is not a multiple of the forced tile width")
    test_igemm_dgrad_act_epilogue(case, act)


@pytest.mark.parametrize("case", [(TCONV_S1P0, 4, 70, 5, 256, 8, 128, 1, 0), (TCONV_S1P0, 2, 200, 5, 256, 8, 128, 1, 0),
                                  (TCONV_S1P0, 1, 5, 5, 256, 8, 128, 1, 0)])
def test_s1p0_persistent(case, lab, monkeypatch):
    """The k4 s1 p0 transposed convolution on the persistent kernel: tiles of one output pixel with 1..16 valid taps, every
    one of them split between blocks (stream-K), sample counts that are not tile multiples; forced on for the small case."""
    monkeypatch.setenv("MMDYN_WSP_MIN_UNITS", "0")
    assert HIP.igemm_stat_tiles(case[0], case[1], case[2], 5, 5, 256, 8, 8, 128) == 64 * ((case[2] + 127) // 128) * 2
    test_igemm_nt(case)


@pytest.mark.parametrize("case", [(CONV, 2, 3, 16, 64, 8, 128, 2, -1), (CONV, 1, 37, 8, 128, 5, 256, 1, 0), (TCONV_S2P1, 2, 5, 8, 128, 16, 64, 1, 0),
                                  (TCONV_S1P0, 2, 70, 5, 256, 8, 128, 1, 0), (CONV, 4, 40, 16, 64, 8, 128, 2, -1)])
def test_igemm_all16_persistent(case, wsp, store16, monkeypatch):
    """Both operands 16-bit in HBM (the convolution-level launches of the storage modes) on the persistent kernel: 64-channel
    K-steps on v_mfma_f32_16x16x32_{bf16,f16}, 16-bit outputs stored four columns per lane, the BatchNorm-backward operand
    loaded likewise; every tile split (forced onto the small shapes)."""
    mode, G, Bg, Hi, Cin, Ho, N, stride, offset = case
    if N % int(wsp.split(",")[1]):
        pytest.skip("N is not a multiple of the forced tile width")
    monkeypatch.setenv("MMDYN_WSP_MIN_UNITS", "0")
    monkeypatch.setenv("MMDYN_WSP_B16", "1")                 # (the product does not route 16-bit launches here: measured slower)
    Bt, rows = G * Bg, G * Bg * Ho * Ho
    A, Bp, bias = bf(rnd(Bt * Hi * Hi, Cin, seed=141)), bf(rnd(16, N, Cin, seed=142, scale=0.2)), rnd(N, seed=143)
    T = HIP.igemm_stat_tiles(mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, all16=True)
    post = lambda i, t: (t.sum(1) if t.dim() == 4 else t.float())
    for c_dtype in (S16, torch.float32):
        C, Ca, stats = torch.zeros(rows, N, dtype=c_dtype), torch.zeros(rows, N, dtype=c_dtype), torch.zeros(G, T, 2, N)
        both("igemm_nt", [A, Bp, None, C, None, stats, None, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, N, stride, offset, 0, 1],
             [3, 5], post, tol=4e-3 if c_dtype == S16 else 2e-5)
        both("igemm_nt", [A, Bp, bias, C, Ca, None, None, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, N, stride, offset, 1, 1],
             [3, 4], post, tol=4e-3 if c_dtype == S16 else 2e-5)
    if mode != TCONV_S1P0:
        y = bf(rnd(rows, N, seed=144) * 1.5 + 0.2)
        mean, rstd = rnd(G, N, seed=145) * 0.3, rnd(G, N, seed=146).abs() + 0.5
        gamma, beta = rnd(N, seed=147) + 1.2, rnd(N, seed=148)
        both("igemm_nt_dgrad_bn", [A, Bp, torch.zeros(rows, N, dtype=S16), torch.zeros(G, T, 2, N), y, mean, rstd, gamma, beta, mode, G,
                                   Bg, Hi, Hi, Cin, Ho, Ho, N, stride, offset], [2, 3], post, tol=4e-3)
        both("igemm_nt_dgrad_act", [A, Bp, torch.zeros(rows, N, dtype=S16), y, 1, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, stride, offset],
             [2], lambda i, t: t.float(), tol=4e-3)


def test_persistent_kernel_whole_tiles_and_split_tiles_agree(lab, monkeypatch):
    """The same launch cut two ways -- 512 tiles over 256 blocks (whole tiles only) and with MMDYN_WSP_UNITS_PER_BLOCK moved off
    a tile multiple (split tiles + fix-up) -- and the one-tile-per-block ring kernel: equal to summation-order error."""
    mode, G, Bg, Hi, Cin, Ho, N, stride, offset = CONV, 4, 128, 16, 64, 8, 128, 2, -1
    Bt = G * Bg
    A, Bp = rnd(Bt * Hi * Hi, Cin, seed=71).to(DEV), rnd(16, N, Cin, seed=72, scale=0.2).to(DEV)
    outs = []
    for env in ({"MMDYN_WSP": "0"}, {"MMDYN_WSP_TILE": "128,128"}, {"MMDYN_WSP_TILE": "128,128", "MMDYN_WSP_UNITS_PER_BLOCK": "19"}):
        for k in ("MMDYN_WSP", "MMDYN_WSP_TILE", "MMDYN_WSP_UNITS_PER_BLOCK"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        T = HIP.igemm_stat_tiles(mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N)
        C, st = torch.zeros(Bt * Ho * Ho, N, device=DEV), torch.zeros(G, T, 2, N, device=DEV)
        HIP.igemm_nt(A, Bp, None, C, None, st, None, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, N, stride, offset, 0, 1)
        torch.cuda.synchronize()
        outs.append((C.cpu(), st.sum(1).cpu()))
    for C, st in outs[1:]:
        assert rel(C, outs[0][0]) < 2e-6 and rel(st, outs[0][1]) < 2e-5


@pytest.mark.parametrize("case", [c for c in IGEMM_CASES if c[0] != DENSE][:6] + [IGEMM_CASES[0]])
def test_igemm_dgrad_bn_epilogue_regstage(case, regstage):
    test_igemm_dgrad_bn_epilogue(case)


@pytest.mark.parametrize("case", [c for c in IGEMM_CASES if c[0] != DENSE][:6] + [IGEMM_CASES[0]])
def test_igemm_dgrad_bn_epilogue_mfma16(case, mfma16):
    test_igemm_dgrad_bn_epilogue(case)


@pytest.mark.parametrize("d16_tile", D16_TILES, indirect=True)
@pytest.mark.parametrize("case", IGEMM_CASES)
def test_igemm_d16(case, d16_tile):
    test_igemm_nt(case)


@pytest.mark.parametrize("d16_tile", D16_TILES, indirect=True)
@pytest.mark.parametrize("case", [c for c in IGEMM_CASES if c[0] != DENSE][:6] + [IGEMM_CASES[0]])
def test_igemm_d16_dgrad_bn_epilogue(case, d16_tile):
    test_igemm_dgrad_bn_epilogue(case)


@pytest.mark.parametrize("d16_tile", D16_TILES, indirect=True)
@pytest.mark.parametrize("rows,K,N,splitk", [(256, 6400, 512, 25), (64, 512, 256, 3), (1024, 6400, 256, 8), (5, 64, 32, 2)])
def test_igemm_d16_splitk(rows, K, N, splitk, d16_tile):
    test_igemm_splitk(rows, K, N, splitk)


@pytest.mark.parametrize("path", ["lds", "d16"])
def test_igemm_full_size_shapes(path, monkeypatch, request):
    """The bs-256 step's own decoder shapes, tile rules unforced -- "lds": the default path (LDS-tiled kernels, 16x16x4 MFMA
    on launches this large), "d16": the opt-in wave-independent kernels (MMDYN_D16=1) -- checked through row-sum /
    column-sum identities instead of an O(M N K) reference: sum_n C[row][n] = A_row . (sum_n B_n) per tap."""
    if path == "d16":
        request.getfixturevalue("lab")
        monkeypatch.setenv("MMDYN_D16", "1")
    for mode, G, Bg, Hi, Cin, Ho, N, stride, offset in [(TCONV_S2P1, 4, 256, 8, 128, 16, 64, 1, 0),
                                                        (CONV, 1, 1024, 16, 64, 8, 128, 2, -1),
                                                        (TCONV_S1P0, 4, 256, 5, 256, 8, 128, 1, 0),
                                                        (TCONV_S2P1, 4, 256, 16, 64, 32, 32, 1, 0)]:
        Bt = G * Bg
        A = rnd(Bt * Hi * Hi, Cin, seed=61).to(DEV)
        Bp = rnd(16, N, Cin, seed=62, scale=0.2).to(DEV)
        C = torch.zeros(Bt * Ho * Ho, N, device=DEV)
        T = HIP.igemm_stat_tiles(mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N)
        stats = torch.zeros(G, T, 2, N, device=DEV)
        HIP.igemm_nt(A, Bp, None, C, None, stats, None, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, N, stride, offset, 0, 1)
        # the same launch with the N columns collapsed into 32 identical columns of sum_n B: every column = row sums of C
        Bs = Bp.sum(1, keepdim=True).expand(16, 32, Cin).contiguous()
        Cs = torch.zeros(Bt * Ho * Ho, 32, device=DEV)
        HIP.igemm_nt(A, Bs, None, Cs, None, None, None, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, 32, 32, stride, offset, 0, 1)
        assert rel(C.sum(1), Cs[:, 0]) < 2e-5, (mode, "row sums")
        # BatchNorm partial sums are the column sums of what was written
        Cg = C.view(G, -1, N)
        assert rel(stats[:, :, 0].sum(1), Cg.sum(1)) < 2e-5 and rel(stats[:, :, 1].sum(1), (Cg * Cg).sum(1)) < 2e-5


@pytest.fixture()
def bf16_mode():
    """bf16 matrix cores: the kernels round fp32 operands to bf16 (RNE) and accumulate in fp32; the emulation does
    the same rounding, so both agree to fp32 summation-order error -- a wrong rounding mode or k-slice layout would
    show up at the 1e-3 level."""
    HIP.precision = EMU.precision = "bf16"
    yield
    HIP.precision = EMU.precision = "fp32"


@pytest.mark.parametrize("case", IGEMM_CASES)
def test_igemm_nt_bf16(case, bf16_mode):
    test_igemm_nt(case)


@pytest.mark.parametrize("rows,K,N,splitk", [(256, 6400, 512, 25), (64, 512, 256, 3), (1024, 6400, 256, 8), (5, 64, 32, 2)])
def test_igemm_splitk(rows, K, N, splitk):
    A, Bp, bias = rnd(rows, K, seed=4), rnd(N, K, seed=5, scale=0.1), rnd(N, seed=6)
    ws = torch.zeros(splitk, rows, N)
    C, Ca = torch.zeros(rows, N), torch.zeros(rows, N)
    ga, ca = both("igemm_nt", [A, Bp, None, C, None, None, ws, DENSE, 1, rows, 1, 1, K, 1, 1, N, N, 1, 0, 0, splitk],
                  [6], lambda i, t: t.sum(0))
    HIP.splitk_reduce(ga[6], bias.to(DEV), ga[3], Ca.to(DEV), splitk, rows, N, 1)
    ref = A @ Bp.t() + bias
    assert rel(ga[3], ref) < 2e-5


WGRAD_CASES = [
    # mode, Bt, Hr, Cd, Hi, Cg, stride, offset, cg_canon, perm
    (DENSE, 300, 1, 512, 1, 512, 1, 0, None, 0),
    (DENSE, 4 * 1024, 1, 32, 1, 64, 1, 0, 48, 0),
    (DENSE, 64, 1, 512, 1, 6400, 1, 0, None, 1),
    (DENSE, 64, 1, 6400, 1, 256, 1, 0, None, 2),
    (CONV, 3, 16, 64, 32, 32, 2, -1, None, 0),
    (CONV, 3, 8, 128, 16, 64, 2, -1, None, 0),
    (CONV, 5, 5, 256, 8, 128, 1, 0, None, 0),
    (CONV, 2, 16, 64, 32, 32, 2, -1, None, 0),
    (CONV, 3, 8, 128, 16, 64, 2, -1, None, 0),
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_wgrad(case):
    mode, Bt, Hr, Cd, Hi, Cg, stride, offset, cgc, perm = case
    rows = Bt * Hr * Hr
    taps = 16 if mode == CONV else 1
    D = rnd(rows, Cd, seed=7)
    Gt = rnd(Bt * Hi * Hi, Cg, seed=8)
    chunks = HIP.wgrad_chunks(mode, rows, Cd, Cg)
    assert chunks % 4 == 0
    partial = torch.zeros(chunks, taps, Cd, Cg)
    ga, ca = both("wgrad_tn", [D, Gt, partial, mode, Bt, Hr, Hr, Cd, Hi, Hi, Cg, stride, offset, chunks], [2],
                  lambda i, t: t.sum(0), tol=5e-5)
    cgc = Cg if cgc is None else cgc
    canon_g = torch.zeros(Cd * cgc * taps, device=DEV)
    canon_c = torch.zeros(Cd * cgc * taps)
    HIP.wgrad_reduce(ga[2], canon_g, chunks, taps, Cd, Cg, cgc, perm, 0.0)
    EMU.wgrad_reduce(ca[2], canon_c, chunks, taps, Cd, Cg, cgc, perm, 0.0)
    assert rel(canon_g, canon_c) < 5e-5
    # accumulate form
    HIP.wgrad_reduce(ga[2], canon_g, chunks, taps, Cd, Cg, cgc, perm, 1.0)
    assert rel(canon_g, 2 * canon_c) < 5e-5


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_wgrad_bf16(case, bf16_mode):
    test_wgrad(case)


S16 = torch.bfloat16          # 16-bit storage type of the running storage test (set by the store16 fixture)


@pytest.fixture(params=["bf16s", "fp16s"])
def store16(request):
    """The two 16-bit storage modes: bf16 ("bf16s") and IEEE half ("fp16s") activations / packed weights; the matrix cores
    run in the same format.  The emulation widens, computes with operands rounded to the format and rounds what it stores."""
    global S16
    S16 = torch.float16 if request.param == "fp16s" else torch.bfloat16
    HIP.precision = EMU.precision = request.param
    yield request.param
    HIP.precision = EMU.precision = "fp32"
    S16 = torch.bfloat16


def bf(t):
    return t.to(S16)


@pytest.mark.parametrize("case", [IGEMM_CASES[1], IGEMM_CASES[4], IGEMM_CASES[6], IGEMM_CASES[9], IGEMM_CASES[13]])
def test_igemm_bf16_storage(case, store16):
    """bf16 activation storage: A and / or C (+ the activated copy, + the BatchNorm-backward operand) are bf16 in HBM.
    The emulation widens, computes in fp32 and rounds the outputs, so agreement is to one bf16 ulp of the outputs
    (a value on a rounding boundary may fall either way after fp32 summation-order differences)."""
    mode, G, Bg, Hi, Cin, Ho, N, stride, offset = case
    Bt = G * Bg
    taps = 16 if mode != DENSE else 1
    A = bf(rnd(Bt * Hi * Hi, Cin, seed=41))
    Bp = rnd(taps, N, Cin, seed=42, scale=0.2)
    bias = rnd(N, seed=43)
    rows = Bt * Ho * Ho
    T = HIP.igemm_stat_tiles(mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N)
    post = lambda i, t: (t.sum(1) if t.dim() == 4 else t.float())
    for c_dtype in (S16, torch.float32):
        C, Ca, stats = torch.zeros(rows, N, dtype=c_dtype), torch.zeros(rows, N, dtype=c_dtype), torch.zeros(G, T, 2, N)
        both("igemm_nt", [A, Bp, None, C, None, stats, None, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, N, stride, offset, 0, 1],
             [3, 5], post, tol=4e-3 if c_dtype == S16 else 2e-5)
        both("igemm_nt", [A, Bp, bias, C, Ca, None, None, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, N, stride, offset, 1, 1],
             [3, 4], post, tol=4e-3 if c_dtype == S16 else 2e-5)
    y = bf(rnd(rows, N, seed=44) * 1.5 + 0.2)
    mean, rstd = rnd(G, N, seed=45) * 0.3, rnd(G, N, seed=46).abs() + 0.5
    gamma, beta = rnd(N, seed=47) + 1.2, rnd(N, seed=48)
    C, stats = torch.zeros(rows, N, dtype=S16), torch.zeros(G, T, 2, N)
    both("igemm_nt_dgrad_bn", [A, Bp, C, stats, y, mean, rstd, gamma, beta, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, stride,
                               offset], [2, 3], post, tol=4e-3)


def test_elementwise_bf16_storage(store16):
    G, rpg, C = 3, 1000, 64
    y, da = bf(rnd(G * rpg, C, seed=51) * 2 + 0.3), bf(rnd(G * rpg, C, seed=52))
    mean, rstd = rnd(G, C, seed=53) * 0.2, rnd(G, C, seed=54).abs() + 0.6
    gamma, beta = rnd(C, seed=55) + 1.5, rnd(C, seed=56)
    f32 = lambda i, t: t.float()
    both("bn_swish_fwd", [y, mean, rstd, gamma, beta, torch.zeros(G * rpg, C, dtype=S16), G, rpg, C], [5], f32,
         tol=4e-3)
    T = HIP.colstats_tiles(rpg)
    ga, ca = both("bn_swish_bwd_reduce", [da, y, mean, rstd, gamma, beta, torch.zeros(G, T, 2, C), G, rpg, C], [6],
                  lambda i, t: t.sum(1), tol=1e-4)
    sums = ca[6].sum(1)
    both("bn_swish_bwd_apply", [da, y, mean, rstd, gamma, beta, sums, torch.zeros(G * rpg, C, dtype=S16), G, rpg,
                                C, False], [7], f32, tol=4e-3)
    both("act_bwd", [da, y, torch.zeros(G * rpg, C, dtype=S16), 1], [2], f32, tol=4e-3)
    a = bf(rnd(2 * 32 * 32, 32, seed=57))
    both("tconv_out3_fwd", [a, rnd(32, 3, 4, 4, seed=58, scale=0.2), torch.zeros(2, 3, 64, 64), 2, 32, 32], [2], tol=2e-5)
    # wgrad with bf16 operands (dense, conv and the im2col mode with a bf16 dense operand)
    # (all tile shapes of the all-bf16 kernel: 128x128, 64x64, 64x32 and 32x64 with two waves per tile, ragged row counts)
    for mode, Bt, Hr, Cd, Hi, Cg, stride, offset in ((DENSE, 300, 1, 512, 1, 512, 1, 0), (CONV, 3, 8, 128, 16, 64, 2, -1),
                                                     (CONV, 5, 5, 256, 8, 128, 1, 0), (CONV, 3, 16, 64, 32, 32, 2, -1),
                                                     (CONV, 2, 8, 32, 16, 64, 2, -1), (DENSE, 77, 1, 256, 1, 6400, 1, 0)):
        rows, taps = Bt * Hr * Hr, 16 if mode == CONV else 1
        D, Gt = bf(rnd(rows, Cd, seed=59)), bf(rnd(Bt * Hi * Hi, Cg, seed=60))
        chunks = HIP.wgrad_chunks(mode, rows, Cd, Cg)
        both("wgrad_tn", [D, Gt, torch.zeros(chunks, taps, Cd, Cg), mode, Bt, Hr, Hr, Cd, Hi, Hi, Cg, stride, offset, chunks],
             [2], lambda i, t: t.sum(0), tol=5e-5)
    D, x = bf(rnd(2 * 1024, 32, seed=61)), rnd(2 * 3 * 64 * 64, seed=62)
    chunks = HIP.wgrad_chunks(IM2COL3, 2 * 1024, 32, 64)
    both("wgrad_tn", [D, x, torch.zeros(chunks, 1, 32, 64), IM2COL3, 2, 32, 32, 32, 64, 64, 64, 1, 0, chunks], [2],
         lambda i, t: t.sum(0), tol=5e-5)


def test_bf16_differs_from_fp32_by_bf16_rounding_only():
    """Sanity on the size of the effect: relative error of the bf16 product vs the fp32 one is ~2^-9 per operand."""
    A, Bp = rnd(512, 256, seed=21), rnd(128, 256, seed=22, scale=0.2)
    outs = []
    for prec in ("fp32", "bf16"):
        HIP.precision = prec
        C = torch.zeros(512, 128, device=DEV)
        HIP.igemm_nt(A.to(DEV), Bp.to(DEV), None, C, None, None, None, DENSE, 1, 512, 1, 1, 256, 1, 1, 128, 128, 1, 0, 0, 1)
        outs.append(C.cpu())
    HIP.precision = "fp32"
    assert rel(outs[0], A @ Bp.t()) < 2e-5
    assert 1e-4 < rel(outs[1], outs[0]) < 1e-2


def test_pack_and_layout_kernels():
    W = rnd(64, 32, 4, 4, seed=9)
    for swap in (0, 1):
        both("pack_conv_weight", [W, torch.zeros(16 * 64 * 32), 64, 32, swap], [1], tol=0)
    for mode, ri, ci, ro, co in [(0, 32, 48, 32, 64), (1, 32, 48, 64, 32), (2, 8, 6400, 8, 6400), (3, 6400, 8, 6400, 8),
                                 (4, 8, 6400, 6400, 8), (5, 6400, 8, 8, 6400), (3, 6400, 1, 6400, 1)]:
        both("repack2d", [rnd(ri, ci, seed=10), torch.zeros(ro * co), ri, ci, ro, co, mode], [1], tol=0)
    x = rnd(3, 3, 64, 64, seed=11)
    both("im2col_nchw3", [x, torch.zeros(3 * 1024 * 64), 3, 64, 64], [1], tol=0)
    both("col2im_k4", [rnd(3 * 25, 2048, seed=12), torch.zeros(3 * 64 * 128), 3, 5, 5, 8, 8, 128, 2048, 1, 0, 1], [1])
    both("col2im_k4", [rnd(2 * 1024, 64, seed=13), torch.zeros(2 * 3 * 64 * 64), 2, 32, 32, 64, 64, 3, 64, 2, 1, 0], [1])
    both("nchw_to_nhwc", [x, torch.zeros(x.numel()), 3, 3, 4096], [1], tol=0)
    both("nhwc_to_nchw", [x, torch.zeros(x.numel()), 3, 3, 4096], [1], tol=0)


@pytest.mark.parametrize("G,rpg,C", [(1, 1600, 256), (4, 700, 128), (2, 5000, 64), (3, 4096, 32), (1, 100, 256)])
def test_batchnorm_kernels(G, rpg, C):
    y = rnd(G * rpg, C, seed=14) * 2 + 0.3
    da = rnd(G * rpg, C, seed=15)
    gamma, beta = rnd(C, seed=16) + 1.5, rnd(C, seed=17)
    T = HIP.colstats_tiles(rpg)
    partial = torch.zeros(G, T, 2, C)
    ga, ca = both("colstats", [y, partial, G, rpg, C], [1], lambda i, t: t.sum(1))
    mean, rstd = torch.zeros(G, C), torch.zeros(G, C)
    rm, rv, nbt = rnd(C, seed=18), rnd(C, seed=19).abs() + 0.5, torch.zeros((), dtype=torch.long)
    scratch = torch.zeros(32, G, 2, C, dtype=torch.float64)
    args = [partial, mean, rstd, rm, rv, nbt, scratch, G, T, C, rpg, 1e-5, 0.1, 2]
    g_args = [a.to(DEV) if torch.is_tensor(a) else a for a in args]
    g_args[0] = ga[1]
    HIP.bn_finalize(*g_args)
    c_args = [a.clone() if torch.is_tensor(a) else a for a in args]
    c_args[0] = ca[1]
    EMU.bn_finalize(*c_args)
    for i in (1, 2, 3, 4):
        assert rel(g_args[i], c_args[i]) < 1e-5, i
    assert int(g_args[5].cpu()) == 2 * G == int(c_args[5])
    mean, rstd = c_args[1], c_args[2]
    a = torch.zeros(G * rpg, C)
    both("bn_swish_fwd", [y, mean, rstd, gamma, beta, a, G, rpg, C], [5])
    ga, ca = both("bn_swish_bwd_reduce", [da, y, mean, rstd, gamma, beta, partial, G, rpg, C], [6],
                  lambda i, t: t.sum(1), tol=1e-4)
    sums, dg, db = torch.zeros(G, 2, C), torch.zeros(C), torch.zeros(C)
    g2 = [ga[6], sums.to(DEV), dg.to(DEV), db.to(DEV), scratch.to(DEV), G, T, C, 0.0]
    HIP.bn_bwd_finalize(*g2)
    c2 = [ca[6], sums, dg, db, scratch, G, T, C, 0.0]
    EMU.bn_bwd_finalize(*c2)
    for i in (1, 2, 3):
        assert rel(g2[i], c2[i]) < 1e-4, i
    both("bn_swish_bwd_apply", [da, y, mean, rstd, gamma, beta, sums, torch.zeros(G * rpg, C), G, rpg, C], [7], tol=1e-4)
    # the synchronised-BatchNorm pieces: reduce_partials -> (all-reduce) -> finalize_sums must equal bn_finalize on one
    # rank, and with the sums doubled / n doubled ("two identical ranks") the statistics must not move
    s64 = torch.zeros(G, 2, C, dtype=torch.float64, device=DEV)
    HIP.bn_reduce_partials(g_args[0], s64, scratch.to(DEV), G, T, C)
    assert rel(s64, g_args[0].double().sum(1)) < 1e-6
    for world in (1, 2):
        m2, r2 = torch.zeros(G, C, device=DEV), torch.zeros(G, C, device=DEV)
        HIP.bn_finalize_sums(s64 * world, m2, r2, None, None, None, G, C, rpg * world, 1e-5, 0.1, 1)
        assert rel(m2, g_args[1]) < 1e-6 and rel(r2, g_args[2]) < 1e-5
    b64 = torch.zeros(G, 2, C, dtype=torch.float64, device=DEV)
    HIP.bn_reduce_partials(ga[6], b64, scratch.to(DEV), G, T, C)
    sf, dg2, db2 = torch.zeros(G, 2, C, device=DEV), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    HIP.bn_bwd_finalize_sums(b64, None, dg2, db2, G, C, 1.0, 0.0)
    HIP.bn_bwd_finalize_sums(b64 * 2, sf, None, None, G, C, 0.5, 0.0)
    assert rel(dg2, g2[2]) < 1e-5 and rel(db2, g2[3]) < 1e-5 and rel(sf, g2[1]) < 1e-5


def test_elementwise_kernels():
    u, dh = rnd(1000, 513, seed=20) * 4, rnd(1000, 513, seed=21)
    for act in (0, 1, 2):
        both("act_fwd", [u, torch.zeros_like(u), act], [1])
        both("act_bwd", [dh, u, torch.zeros_like(u), act], [2])
    h = rnd(8, 512, seed=22)
    masks = (torch.rand(4, 8, 512) > 0.1).to(torch.uint8)
    both("dropout_expand", [h, masks, torch.zeros(4, 8, 512), 4, 8, 512, 0.1], [2])
    both("dropout_reduce", [rnd(4, 8, 512, seed=23), masks, torch.zeros(8, 512), 4, 8, 512, 0.1], [2])
    both("colsum", [rnd(300, 512, seed=24), torch.zeros(512), 300, 512, 0, 0.0], [1])
    both("colsum", [rnd(64, 6400, seed=25), torch.zeros(6400), 64, 6400, 2, 0.0], [1])
    both("sum_blocks", [rnd(4, 999, seed=26), torch.zeros(999), 4, 999], [1])
    for rows, K, N, act in [(33, 7, 512, 2), (33, 512, 7, 0)]:
        x, W, b = rnd(rows, K, seed=27), rnd(N, K, seed=28), rnd(N, seed=29)
        both("linear_small_fwd", [x, W, b, torch.zeros(rows, N), rows, K, N, act], [3])
        both("linear_small_bwd", [rnd(rows, N, seed=30), x, W, torch.zeros(rows, K), torch.zeros(N, K), torch.zeros(N),
                                  rows, K, N, 0.0], [3, 4, 5])


def test_adam_matches_torch():
    n = 100003
    p0, g = rnd(n, seed=31), rnd(n, seed=32) * 0.01
    p = p0.clone().to(DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    state = torch.zeros(3, dtype=torch.float64, device=DEV)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3)
    for step in range(3):
        HIP.adam_step(p, (g * (step + 1)).to(DEV), m, v, state, 1e-3, 0.9, 0.999, 1e-8, 1.0)
        ref.grad = g * (step + 1)
        opt.step()
    assert float(state[0].cpu()) == 3.0
    torch.testing.assert_close(p.cpu(), ref.detach(), rtol=1e-6, atol=1e-7)


def test_guarded_adam_skips_a_step_whose_gradient_overflowed():
    """mmdyn_adam_step_guarded (the fp16 modes): a gradient buffer holding an inf or a NaN leaves parameters, moments and the
    step count untouched and is counted; finite gradients step exactly like mmdyn_adam_step."""
    n = 100003
    p0, g = rnd(n, seed=33), rnd(n, seed=34) * 0.01
    p, pr = p0.clone().to(DEV), p0.clone().to(DEV)
    m, v, mr, vr = (torch.zeros(n, device=DEV) for _ in range(4))
    state, ref_state = torch.zeros(6, dtype=torch.float64, device=DEV), torch.zeros(3, dtype=torch.float64, device=DEV)
    bad = g.clone()
    bad[n - 2] = float("inf")
    nan = g.clone()
    nan[7] = float("nan")
    for gg, finite in ((g, True), (bad, False), (g * 2, True), (nan, False), (g * 3, True)):
        before = (p.clone(), m.clone(), v.clone())
        HIP.adam_step(p, gg.to(DEV), m, v, state, 1e-3, 0.9, 0.999, 1e-8, 1.0, guarded=True)
        if finite:
            HIP.adam_step(pr, gg.to(DEV), mr, vr, ref_state, 1e-3, 0.9, 0.999, 1e-8, 1.0)
            assert torch.equal(p, pr) and torch.equal(m, mr) and torch.equal(v, vr)
        else:
            assert torch.equal(p, before[0]) and torch.equal(m, before[1]) and torch.equal(v, before[2])
    assert float(state[0]) == 3.0 and float(state[4]) == 2.0 and float(ref_state[0]) == 3.0
    with pytest.raises(ValueError):
        HIP.adam_step(p, g.to(DEV), m, v, ref_state, 1e-3, 0.9, 0.999, 1e-8, 1.0, guarded=True)


def test_latent_and_loss_kernels():
    B, L, P = 5, 256, 3
    heads = [rnd(B, 2 * L, seed=40 + i) for i in range(3)]
    dheads_g = [torch.zeros(B, 2 * L, device=DEV) for _ in range(3)]
    dheads_c = [torch.zeros(B, 2 * L) for _ in range(3)]
    eps = torch.randn(P, B, L, generator=torch.Generator().manual_seed(5))
    subsets = [(1, 1, 0), (0, 1, 1), (1, 1, 1)]

    def passes(hs, ds):
        out = []
        for s in subsets:
            out.append({"mu": [hs[m][:, :L] if s[m] else None for m in range(3)],
                        "lv": [hs[m][:, L:] if s[m] else None for m in range(3)],
                        "dmu": [ds[m][:, :L] if s[m] else None for m in range(3)],
                        "dlv": [ds[m][:, L:] if s[m] else None for m in range(3)], "ld": [2 * L] * 3})
        return out

    hg = [h.to(DEV) for h in heads]
    mu_g, lv_g, z_g = (torch.zeros(P, B, L, device=DEV) for _ in range(3))
    kl_g = torch.zeros(P, dtype=torch.float64, device=DEV)
    HIP.poe_fwd(passes(hg, dheads_g), eps.to(DEV), mu_g, lv_g, z_g, kl_g, 1, P, B, L)
    mu_c, lv_c, z_c = (torch.zeros(P, B, L) for _ in range(3))
    kl_c = torch.zeros(P, dtype=torch.float64)
    EMU.poe_fwd(passes(heads, dheads_c), eps, mu_c, lv_c, z_c, kl_c, 1, P, B, L)
    for a, b in ((mu_g, mu_c), (lv_g, lv_c), (z_g, z_c), (kl_g, kl_c)):
        assert rel(a, b) < 1e-5
    dz = torch.randn(P, B, L, generator=torch.Generator().manual_seed(6))
    # each pass writes its own gradient rows; run pass by pass so shared expert buffers are compared per pass
    for p in range(P):
        for d in dheads_g + dheads_c:
            d.zero_()
        HIP.poe_bwd(passes(hg, dheads_g)[p:p + 1], eps[p:p + 1].to(DEV), mu_g[p:p + 1].contiguous(),
                    lv_g[p:p + 1].contiguous(), dz[p:p + 1].to(DEV), None, None, 0.02 / B, 1, 1, B, L)
        EMU.poe_bwd(passes(heads, dheads_c)[p:p + 1], eps[p:p + 1], mu_c[p:p + 1], lv_c[p:p + 1], dz[p:p + 1], None, None,
                    0.02 / B, 1, 1, B, L)
        for m in range(3):
            if subsets[p][m]:
                assert rel(dheads_g[m], dheads_c[m]) < 1e-4, (p, m)
    # single-expert reparametrisation + KL
    z1g, kl1g = torch.zeros(B, L, device=DEV), torch.zeros(1, dtype=torch.float64, device=DEV)
    HIP.reparam_fwd(hg[0][:, :L], hg[0][:, L:], eps[0].to(DEV), z1g, kl1g, B, L, 2 * L)
    z1c, kl1c = torch.zeros(B, L), torch.zeros(1, dtype=torch.float64)
    EMU.reparam_fwd(heads[0][:, :L], heads[0][:, L:], eps[0], z1c, kl1c, B, L, 2 * L)
    assert rel(z1g, z1c) < 1e-5 and rel(kl1g, kl1c) < 1e-6
    HIP.reparam_bwd(hg[0][:, :L], hg[0][:, L:], eps[0].to(DEV), dz[0].to(DEV), 0.3, dheads_g[0][:, :L], dheads_g[0][:, L:],
                    B, L, 2 * L)
    EMU.reparam_bwd(heads[0][:, :L], heads[0][:, L:], eps[0], dz[0], 0.3, dheads_c[0][:, :L], dheads_c[0][:, L:], B, L, 2 * L)
    assert rel(dheads_g[0], dheads_c[0]) < 1e-5
    # reconstruction terms
    n = 3 * 3 * 64 * 64
    logits, target = rnd(3, 3, 64, 64, seed=50) * 6, torch.rand(3, 3, 64, 64)
    mask = (torch.rand(3, 1, 64, 64) > 0.5).float()
    for mk in (None, mask):
        lg, lc = torch.zeros(1, dtype=torch.float64, device=DEV), torch.zeros(1, dtype=torch.float64)
        dg, dc = torch.zeros(n, device=DEV), torch.zeros(n)
        HIP.bce_logits(logits.to(DEV), target.to(DEV), None if mk is None else mk.to(DEV), dg, lg, n, 3 * 4096, 4096, 0.25)
        EMU.bce_logits(logits, target, mk, dc, lc, n, 3 * 4096, 4096, 0.25)
        assert rel(lg, lc) < 1e-6 and rel(dg, dc) < 1e-5
    r, t = rnd(9, 7, seed=51), rnd(9, 7, seed=52)
    lg, lc = torch.zeros(1, dtype=torch.float64, device=DEV), torch.zeros(1, dtype=torch.float64)
    dg, dc = torch.zeros(63, device=DEV), torch.zeros(63)
    HIP.mse(r.to(DEV), t.to(DEV), dg, lg, 63, 0.5)
    EMU.mse(r, t, dc, lc, 63, 0.5)
    assert rel(lg, lc) < 1e-6 and rel(dg, dc) < 1e-6
    acc = torch.tensor([[1.0, 2.0, 3.0], [0.5, 0.0, 0.25], [7.0, 8.0, 9.0]], dtype=torch.float64)
    loss_g, part_g = torch.zeros(1, device=DEV), torch.zeros(3, device=DEV)
    HIP.elbo_assemble(acc[0].to(DEV), acc[1].to(DEV), acc[2].to(DEV), loss_g, part_g, 3, 4, 0.02, 1000.0)
    want = (acc[0] + 1000.0 * acc[1] + 0.02 * acc[2]) / 4
    assert rel(part_g, want) < 1e-6 and abs(float(loss_g.cpu()) - float(want.sum())) < 1e-3


def test_random_kernels_statistics():
    n = 1 << 20
    m = torch.zeros(n, dtype=torch.uint8, device=DEV)
    HIP.random_masks(m, 0.1, 1234, 0)
    keep = float(m.float().mean().cpu())
    assert abs(keep - 0.9) < 3e-3
    z = torch.zeros(n, device=DEV)
    HIP.random_normal(z, 99, 0)
    assert abs(float(z.mean().cpu())) < 5e-3 and abs(float(z.std().cpu()) - 1.0) < 5e-3
    z2 = torch.zeros(n, device=DEV)
    HIP.random_normal(z2, 99, n // 4)
    assert not torch.equal(z, z2)
    HIP.random_normal(z2, 99, 0)
    assert torch.equal(z, z2)


def test_cpu_tensor_is_rejected_loudly():
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        HIP.act_fwd(torch.zeros(8), torch.zeros(8), 1)


@pytest.mark.parametrize("Bt", [1, 5])
def test_tconv_out3_direct(Bt):
    a = rnd(Bt * 32 * 32, 32, seed=60)
    w = rnd(32, 3, 4, 4, seed=61)
    both("tconv_out3_fwd", [a, w, torch.zeros(Bt, 3, 64, 64), Bt, 32, 32], [2])


@pytest.mark.parametrize("Bt", [2, 7])
def test_im2col_on_the_fly_modes(Bt):
    x = rnd(Bt, 3, 64, 64, seed=70)
    Bp = rnd(1, 32, 64, seed=71, scale=0.2)
    C, Ca = torch.zeros(Bt * 1024, 32), torch.zeros(Bt * 1024, 32)
    both("igemm_nt", [x, Bp, None, C, Ca, None, None, IM2COL3, 1, Bt, 64, 64, 64, 32, 32, 32, 32, 1, 0, 1, 1], [3, 4])
    D = rnd(Bt * 1024, 32, seed=72)
    chunks = HIP.wgrad_chunks(IM2COL3, Bt * 1024, 32, 64)
    partial = torch.zeros(chunks, 1, 32, 64)
    both("wgrad_tn", [D, x, partial, IM2COL3, Bt, 32, 32, 32, 64, 64, 64, 1, 0, chunks], [2], lambda i, t: t.sum(0), tol=5e-5)


@pytest.mark.parametrize("G,Bg", [(1, 2), (2, 3), (4, 8)])
def test_conv3_direct_kernels(G, Bg):
    """The 3-channel layers' own kernels (conv3.hip) behind the IM2COL3 geometry: forward with per-tile BatchNorm
    partials, input gradient with the BatchNorm+Swish backward epilogue, weight gradient -- grouped batches."""
    Bt = G * Bg
    x = rnd(Bt, 3, 64, 64, seed=90)
    Bp = rnd(1, 32, 64, seed=91, scale=0.2)
    Bp[:, :, 48:] = 0
    rows = Bt * 1024
    T = HIP.igemm_stat_tiles(IM2COL3, G, Bg, 64, 64, 64, 32, 32, 32)
    assert T == Bg * 8
    post = lambda i, t: t.sum(1) if t.dim() == 4 else t
    C, stats = torch.zeros(rows, 32), torch.zeros(G, T, 2, 32)
    both("igemm_nt", [x, Bp, None, C, None, stats, None, IM2COL3, G, Bg, 64, 64, 64, 32, 32, 32, 32, 1, 0, 0, 1], [3, 5], post)
    y = rnd(rows, 32, seed=92) * 1.5 + 0.2
    mean, rstd = rnd(G, 32, seed=93) * 0.3, rnd(G, 32, seed=94).abs() + 0.5
    gamma, beta = rnd(32, seed=95) + 1.2, rnd(32, seed=96)
    both("igemm_nt_dgrad_bn", [x, Bp, C, stats, y, mean, rstd, gamma, beta, IM2COL3, G, Bg, 64, 64, 64, 32, 32, 32, 1, 0],
         [2, 3], post, tol=5e-5)
    D = rnd(rows, 32, seed=97)
    for chunks in (HIP.wgrad_chunks(IM2COL3, rows, 32, 64), 4, 12):
        both("wgrad_tn", [D, x, torch.zeros(chunks, 1, 32, 64), IM2COL3, Bt, 32, 32, 32, 64, 64, 64, 1, 0, chunks], [2],
             lambda i, t: t.sum(0), tol=5e-5)


def test_bf16_packed_weights(store16):
    """bf16 precision modes pack the GEMM operands straight to bf16 (pack kernels with a bf16 destination) and the
    implicit GEMM reads them as such: same products as rounding fp32 packed weights inside the kernel."""
    f32 = lambda i, t: t.float()
    Wc = rnd(128, 64, 4, 4, seed=101, scale=0.2)
    for swap in (0, 1):
        both("pack_conv_weight", [Wc, torch.zeros(16, 64 if swap else 128, 128 if swap else 64, dtype=S16), 128, 64, swap],
             [1], f32, tol=0.0)
    W = rnd(40, 500, seed=102)
    out = torch.zeros(64, 512, dtype=S16)
    both("repack2d_ld", [W, out.view(-1)[512 * 8:], 40, 500, 40, 512, 512, 0], [1], f32, tol=0.0)
    for mode, G, Bg, Hi, Cin, Ho, N, stride, offset in ((CONV, 2, 3, 16, 64, 8, 128, 2, -1), (TCONV_S2P1, 1, 5, 8, 128, 16, 64, 1, 0),
                                                       (DENSE, 1, 300, 1, 512, 1, 512, 1, 0), (TCONV_S1P0, 2, 70, 5, 256, 8, 128, 1, 0)):
        Bt, taps = G * Bg, 16 if mode != DENSE else 1
        Bp = bf(rnd(taps, N, Cin, seed=103, scale=0.2))
        for a_dtype in (torch.float32, S16):
            A = rnd(Bt * Hi * Hi, Cin, seed=104)
            A = bf(A) if a_dtype == S16 else A
            C = torch.zeros(Bt * Ho * Ho, N, dtype=a_dtype)
            both("igemm_nt", [A, Bp, None, C, None, None, None, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, N, stride, offset, 0, 1],
                 [3], f32, tol=4e-3 if a_dtype == S16 else 2e-5)
    # the 3-channel layer's kernel takes the bf16 [32][64] operand too
    x = rnd(3, 3, 64, 64, seed=105)
    Bp = bf(rnd(1, 32, 64, seed=106, scale=0.2))
    both("igemm_nt", [x, Bp, None, torch.zeros(3 * 1024, 32), torch.zeros(3 * 1024, 32), None, None, IM2COL3, 1, 3, 64, 64, 64, 32,
                      32, 32, 32, 1, 0, 1, 1], [3, 4])


def test_igemm_bf16_mixed_output_and_one_pixel_walk(monkeypatch, lab, store16):
    """(i) A Linear layer whose activated output alone is bf16 (fp32 pre-activation for the backward, bf16 operand for the
    convolution behind it: storage flag bit 6).  (ii) The k4 s1 p0 transposed convolution on all-bf16 operands in its three
    block walks -- pairs of pixels (default below 2048 blocks), one pixel per block (default from 2048 blocks on, forced
    here on a small launch too) -- with bias, BatchNorm partial sums and ragged sample counts."""
    f32 = lambda i, t: t.float()
    try:
        for rows, K, N in ((300, 256, 6400), (37, 288, 640)):
            A, Bp, bias = rnd(rows, K, seed=120), bf(rnd(1, N, K, seed=121, scale=0.2)), rnd(N, seed=122)
            both("igemm_nt", [A, Bp, bias, torch.zeros(rows, N), torch.zeros(rows, N, dtype=S16), None, None, DENSE, 1,
                              rows, 1, 1, K, 1, 1, N, N, 1, 0, 1, 1], [3, 4], f32, tol=4e-3)
        with pytest.raises(ValueError):      # a bf16 pre-activation next to an fp32 activated output is not a storage layout
            HIP.igemm_nt(A.to(DEV), Bp.to(DEV), None, torch.zeros(rows, N, dtype=S16, device=DEV),
                         torch.zeros(rows, N, device=DEV), None, None, DENSE, 1, rows, 1, 1, K, 1, 1, N, N, 1, 0, 1, 1)
        post = lambda i, t: t.float().sum(1) if t.dim() == 4 else t.float()
        for forced, G, Bg in ((None, 4, 256), ("2", 4, 256), ("4", 2, 70), (None, 2, 70), ("4", 1, 5)):
            if forced is None:
                monkeypatch.delenv("MMDYN_S1P0_SPLIT", raising=False)
            else:
                monkeypatch.setenv("MMDYN_S1P0_SPLIT", forced)
            Bt = G * Bg
            A, Bp, bias = bf(rnd(Bt * 25, 256, seed=123)), bf(rnd(16, 128, 256, seed=124, scale=0.1)), rnd(128, seed=125)
            T = HIP.igemm_stat_tiles(TCONV_S1P0, G, Bg, 5, 5, 256, 8, 8, 128, all16=True)
            both("igemm_nt", [A, Bp, bias, torch.zeros(Bt * 64, 128, dtype=S16), None, torch.zeros(G, T, 2, 128), None,
                              TCONV_S1P0, G, Bg, 5, 5, 256, 8, 8, 128, 128, 1, 0, 0, 1], [3, 5], post, tol=4e-3)
    finally:
        monkeypatch.delenv("MMDYN_S1P0_SPLIT", raising=False)


@pytest.mark.parametrize("H,Cin,G,Bg", [(16, 64, 2, 3), (16, 64, 1, 1), (16, 64, 4, 37), (32, 32, 2, 3), (32, 32, 1, 5), (64, 32, 2, 2),
                                        (64, 32, 1, 3)])
def test_tconv_patch_kernel(H, Cin, G, Bg):
    """The k4 s2 p1 transposed convolutions with 32 output channels (64 -> 32 on 16x16 inputs; the 32 -> 32 stages of the
    128 / 256 pixel stacks on 32x32 / 64x64 inputs) run the patch-resident kernel (tconv_patch.hip: a tile of input rows staged
    in LDS once for all parity classes and taps) in every fp32 launch form: BatchNorm partial sums (one tile per row tile of an
    image), bias + Swish second output, and the input-gradient form with the BatchNorm+Swish backward epilogue."""
    Bt, Ho = G * Bg, 2 * H
    x = rnd(Bt * H * H, Cin, seed=130)
    Bp = rnd(16, 32, Cin, seed=131, scale=0.1)
    rows = Bt * Ho * Ho
    T = HIP.igemm_stat_tiles(TCONV_S2P1, G, Bg, H, H, Cin, Ho, Ho, 32)
    assert T == Bg * {16: 1, 32: 2, 64: 8}[H]
    post = lambda i, t: t.sum(1) if t.dim() == 4 else t
    both("igemm_nt", [x, Bp, None, torch.zeros(rows, 32), None, torch.zeros(G, T, 2, 32), None, TCONV_S2P1, G, Bg, H, H, Cin, Ho, Ho,
                      32, 32, 1, 0, 0, 1], [3, 5], post)
    both("igemm_nt", [x, Bp, rnd(32, seed=132), torch.zeros(rows, 32), torch.zeros(rows, 32), None, None, TCONV_S2P1, G, Bg, H, H,
                      Cin, Ho, Ho, 32, 32, 1, 0, 1, 1], [3, 4])
    y = rnd(rows, 32, seed=133) * 1.5 + 0.2
    mean, rstd = rnd(G, 32, seed=134) * 0.3, rnd(G, 32, seed=135).abs() + 0.5
    gamma, beta = rnd(32, seed=136) + 1.2, rnd(32, seed=137)
    both("igemm_nt_dgrad_bn", [x, Bp, torch.zeros(rows, 32), torch.zeros(G, T, 2, 32), y, mean, rstd, gamma, beta, TCONV_S2P1, G, Bg,
                               H, H, Cin, Ho, Ho, 32, 1, 0], [2, 3], post, tol=5e-5)


def test_bce_logits_groups_equals_per_pass_launches():
    """One launch for all decoder passes of a modality (shared target, one loss slot per pass, a discarded pass marked by
    slot -1) against one mmdyn_bce_logits per pass."""
    B, G = 5, 4
    n = B * 3 * 64 * 64
    lg, tg = rnd(G * n, seed=400) * 3, torch.rand(n, generator=torch.Generator().manual_seed(401))
    slots = [3, -1, 0, 5]
    ref_loss, ref_d = torch.zeros(8, dtype=torch.float64, device=DEV), torch.zeros(G * n, device=DEV)
    lgd, tgd = lg.to(DEV), tg.to(DEV)
    for g, s in enumerate(slots):
        if s >= 0:
            HIP.bce_logits(lgd[g * n:(g + 1) * n], tgd, None, ref_d[g * n:(g + 1) * n], ref_loss[s:s + 1], n, 3 * 4096, 4096, 0.2)
    loss, d = torch.zeros(8, dtype=torch.float64, device=DEV), torch.full((G * n,), 7.0, device=DEV)
    HIP.bce_logits_groups(lgd, tgd, d, loss, slots, n, 0.2)
    torch.cuda.synchronize()
    assert torch.equal(d.cpu(), ref_d.cpu())
    assert torch.allclose(loss.cpu(), ref_loss.cpu(), rtol=1e-12)
    cpu_loss, cpu_d = torch.zeros(8, dtype=torch.float64), torch.zeros(G * n)
    EMU.bce_logits_groups(lg.clone(), tg.clone(), cpu_d, cpu_loss, slots, n, 0.2)
    assert rel(d.cpu(), cpu_d) <= 2e-5 and torch.allclose(loss.cpu(), cpu_loss, rtol=1e-5)


@pytest.mark.parametrize("mask_c", [1, 3])
def test_bce_logits_groups_masked(mask_c):
    """The --mask-loss form of the grouped launch: masked sums + gradients equal one masked mmdyn_bce_logits per pass, the
    unmasked slots equal the unmasked launch (problems.py:445-447, 495-505)."""
    B, G = 5, 3
    n = B * 3 * 64 * 64
    lg, tg = rnd(G * n, seed=410) * 3, torch.rand(n, generator=torch.Generator().manual_seed(411))
    mk = (torch.rand(B * mask_c * 4096, generator=torch.Generator().manual_seed(412)) > 0.35).float()
    slots = [2, -1, 6]
    lgd, tgd, mkd = lg.to(DEV), tg.to(DEV), mk.to(DEV)
    ref_loss, ref_un = torch.zeros(8, dtype=torch.float64, device=DEV), torch.zeros(8, dtype=torch.float64, device=DEV)
    ref_d = torch.zeros(G * n, device=DEV)
    for g, s in enumerate(slots):
        if s >= 0:
            HIP.bce_logits(lgd[g * n:(g + 1) * n], tgd, mkd, ref_d[g * n:(g + 1) * n], ref_loss[s:s + 1], n, 3 * 4096, 4096, 0.2,
                           mask_channels=mask_c)
            HIP.bce_logits(lgd[g * n:(g + 1) * n], tgd, None, None, ref_un[s:s + 1], n, 3 * 4096, 4096, 0.2)
    loss, un = torch.zeros(8, dtype=torch.float64, device=DEV), torch.zeros(8, dtype=torch.float64, device=DEV)
    d = torch.full((G * n,), 7.0, device=DEV)
    HIP.bce_logits_groups(lgd, tgd, d, loss, slots, n, 0.2, mask=mkd, chw=3 * 4096, hw=4096, mask_channels=mask_c,
                          unmasked_slots=un)
    torch.cuda.synchronize()
    assert torch.equal(d.cpu(), ref_d.cpu())
    assert torch.allclose(loss.cpu(), ref_loss.cpu(), rtol=1e-12) and torch.allclose(un.cpu(), ref_un.cpu(), rtol=1e-12)
    cpu_loss, cpu_un, cpu_d = torch.zeros(8, dtype=torch.float64), torch.zeros(8, dtype=torch.float64), torch.zeros(G * n)
    EMU.bce_logits_groups(lg.clone(), tg.clone(), cpu_d, cpu_loss, slots, n, 0.2, mask=mk, chw=3 * 4096, hw=4096,
                          mask_channels=mask_c, unmasked_slots=cpu_un)
    assert rel(d.cpu(), cpu_d) <= 2e-5 and torch.allclose(loss.cpu(), cpu_loss, rtol=1e-5) and torch.allclose(un.cpu(), cpu_un, rtol=1e-5)
    with pytest.raises(ValueError):
        HIP.bce_logits_groups(lgd, tgd, d, loss, slots, n, 0.2, mask=mkd[:-4], chw=3 * 4096, hw=4096, mask_channels=mask_c)


@pytest.mark.parametrize("H", [32, 128, 256])
def test_im2col_other_image_sizes(H):
    """The 3-channel layers at other image sizes.  128 / 256 (the extended stacks of BASELINE configs[3] / configs[4]) run
    the specialised kernels of conv3.hip with 2 / 4 segments per output row (segment borders need the halo columns, image
    borders zeros); anything else (32 here) takes the tiled implicit-GEMM / weight-gradient kernels with the on-the-fly
    im2col gather.  Forward with statistics, input gradient with the BatchNorm epilogue, weight gradient."""
    G, Bg = 2, 3
    Bt, Ho = G * Bg, H // 2
    x = rnd(Bt, 3, H, H, seed=110)
    Bp = rnd(1, 32, 64, seed=111, scale=0.2)
    Bp[:, :, 48:] = 0
    rows = Bt * Ho * Ho
    T = HIP.igemm_stat_tiles(IM2COL3, G, Bg, H, H, 64, Ho, Ho, 32)
    post = lambda i, t: t.sum(1) if t.dim() == 4 else t
    C, Ca, stats = torch.zeros(rows, 32), torch.zeros(rows, 32), torch.zeros(G, T, 2, 32)
    both("igemm_nt", [x, Bp, None, C, None, stats, None, IM2COL3, G, Bg, H, H, 64, Ho, Ho, 32, 32, 1, 0, 0, 1], [3, 5], post)
    both("igemm_nt", [x, Bp, None, C, Ca, None, None, IM2COL3, G, Bg, H, H, 64, Ho, Ho, 32, 32, 1, 0, 1, 1], [3, 4], post)
    y = rnd(rows, 32, seed=112) * 1.5 + 0.2
    mean, rstd = rnd(G, 32, seed=113) * 0.3, rnd(G, 32, seed=114).abs() + 0.5
    gamma, beta = rnd(32, seed=115) + 1.2, rnd(32, seed=116)
    both("igemm_nt_dgrad_bn", [x, Bp, C, stats, y, mean, rstd, gamma, beta, IM2COL3, G, Bg, H, H, 64, Ho, Ho, 32, 1, 0],
         [2, 3], post, tol=5e-5)
    D = rnd(rows, 32, seed=117)
    chunks = HIP.wgrad_chunks(IM2COL3, rows, 32, 64)
    both("wgrad_tn", [D, x, torch.zeros(chunks, 1, 32, 64), IM2COL3, Bt, Ho, Ho, 32, H, H, 64, 1, 0, chunks], [2],
         lambda i, t: t.sum(0), tol=5e-5)


def test_sgd_matches_torch():
    n = 50001
    p0, g = rnd(n, seed=80), rnd(n, seed=81) * 0.1
    p, buf = p0.clone().to(DEV), torch.zeros(n, device=DEV)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.SGD([ref], lr=1e-2, momentum=0.9, weight_decay=5e-4)
    for step in range(3):
        HIP.sgd_step(p, (g * (step + 1)).to(DEV), buf, 1e-2, 0.9, 5e-4, 1.0, step == 0)
        ref.grad = g * (step + 1)
        opt.step()
    torch.testing.assert_close(p.cpu(), ref.detach(), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("size,cond,w_dtype", [(64, 0, torch.float32), (128, 0, torch.float32), (64, 3, torch.float32),
                                               (64, 0, torch.bfloat16)])
def test_pack_plan_equals_per_entry_packs(size, cond, w_dtype):
    """mmdyn_pack_plan (one launch, LDS-tiled transposes / permutations) against the element-wise pack kernels entry by
    entry: every packed operand of an encoder, its heads and a decoder, bit for bit (a pack is a permutation + padding)."""
    from mmdyn_hip import layers
    from mmdyn_hip.models.shapes import image_encoder_shapes, image_decoder_shapes
    P = {}
    for k, shp in list(image_encoder_shapes("e", 256, cond, size).items()) + list(image_decoder_shapes("d", 256, cond, size).items()):
        if "running" in k or "num_batches" in k:
            continue
        P[k] = rnd(*shp, seed=len(P) + 7).to(DEV)
    enc = {k[2:]: v for k, v in P.items() if k.startswith("e.")}
    dec = {k[2:]: v for k, v in P.items() if k.startswith("d.")}
    specs = {"e": layers.encoder_pack_specs(enc), "h": layers.heads_pack_specs(enc), "d": layers.decoder_pack_specs(dec)}
    prev = layers.W_DTYPE
    layers.W_DTYPE = w_dtype
    try:
        plan = layers.PackPlan(specs, early=("W1p", "W2k", "Wf"), w_dtype=w_dtype)
        plan.run_early()
        plan.run_late()
        ref = {k: layers.pack_now(v) for k, v in specs.items()}
    finally:
        layers.W_DTYPE = prev
    torch.cuda.synchronize()
    for grp in specs:
        for name, t in ref[grp].items():
            assert torch.equal(plan.packed[grp][name].float().cpu(), t.float().cpu()), (grp, name)
