#!/usr/bin/env python3
"""Diagnostic: implicit-GEMM launches of one bs=256 train step in the fp32x3 arithmetic, each alone on the chip:
(x3) what the product dispatch runs when the operands are fp32 -- the persistent kernel that splits its fragments in the MFMA
waves' registers, or the register-staged split kernel -- against (p3) the plane-ring kernel on operands that ARRIVE split
(igemm_wsp3_kernel), in the configurations named on the command line.  LAB build of the library (forces tiles / thresholds).
usage: ab_p3.py [patch] [cfg ...]     cfg = "BM,BN,S" (forced through MMDYN_P3_TILE) or "rule" (the library's own pick; default);
"patch": the 32-channel up-sampling launches instead (tconv_patch_kernel on fp32 operands against its plane form)"""
import os
import statistics
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
from mmdyn_hip import ops, _lib  # noqa: E402

HIP = ops.HipBackend(lib_path=_lib.LAB_LIB_PATH)
HIP.fp32_split = True
# mode,G,Bg,Hi,Cin,Ho,N,stride,offset, kind
SHAPES = [
    (4, 4, 256, 5, 256, 8, 128, 1, 0, "stats"),         # decoder layer 1: the k4 s1 p0 transposed convolution
    (1, 1, 1024, 8, 128, 5, 256, 1, 0, "actbwd"),       # decoder layer-1 input gradient
    (1, 4, 256, 16, 64, 8, 128, 2, -1, "bnbwd"),        # decoder layer-2 input gradient
    (1, 4, 256, 32, 32, 16, 64, 2, -1, "bnbwd"),        # decoder layer-3 input gradient (N = 64)
    (2, 4, 256, 8, 128, 16, 64, 1, 0, "stats"),         # decoder layer 2 forward (N = 64)
    (1, 1, 256, 8, 128, 5, 256, 1, 0, "stats"),         # encoder conv4
    (1, 1, 256, 16, 64, 8, 128, 2, -1, "stats"),        # encoder conv3
    (1, 1, 256, 32, 32, 16, 64, 2, -1, "stats"),        # encoder conv2 (N = 64)
    (2, 1, 256, 8, 128, 16, 64, 1, 0, "bnbwd"),         # encoder conv3 input gradient (N = 64)
    (4, 1, 256, 5, 256, 8, 128, 1, 0, "plain"),         # encoder conv4 input gradient (k4 s1 p0, one group)
]


PATCH_SHAPES = [
    (2, 4, 256, 16, 64, 32, 32, 1, 0, "stats"),         # decoder layer 3 forward (64 -> 32 channels)
    (2, 1, 256, 16, 64, 32, 32, 1, 0, "actbwd"),        # encoder conv2 input gradient
    (2, 3, 128, 32, 32, 64, 32, 1, 0, "stats"),         # 128-pixel stacks: 32 -> 32 channels on 32x32 inputs
    (2, 3, 128, 32, 32, 64, 32, 1, 0, "bnbwd"),
    (2, 1, 64, 64, 32, 128, 32, 1, 0, "stats"),         # 256-pixel stacks: 32 -> 32 channels on 64x64 inputs
]


def event_ms(fn, reps):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    dev = "cuda"
    args = list(sys.argv[1:])
    shapes = SHAPES
    if args and args[0] == "patch":
        shapes, args = PATCH_SHAPES, args[1:]
    cfgs = args or ["rule"]
    os.environ["MMDYN_P3_MIN_UNITS"] = "1"
    os.environ["MMDYN_P3_N64"] = "1"
    tot = {}
    for sh in shapes:
        mode, G, Bg, Hi, Cin, Ho, N, stride, offset, kind = sh
        Bt = G * Bg
        A = torch.randn(Bt * Hi * Hi, Cin, device=dev)
        Bp = torch.randn(16, N, Cin, device=dev) * 0.1
        Ap, Bq = ops.Planes(A.shape[0], Cin, dev), ops.Planes(16 * N, Cin, dev)
        HIP.split_planes(A, Ap)
        HIP.split_planes(Bp.view(-1, Cin), Bq)
        rows = Bt * Ho * Ho
        C = torch.empty(rows, N, device=dev)
        y = torch.randn(rows, N, device=dev)
        mean, rstd = torch.randn(G, N, device=dev), torch.rand(G, N, device=dev) + 0.5
        gamma, beta = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)

        def launch(a, b, planes):
            T = HIP.igemm_stat_tiles(mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, planes=planes) if kind in ("stats", "bnbwd") else 0
            st = torch.empty(G, T, 2, N, device=dev) if T else None
            ws = HIP._slabs(C, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, planes=planes)
            if kind == "bnbwd":
                return lambda: HIP.igemm_nt_dgrad_bn(a, b, C, st, y, mean, rstd, gamma, beta, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, stride, offset)
            if kind == "actbwd":
                return lambda: HIP.igemm_nt_dgrad_act(a, b, C, y, 1, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, stride, offset)
            return lambda: HIP.igemm_nt(a, b, None, C, None, st, ws, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, N, stride, offset, 0, 1)
        variants = ["x3"] + cfgs
        res, times = {}, {v: [] for v in variants}
        for rnd in range(4):
            for v in variants:
                if v == "x3":
                    fn = launch(A, Bp, False)
                else:
                    if v == "rule":
                        os.environ.pop("MMDYN_P3_TILE", None)
                    else:
                        os.environ["MMDYN_P3_TILE"] = v
                    if not HIP.igemm_planes_served(mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N) or (v != "rule" and N % int(v.split(",")[1])):
                        times[v].append(float("nan"))
                        continue
                    fn = launch(Ap, Bq, True)
                if rnd == 0:
                    for _ in range(3):
                        fn()
                    torch.cuda.synchronize()
                    res[v] = C.clone()
                times[v].append(event_ms(fn, 10))
        fl = 2.0 * rows * N * Cin * ({0: 1, 1: 16, 2: 4}[mode]) if mode != 4 else 2.0 * Bt * Hi * Hi * N * 16 * Cin
        line = f"{str(sh):50s}"
        for v in variants:
            m = statistics.median(times[v])
            tot[v] = tot.get(v, 0.0) + (m if m == m else statistics.median(times["x3"]))
            err = float((res[v] - res["x3"]).norm() / (res["x3"].norm() + 1e-30)) if v in res else float("nan")
            line += f" | {v:9s} {m * 1e3:7.1f} us {fl / m / 1e9:6.1f} TF/s" + (f" d{err:.0e}" if v != "x3" else "")
        print(line, flush=True)
    print("sums (ms; a configuration that does not serve a shape is booked at the x3 time):", {k: round(v, 3) for k, v in tot.items()})


if __name__ == "__main__":
    main()
