#!/usr/bin/env python3
"""Diagnostic: the N % 128 == 0 implicit-GEMM launches of one bs=256 train step in the fp32x3 arithmetic, each alone on the chip:
(x3) the persistent kernel that splits its fp32 fragments in the MFMA waves' registers (igemm_wsp_kernel<X3>) against (p3) the
plane-ring kernel on operands that ARRIVE split (igemm_wsp3_kernel); the split launch itself (mmdyn_split_planes: 10 bytes per
element) is timed beside them.  Interleaved rounds in one process, product library."""
import os
import statistics
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
from mmdyn_hip import ops  # noqa: E402

HIP = ops.HipBackend()
HIP.fp32_split = True
# mode,G,Bg,Hi,Cin,Ho,N,stride,offset, kind
SHAPES = [
    (4, 4, 256, 5, 256, 8, 128, 1, 0, "stats"),         # decoder layer 1: the k4 s1 p0 transposed convolution
    (1, 1, 1024, 8, 128, 5, 256, 1, 0, "actbwd"),       # decoder layer-1 input gradient
    (1, 4, 256, 16, 64, 8, 128, 2, -1, "bnbwd"),        # decoder layer-2 input gradient
    (1, 4, 256, 16, 64, 8, 128, 2, -1, "plain"),
    (4, 4, 128, 5, 256, 8, 128, 1, 0, "stats"),         # the same at the bs 128 share
    (1, 1, 512, 8, 128, 5, 256, 1, 0, "actbwd"),
    (1, 4, 128, 16, 64, 8, 128, 2, -1, "bnbwd"),
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
    tot = {"x3": 0.0, "p3": 0.0}
    for sh in SHAPES:
        mode, G, Bg, Hi, Cin, Ho, N, stride, offset, kind = sh
        if not HIP.igemm_planes_served(mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N):
            print(sh, "not served")
            continue
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
        T = HIP.igemm_stat_tiles(mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N) if kind in ("stats", "bnbwd") else 0
        st = torch.empty(G, T, 2, N, device=dev) if T else None
        ws = HIP._slabs(C, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N)

        def launch(a, b):
            if kind == "bnbwd":
                return lambda: HIP.igemm_nt_dgrad_bn(a, b, C, st, y, mean, rstd, gamma, beta, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, stride, offset)
            if kind == "actbwd":
                return lambda: HIP.igemm_nt_dgrad_act(a, b, C, y, 1, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, stride, offset)
            return lambda: HIP.igemm_nt(a, b, None, C, None, st, ws, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, N, stride, offset, 0, 1)
        fns = {"x3": launch(A, Bp), "p3": launch(Ap, Bq)}
        res, times = {}, {"x3": [], "p3": []}
        for rnd in range(5):
            for k in ("x3", "p3"):
                if rnd == 0:
                    for _ in range(3):
                        fns[k]()
                    torch.cuda.synchronize()
                    res[k] = C.clone()
                times[k].append(event_ms(fns[k], 10))
        t_split = event_ms(lambda: HIP.split_planes(A, Ap), 10)
        fl = 2.0 * rows * N * Cin * ({0: 1, 1: 16, 2: 4}[mode]) if mode != 4 else 2.0 * Bt * Hi * Hi * N * 16 * Cin
        m0, m1 = statistics.median(times["x3"]), statistics.median(times["p3"])
        tot["x3"] += m0
        tot["p3"] += m1
        same = torch.equal(res["x3"], res["p3"])
        print(f"{str(sh):52s} x3 {m0 * 1e3:7.1f} us {fl / m0 / 1e9:6.1f} TF/s | p3 {m1 * 1e3:7.1f} us {fl / m1 / 1e9:6.1f} TF/s "
              f"| x{m0 / m1:5.2f}  bit-identical {same} | split of A {t_split * 1e3:6.1f} us ({A.numel() * 10 / t_split / 1e9:5.2f} TB/s)", flush=True)
    print(f"sum x3 {tot['x3']:.3f} ms, p3 {tot['p3']:.3f} ms")


if __name__ == "__main__":
    main()
