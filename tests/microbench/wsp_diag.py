#!/usr/bin/env python3
"""Diagnostic (LAB build): where a K-step of the persistent ring kernel goes -- cycle stamps inside igemm_wsp_kernel
(MMDYN_WSP_DIAG=1) on the large launches of the step, each alone on the chip.
usage: wsp_diag.py [tile]"""
import ctypes
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
from mmdyn_hip import ops, _lib  # noqa: E402

HIP = ops.HipBackend(lib_path=_lib.LAB_LIB_PATH)
SHAPES = [(1, 1, 1024, 8, 128, 5, 256, 1, 0, "actbwd"), (1, 4, 256, 32, 32, 16, 64, 2, -1, "bnbwd"),
          (2, 4, 256, 8, 128, 16, 64, 1, 0, "stats"), (1, 4, 256, 16, 64, 8, 128, 2, -1, "bnbwd"),
          (1, 4, 256, 16, 64, 8, 128, 2, -1, "plain")]


def main():
    dev = "cuda"
    if len(sys.argv) > 1:
        os.environ["MMDYN_WSP_TILE"] = sys.argv[1]
    lib = ctypes.CDLL(_lib.LAB_LIB_PATH)
    out = (ctypes.c_ulonglong * 8)()
    for sh in SHAPES:
        mode, G, Bg, Hi, Cin, Ho, N, stride, offset, kind = sh
        Bt = G * Bg
        A = torch.randn(Bt * Hi * Hi, Cin, device=dev)
        Bp = torch.randn(16, N, Cin, device=dev) * 0.1
        rows = Bt * Ho * Ho
        C = torch.empty(rows, N, device=dev)
        y = torch.randn(rows, N, device=dev)
        mean, rstd = torch.randn(G, N, device=dev), torch.rand(G, N, device=dev) + 0.5
        gamma, beta = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
        T = HIP.igemm_stat_tiles(mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N) if kind in ("stats", "bnbwd") else 0
        st = torch.empty(G, T, 2, N, device=dev) if T else None
        if kind == "bnbwd":
            fn = lambda: HIP.igemm_nt_dgrad_bn(A, Bp, C, st, y, mean, rstd, gamma, beta, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, stride, offset)
        elif kind == "actbwd":
            fn = lambda: HIP.igemm_nt_dgrad_act(A, Bp, C, y, 1, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, stride, offset)
        else:
            fn = lambda: HIP.igemm_nt(A, Bp, None, C, None, st, None, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, N, stride, offset, 0, 1)
        os.environ["MMDYN_WSP_DIAG"] = "0"
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        os.environ["MMDYN_WSP_DIAG"] = "1"
        lib.mmdyn_lab_wsp_diag(out)
        reps = 10
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) / reps * 1e3
        lib.mmdyn_lab_wsp_diag(out)
        d = [float(x) for x in out]
        fl = 2.0 * rows * N * Cin * (16 if mode == 1 else 4)
        print(f"{str(sh):50s} {us:7.1f} us ({fl / us / 1e6:5.1f} TF/s with stamps) | MFMA wave: barrier {100 * d[0] / max(d[3], 1):4.1f} %, "
              f"K loops {100 * d[1] / max(d[3], 1):4.1f} %, epilogue {100 * d[2] / max(d[3], 1):4.1f} % of {d[3] / reps / 256:9.0f} cycles per block"
              f" | loader: vmcnt {100 * d[4] / max(d[7], 1):4.1f} %, barrier {100 * d[5] / max(d[7], 1):4.1f} %, issue {100 * d[6] / max(d[7], 1):4.1f} %",
              flush=True)


if __name__ == "__main__":
    main()
