#!/usr/bin/env python3
"""Diagnostic: every igemm_nt shape of the bs=256 step under each forced tile (MMDYN_IGEMM_TILE)."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
from mmdyn_hip import ops  # noqa: E402

IGEMM = [  # mode,G,Bg,Hi,Wi,Cin,Ho,Wo,N,ldc,stride,offset,act,splitk  (count per step)
    ((1, 1, 1024, 8, 8, 128, 5, 5, 256, 256, 1, 0, 0, 1), 2),
    ((4, 4, 256, 5, 5, 256, 8, 8, 128, 128, 1, 0, 0, 1), 2),
    ((2, 4, 256, 16, 16, 64, 32, 32, 32, 32, 1, 0, 0, 1), 2),
    ((2, 4, 256, 8, 8, 128, 16, 16, 64, 64, 1, 0, 0, 1), 2),
    ((1, 1, 1024, 32, 32, 32, 16, 16, 64, 64, 2, -1, 0, 1), 2),
    ((1, 1, 1024, 16, 16, 64, 8, 8, 128, 128, 2, -1, 0, 1), 2),
    ((3, 1, 1024, 64, 64, 64, 32, 32, 32, 32, 1, 0, 0, 1), 2),
    ((1, 1, 256, 8, 8, 128, 5, 5, 256, 256, 1, 0, 0, 1), 2),
    ((0, 1, 6400, 1, 1, 256, 1, 1, 2048, 2048, 1, 0, 0, 1), 2),
    ((2, 1, 256, 16, 16, 64, 32, 32, 32, 32, 1, 0, 0, 1), 2),
    ((0, 1, 1024, 1, 1, 256, 1, 1, 6400, 6400, 1, 0, 1, 1), 2),
    ((3, 1, 256, 64, 64, 64, 32, 32, 32, 32, 1, 0, 1, 1), 2),
    ((1, 1, 256, 16, 16, 64, 8, 8, 128, 128, 2, -1, 0, 1), 2),
    ((1, 1, 256, 32, 32, 32, 16, 16, 64, 64, 2, -1, 0, 1), 2),
    ((2, 1, 256, 8, 8, 128, 16, 16, 64, 64, 1, 0, 0, 1), 2),
]
TILES = ["128,128", "128,64", "64,128", "64,64", "256,32", "128,32"]


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def main():
    dev = "cuda"
    for sh, cnt in IGEMM:
        mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, ldc, stride, offset, act, splitk = sh
        Bt = G * Bg
        taps = 1 if mode in (0, 3) else 16
        A = torch.randn(Bt * 3 * Hi * Wi if mode == 3 else Bt * Hi * Wi * Cin, device=dev)
        Bp = torch.randn(taps, N, Cin, device=dev) * 0.1
        C = torch.empty(Bt * Ho * Wo, N, device=dev)
        Ca = torch.empty(Bt * Ho * Wo, N, device=dev) if act else None
        bias = torch.zeros(N, device=dev) if act else None
        macs = {0: Cin, 1: 16 * Cin, 2: 4 * Cin, 3: 48, 4: Cin * 25 / 4}[mode]
        fl = 2.0 * Bt * Ho * Wo * N * macs
        os.environ.pop("MMDYN_IGEMM_TILE", None)
        base = timeit(lambda: ops.B.igemm_nt(A, Bp, bias, C, Ca, None, None, *sh))
        line = f"{str(sh):58s} x{cnt} default {base:7.1f}us {fl / base / 1e6:6.1f}TF |"
        for tile in TILES:
            bn = int(tile.split(",")[1])
            if N % bn or (bn == 32 and N % 64 == 0):
                continue
            os.environ["MMDYN_IGEMM_TILE"] = tile
            try:
                us = timeit(lambda: ops.B.igemm_nt(A, Bp, bias, C, Ca, None, None, *sh))
                line += f" {tile}:{us:7.1f}"
            except Exception as e:
                line += f" {tile}: err"
        print(line, flush=True)


if __name__ == "__main__":
    main()
