#!/usr/bin/env python3
"""Diagnostic: time the MFMA GEMM launches of one bs=256 train step in isolation (per shape TFLOP/s)."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
from mmdyn_hip import ops  # noqa: E402

IGEMM = [  # mode,G,Bg,Hi,Wi,Cin,Ho,Wo,N,ldc,stride,offset,act,splitk
    (1, 1, 1024, 8, 8, 128, 5, 5, 256, 256, 1, 0, 0, 1),
    (0, 1, 25600, 1, 1, 256, 1, 1, 2048, 2048, 1, 0, 0, 1),
    (2, 4, 256, 16, 16, 64, 32, 32, 32, 32, 1, 0, 0, 1),
    (2, 4, 256, 8, 8, 128, 16, 16, 64, 64, 1, 0, 0, 1),
    (1, 1, 1024, 32, 32, 32, 16, 16, 64, 64, 2, -1, 0, 1),
    (1, 1, 1024, 16, 16, 64, 8, 8, 128, 128, 2, -1, 0, 1),
    (1, 1, 256, 8, 8, 128, 5, 5, 256, 256, 1, 0, 0, 1),
    (0, 1, 1048576, 1, 1, 32, 1, 1, 64, 64, 1, 0, 0, 1),
    (0, 1, 6400, 1, 1, 256, 1, 1, 2048, 2048, 1, 0, 0, 1),
    (0, 1, 1048576, 1, 1, 64, 1, 1, 32, 32, 1, 0, 0, 1),
    (0, 1, 1024, 1, 1, 256, 1, 1, 6400, 6400, 1, 0, 1, 1),
    (2, 1, 256, 16, 16, 64, 32, 32, 32, 32, 1, 0, 0, 1),
    (1, 1, 256, 16, 16, 64, 8, 8, 128, 128, 2, -1, 0, 1),
    (1, 1, 256, 32, 32, 32, 16, 16, 64, 64, 2, -1, 0, 1),
    (2, 1, 256, 8, 8, 128, 16, 16, 64, 64, 1, 0, 0, 1),
]
WGRAD = [  # mode,Bt,Hr,Wr,Cd,Hi,Wi,Cg,stride,offset,chunks
    (1, 1024, 5, 5, 256, 8, 8, 128, 1, 0, 32),
    (1, 1024, 16, 16, 64, 32, 32, 32, 2, -1, 128),
    (1, 1024, 8, 8, 128, 16, 16, 64, 2, -1, 32),
    (0, 1048576, 1, 1, 32, 1, 1, 64, 1, 0, 2048),
    (1, 256, 5, 5, 256, 8, 8, 128, 1, 0, 32),
    (0, 1024, 1, 1, 6400, 1, 1, 256, 1, 0, 8),
    (1, 256, 16, 16, 64, 32, 32, 32, 2, -1, 128),
]


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
    return s.elapsed_time(e) / reps


def main():
    dev = "cuda"
    tot_ms, tot_fl = 0.0, 0.0
    for sh in IGEMM:
        mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, ldc, stride, offset, act, splitk = sh
        Bt = G * Bg
        taps = 1 if mode == 0 else 16
        A = torch.randn(Bt * Hi * Wi, Cin, device=dev)
        Bp = torch.randn(taps, N, Cin, device=dev) * 0.1
        C = torch.empty(Bt * Ho * Wo, N, device=dev)
        try:
            ms = timeit(lambda: ops.B.igemm_nt(A, Bp, None, C, None, None, None, *sh))
        except Exception as e:
            print(f"igemm {str(sh):62s} skipped ({e})")
            continue
        fl = 2.0 * Bt * Ho * Wo * N * Cin * (1 if mode == 0 else (16 if mode == 1 else 4))
        tot_ms += ms
        tot_fl += fl
        print(f"igemm {str(sh):62s} {ms * 1e3:8.1f} us {fl / ms / 1e9:6.1f} TF/s")
    print(f"igemm total {tot_ms:.3f} ms  {tot_fl / tot_ms / 1e9:.1f} TF/s")
    tot_ms, tot_fl = 0.0, 0.0
    for sh in WGRAD:
        mode, Bt, Hr, Wr, Cd, Hi, Wi, Cg, stride, offset, chunks = sh
        taps = 16 if mode == 1 else 1
        D = torch.randn(Bt * Hr * Wr, Cd, device=dev)
        Gt = torch.randn(Bt * Hi * Wi, Cg, device=dev)
        part = torch.empty(chunks, taps, Cd, Cg, device=dev)
        ms = timeit(lambda: ops.B.wgrad_tn(D, Gt, part, *sh))
        fl = 2.0 * Bt * Hr * Wr * Cd * Cg * taps
        tot_ms += ms
        tot_fl += fl
        print(f"wgrad {str(sh):62s} {ms * 1e3:8.1f} us {fl / ms / 1e9:6.1f} TF/s")
    print(f"wgrad total {tot_ms:.3f} ms  {tot_fl / tot_ms / 1e9:.1f} TF/s")


if __name__ == "__main__":
    main()
