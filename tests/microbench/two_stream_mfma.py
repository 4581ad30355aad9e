#!/usr/bin/env python3
"""Diagnostic: sustained fp32 MFMA rate of the implicit-GEMM kernel with one stream vs two concurrent streams
(the situation of the two-lane step), on a few of the step's larger shapes."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
from mmdyn_hip import ops  # noqa: E402

SHAPES = [
    (1, 1, 1024, 8, 8, 128, 5, 5, 256, 256, 1, 0, 0, 1),
    (1, 4, 256, 16, 16, 64, 8, 8, 128, 128, 2, -1, 0, 1),
    (2, 4, 256, 8, 8, 128, 16, 16, 64, 64, 1, 0, 0, 1),
    (0, 1, 6400, 1, 1, 256, 1, 1, 2048, 2048, 1, 0, 0, 1),
]


def flops(sh):
    mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N = sh[:9]
    taps = {0: 1, 1: 16, 2: 4}[mode]
    return 2.0 * G * Bg * Ho * Wo * N * taps * Cin


def main():
    dev = "cuda"
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for sh in SHAPES:
        mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N = sh[:9]
        Bt = G * Bg
        taps = 1 if mode == 0 else 16
        bufs = []
        for _ in range(2):
            A = torch.randn(Bt * Hi * Wi * Cin, device=dev)
            Bp = torch.randn(taps, N, Cin, device=dev) * 0.1
            C = torch.empty(Bt * Ho * Wo, N, device=dev)
            bufs.append((A, Bp, C))
        reps = 40

        def run(streams):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for st in streams:
                st.wait_event(a)
            for _ in range(reps):
                for k, st in enumerate(streams):
                    with torch.cuda.stream(st):
                        A, Bp, C = bufs[k]
                        ops.B.igemm_nt(A, Bp, None, C, None, None, None, *sh)
            for st in streams:
                torch.cuda.current_stream().wait_stream(st)
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) * 1e-3

        run([s1])
        t1 = run([s1])
        t2 = run([s1, s2])
        print(f"{sh[:9]}: one stream {flops(sh) * reps / t1 / 1e12:6.1f} TF/s, two streams {2 * flops(sh) * reps / t2 / 1e12:6.1f} TF/s aggregate")


if __name__ == "__main__":
    main()
