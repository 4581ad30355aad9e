#!/usr/bin/env python3
"""Diagnostic: launch a few igemm_nt shapes a few times each (for rocprofv3 --pmc per-dispatch counters)."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
from mmdyn_hip import ops  # noqa: E402

SHAPES = [
    (1, 1, 1024, 8, 8, 128, 5, 5, 256, 256, 1, 0, 0, 1),
    (1, 1, 1024, 16, 16, 64, 8, 8, 128, 128, 2, -1, 0, 1),
    (2, 4, 256, 8, 8, 128, 16, 16, 64, 64, 1, 0, 0, 1),
    (0, 1, 6400, 1, 1, 256, 1, 1, 2048, 2048, 1, 0, 0, 1),
]


def main():
    dev = "cuda"
    for sh in SHAPES:
        mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, ldc, stride, offset, act, splitk = sh
        Bt = G * Bg
        taps = 1 if mode in (0, 3) else 16
        A = torch.randn(Bt * Hi * Wi * Cin, device=dev)
        Bp = torch.randn(taps, N, Cin, device=dev) * 0.1
        C = torch.empty(Bt * Ho * Wo, N, device=dev)
        for _ in range(4):
            ops.B.igemm_nt(A, Bp, None, C, None, None, None, *sh)
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
