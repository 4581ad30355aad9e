#!/usr/bin/env python3
"""Diagnostic: the k4 s1 p0 transposed convolution (decoder layer 1) in the bf16-storage mode, alone on the chip."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
from mmdyn_hip import ops  # noqa: E402
from merged_launch_estimate import timeit  # noqa: E402


def main():
    ops.B.precision = "bf16s"
    for G, Bg in ((4, 256), (1, 256), (4, 128), (1, 128), (8, 256)):
        Bt = G * Bg
        A = torch.randn(Bt * 25, 256, device="cuda").to(torch.bfloat16)
        Bp = (torch.randn(16, 128, 256, device="cuda") * 0.1).to(torch.bfloat16)
        C = torch.empty(Bt * 64, 128, device="cuda", dtype=torch.bfloat16)
        sh = (4, G, Bg, 5, 5, 256, 8, 8, 128, 128, 1, 0, 0, 1)
        ms = timeit(lambda: ops.B.igemm_nt(A, Bp, None, C, None, None, None, *sh))
        T = ops.B.igemm_stat_tiles(4, G, Bg, 5, 5, 256, 8, 8, 128)
        stats = torch.empty(G, T, 2, 128, device="cuda")
        ms2 = timeit(lambda: ops.B.igemm_nt(A, Bp, None, C, None, stats, None, *sh))
        fl = 2.0 * Bt * 25 * 128 * 16 * 256
        print(f"bf16s s1p0 G={G} Bg={Bg}: {ms * 1e3:7.1f} us {fl / ms / 1e9:6.1f} TF/s; with BatchNorm partial sums (T={T}) {ms2 * 1e3:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
