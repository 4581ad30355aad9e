#!/usr/bin/env python3
"""Diagnostic: the BatchNorm apply passes writing fp32 (plain entry points) against writing planes only, alone on the chip."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
from mmdyn_hip import ops  # noqa: E402

HIP = ops.HipBackend()


def event_ms(fn, reps=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    dev = "cuda"
    for G, rpg, C in ((4, 256 * 1024, 32), (4, 256 * 256, 64), (4, 256 * 64, 128), (1, 256 * 25, 256)):
        rows = G * rpg
        y, da = torch.randn(rows, C, device=dev), torch.randn(rows, C, device=dev)
        mean, rstd = torch.randn(G, C, device=dev), torch.rand(G, C, device=dev) + 0.5
        gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        sums = torch.randn(G, 2, C, device=dev)
        out, pl = torch.empty_like(y), ops.Planes(rows, C, dev)
        n = rows * C
        t0 = event_ms(lambda: HIP.bn_swish_bwd_apply(da, y, mean, rstd, gamma, beta, sums, out, G, rpg, C, True))
        t1 = event_ms(lambda: HIP.bn_swish_bwd_apply(da, y, mean, rstd, gamma, beta, sums, None, G, rpg, C, True, planes=pl))
        t2 = event_ms(lambda: HIP.bn_swish_fwd(y, mean, rstd, gamma, beta, out, G, rpg, C))
        t3 = event_ms(lambda: HIP.bn_swish_fwd(y, mean, rstd, gamma, beta, None, G, rpg, C, planes=pl))
        t4 = event_ms(lambda: HIP.split_planes(y, pl))
        print(f"[{rows} x {C}]  bwd apply fp32 {t0 * 1e3:6.1f} us ({n * 12 / t0 / 1e9:5.2f} TB/s)  planes {t1 * 1e3:6.1f} us ({n * 14 / t1 / 1e9:5.2f} TB/s)"
              f" | fwd fp32 {t2 * 1e3:6.1f} us ({n * 8 / t2 / 1e9:5.2f})  planes {t3 * 1e3:6.1f} us ({n * 10 / t3 / 1e9:5.2f}) | split {t4 * 1e3:6.1f} us ({n * 10 / t4 / 1e9:5.2f})")


if __name__ == "__main__":
    main()
