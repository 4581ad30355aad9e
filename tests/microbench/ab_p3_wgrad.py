#!/usr/bin/env python3
"""Diagnostic: the convolution-level weight-gradient GEMMs of one bs=256 train step in the fp32x3 arithmetic, each alone on the
chip: fp32 operands split inside the kernel (pre 0) against operands that arrive split (D, Gt or both as ops.Planes)."""
import os
import statistics
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
from mmdyn_hip import ops  # noqa: E402

HIP = ops.HipBackend()
HIP.fp32_split = True
# Bt, Hr, Cd, Hi, Cg, stride, offset   (rows = Bt*Hr*Hr of D [rows][Cd]; Gt [Bt*Hi*Hi][Cg])
SHAPES = [(1024, 5, 256, 8, 128, 1, 0), (1024, 8, 128, 16, 64, 2, -1), (1024, 16, 64, 32, 32, 2, -1),
          (256, 5, 256, 8, 128, 1, 0), (256, 8, 128, 16, 64, 2, -1), (256, 16, 64, 32, 32, 2, -1)]


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
    tot = {}
    for sh in SHAPES:
        Bt, Hr, Cd, Hi, Cg, stride, offset = sh
        rows = Bt * Hr * Hr
        Dm, Gm = torch.randn(rows, Cd, device=dev), torch.randn(Bt * Hi * Hi, Cg, device=dev)
        Dp, Gp = ops.Planes(rows, Cd, dev), ops.Planes(Bt * Hi * Hi, Cg, dev)
        HIP.split_planes(Dm, Dp)
        HIP.split_planes(Gm, Gp)
        variants = {"fp32": (Dm, Gm), "D": (Dp, Gm), "G": (Dm, Gp), "DG": (Dp, Gp)}
        times, res = {k: [] for k in variants}, {}
        for rnd in range(5):
            for k, (d, g) in variants.items():
                chunks = HIP.wgrad_chunks(1, rows, Cd, Cg, planes=(d is Dp, g is Gp))
                part = torch.empty(chunks, 16, Cd, Cg, device=dev)
                fn = lambda: HIP.wgrad_tn(d, g, part, 1, Bt, Hr, Hr, Cd, Hi, Hi, Cg, stride, offset, chunks)
                if rnd == 0:
                    for _ in range(3):
                        fn()
                    torch.cuda.synchronize()
                    res[k] = part.sum(0)
                times[k].append(event_ms(fn, 10))
        fl = 2.0 * rows * Cd * Cg * 16
        line = f"{str(sh):40s}"
        for k in variants:
            m = statistics.median(times[k])
            tot[k] = tot.get(k, 0.0) + m
            d = float((res[k] - res["fp32"]).norm() / res["fp32"].norm())
            line += f" | {k:4s} {m * 1e3:7.1f} us {fl / m / 1e9:6.1f} TF/s" + ("" if k == "fp32" else f" d{d:.0e}")
        print(line, flush=True)
    print("sums (ms):", {k: round(v, 3) for k, v in tot.items()})


if __name__ == "__main__":
    main()
