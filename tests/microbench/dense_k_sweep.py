#!/usr/bin/env python3
"""Diagnostic: the FC-level DENSE launches as a function of K (fixed cost of a launch against the cost of a K-step), native fp32 and
fp32x3, each launch alone on the chip, replayed from a HIP graph of 20 launches (no host gaps).  usage: dense_k_sweep.py"""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
from mmdyn_hip import ops, layers  # noqa: E402


def graph_us(fn, n=20, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(reps):
            g.replay()
        b.record(s)
        torch.cuda.synchronize()
    return a.elapsed_time(b) / (n * reps) * 1e3


REAL = [(1024, 256, 6400), (1024, 6400, 256), (256, 6400, 512), (256, 512, 6400), (1024, 512, 512), (1024, 512, 256), (1024, 256, 512)]


def real():
    """the step's own FC-level shapes (rows, K, N) through layers.dense (its split-K rule), fp32x3; MMDYN_HIP_LIB / MMDYN_X3_WS from the
    environment (LAB library)"""
    dev = "cuda"
    ops.B.fp32_split = True
    for rows, K, N in REAL:
        A = torch.randn(rows, K, device=dev)
        Bp = torch.randn(N, K, device=dev) * 0.1
        us = graph_us(lambda: layers.dense(A, Bp, None, rows, K, N))
        print(f"  rows {rows:5d} K {K:5d} N {N:5d}: {us:6.1f} us ({2.0 * rows * N * K / us / 1e6:5.1f} TF)", flush=True)


def main():
    dev = "cuda"
    if len(sys.argv) > 1 and sys.argv[1] == "real":
        return real()
    for x3 in (False, True):
        ops.B.fp32_split = x3
        print("fp32x3" if x3 else "native fp32")
        for rows, N in ((1024, 512), (1024, 6400), (256, 512), (256, 6400)):
            line = f"  rows {rows:5d} N {N:5d}:"
            for K in (32, 64, 128, 256, 512, 1024, 2048, 6400):
                A = torch.randn(rows, K, device=dev)
                Bp = torch.randn(N, K, device=dev) * 0.1
                fn = lambda: layers.dense(A, Bp, None, rows, K, N)
                us = graph_us(fn)
                line += f"  K{K}: {us:6.1f} us ({2.0 * rows * N * K / us / 1e6:5.1f} TF)"
            print(line, flush=True)


if __name__ == "__main__":
    main()
