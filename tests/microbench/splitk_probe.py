#!/usr/bin/env python3
"""Diagnostic: dense GEMMs of the step with and without split-K (the split-K form pays a second launch)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
from mmdyn_hip import ops
HIP = ops.HipBackend()
dev = "cuda"
def t(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
for rows, K, N in ((1024, 512, 512), (1024, 256, 512), (1024, 512, 256), (256, 6400, 512), (1024, 6400, 256), (256, 512, 6400)):
    A = torch.randn(rows, K, device=dev); B = torch.randn(N, K, device=dev) * 0.1; bias = torch.randn(N, device=dev)
    C = torch.empty(rows, N, device=dev)
    for sk in (1, 2, 4, 8, 16):
        if K // 32 < sk * 2: continue
        ws = torch.empty(sk, rows, N, device=dev)
        def f():
            if sk == 1:
                HIP.igemm_nt(A, B, bias, C, None, None, None, 0, 1, rows, 1, 1, K, 1, 1, N, N, 1, 0, 0, 1)
            else:
                HIP.igemm_nt(A, B, None, C, None, None, ws, 0, 1, rows, 1, 1, K, 1, 1, N, N, 1, 0, 0, sk)
                HIP.splitk_reduce(ws, bias, C, None, sk, rows, N, 0)
        print(f"{rows}x{K}->{N} splitk {sk:2d}: {t(f):7.1f} us", flush=True)
