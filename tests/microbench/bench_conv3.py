#!/usr/bin/env python3
"""Microbenchmark of the 3-channel layer kernels (conv3.hip): forward, input gradient with the BatchNorm backward
epilogue, weight gradient -- time per launch and algorithmic HBM rate, fp32 and bf16 activation storage."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
from mmdyn_hip import ops  # noqa: E402

IM2COL3 = 3


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3     # us


def main():
    dev = "cuda"
    for dt in (torch.float32, torch.bfloat16):
        ops.B.precision = "fp32" if dt == torch.float32 else "bf16s"
        es = 4 if dt == torch.float32 else 2
        for G, Bg in ((1, 256), (4, 256)):
            Bt, rows = G * Bg, G * Bg * 1024
            x = torch.rand(Bt, 3, 64, 64, device=dev)
            Bp = torch.randn(1, 32, 64, device=dev) * 0.1
            C, Ca = torch.empty(rows, 32, device=dev, dtype=dt), torch.empty(rows, 32, device=dev, dtype=dt)
            y = torch.randn(rows, 32, device=dev).to(dt)
            T = ops.B.igemm_stat_tiles(IM2COL3, G, Bg, 64, 64, 64, 32, 32, 32)
            stats = torch.empty(G, T, 2, 32, device=dev)
            mean, rstd = torch.zeros(G, 32, device=dev), torch.ones(G, 32, device=dev)
            gamma, beta = torch.ones(32, device=dev), torch.zeros(32, device=dev)
            t = timeit(lambda: ops.B.igemm_nt(x, Bp, None, C, Ca, None, None, IM2COL3, G, Bg, 64, 64, 64, 32, 32, 32, 32, 1, 0, 1, 1))
            by = x.numel() * 4 + 2 * rows * 32 * es
            print(f"{str(dt):15s} Bt={Bt:5d} fwd+swish      {t:8.1f} us  {by / t / 1e6:7.2f} TB/s")
            t = timeit(lambda: ops.B.igemm_nt_dgrad_bn(x, Bp, C, stats, y, mean, rstd, gamma, beta, IM2COL3, G, Bg, 64, 64, 64,
                                                       32, 32, 32, 1, 0))
            print(f"{str(dt):15s} Bt={Bt:5d} dgrad+bn_bwd    {t:8.1f} us  {by / t / 1e6:7.2f} TB/s")
            chunks = ops.B.wgrad_chunks(IM2COL3, rows, 32, 64)
            partial = torch.empty(chunks, 1, 32, 64, device=dev)
            t = timeit(lambda: ops.B.wgrad_tn(y, x, partial, IM2COL3, Bt, 32, 32, 32, 64, 64, 64, 1, 0, chunks))
            by = x.numel() * 4 + rows * 32 * es + partial.numel() * 4
            print(f"{str(dt):15s} Bt={Bt:5d} wgrad ({chunks:4d})    {t:8.1f} us  {by / t / 1e6:7.2f} TB/s")
    ops.B.precision = "fp32"


if __name__ == "__main__":
    main()
