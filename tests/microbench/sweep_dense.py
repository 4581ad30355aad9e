#!/usr/bin/env python3
"""Diagnostic: tile / split-K sweep for the small dense GEMMs of the bs=256 step."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
from mmdyn_hip import ops  # noqa: E402

SHAPES = [(256, 512, 6400), (256, 6400, 512), (1024, 512, 512), (1024, 256, 6400), (6400, 256, 2048), (1024, 6400, 256)]
TILES = ["128,128", "128,64", "64,128", "64,64"]


def timeit(fn, reps=30):
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
    for rows, K, N in SHAPES:
        A = torch.randn(rows, K, device=dev)
        Bp = torch.randn(N, K, device=dev) * 0.1
        C = torch.empty(rows, N, device=dev)
        fl = 2.0 * rows * K * N
        print(f"rows {rows} K {K} N {N}")
        for tile in TILES:
            if N % int(tile.split(",")[1]):
                continue
            os.environ["MMDYN_IGEMM_TILE"] = tile
            line = f"  tile {tile:8s}"
            for sk in (1, 2, 3, 4, 8, 16, 32):
                if K // 32 < sk:
                    continue
                ws = torch.empty(sk, rows, N, device=dev) if sk > 1 else None

                def run():
                    ops.B.igemm_nt(A, Bp, None, C, None, None, ws, 0, 1, rows, 1, 1, K, 1, 1, N, N, 1, 0, 0, sk)
                    if sk > 1:
                        ops.B.splitk_reduce(ws, None, C, None, sk, rows, N, 0)
                us = timeit(run)
                line += f"  sk{sk}:{us:6.1f}us/{fl / us / 1e6:5.1f}TF"
            print(line, flush=True)


if __name__ == "__main__":
    main()
