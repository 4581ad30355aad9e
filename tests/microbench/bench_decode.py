#!/usr/bin/env python3
"""Diagnostic: frame-decode kernel (uint8 256x256x3 -> float32 3x64x64, Pillow-exact) throughput."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
from mmdyn_hip.utils.datasets import FrameDecoder  # noqa: E402


def main():
    dev = "cuda"
    store = torch.randint(0, 256, (4096, 256, 256, 3), dtype=torch.uint8, device=dev)      # 805 MB of frames
    dec = FrameDecoder(256, 256, 64, dev)
    for n in (256, 1024, 4096):
        idx = torch.randperm(4096, device=dev)[:n].to(torch.int32)
        for _ in range(3):
            dec(store, idx)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            dec(store, idx)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) / 20 * 1e3
        byt = n * (256 * 256 * 3 + 3 * 64 * 64 * 4)
        print(f"frames {n:5d}: {us:8.1f} us  {byt / us / 1e3:7.1f} GB/s algorithmic  {n / us:6.2f} frames/us")


if __name__ == "__main__":
    main()
