#!/usr/bin/env python3
"""Diagnostic (LAB build): the weight-gradient GEMMs of the 128 x 64 channel layers on 64x64 tiles (product) against 128x64 / 64x128
tiles (MMDYN_WGRAD_128x64=1), at the product's partial-slab count and at twice that; interleaved rounds after a clock warm-up."""
import os
import statistics
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
from mmdyn_hip import ops, _lib  # noqa: E402

HIP = ops.HipBackend(lib_path=_lib.LAB_LIB_PATH)


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
    x = torch.randn(8192, 8192, device=dev)
    for _ in range(40):                      # ~0.3 s of matrix work: clocks up before anything is timed
        x @ x
    torch.cuda.synchronize()
    shapes = [(1, 1024, 8, 128, 16, 64, 2, -1), (1, 256, 8, 128, 16, 64, 2, -1), (1, 1024, 16, 64, 8, 128, 1, 0)]
    for sh in shapes:
        mode, Bt, Hr, Cd, Hi, Cg, stride, offset = sh
        rows, taps = Bt * Hr * Hr, 16
        Dm = torch.randn(rows, Cd, device=dev)
        Gm = torch.randn(Bt * Hi * Hi, Cg, device=dev)
        base = HIP.wgrad_chunks(mode, rows, Cd, Cg)
        variants = [("0", base), ("1", base), ("1", 2 * base), ("0", 2 * base)]
        parts = {c: torch.empty(c, taps, Cd, Cg, device=dev) for c in {v[1] for v in variants}}
        times, res = {v: [] for v in variants}, {}
        for rnd in range(6):
            for v in variants:
                flag, chunks = v
                os.environ["MMDYN_WGRAD_128x64"] = flag
                fn = lambda: HIP.wgrad_tn(Dm, Gm, parts[chunks], mode, Bt, Hr, Hr, Cd, Hi, Hi, Cg, stride, offset, chunks)
                if rnd == 0:
                    fn()
                    torch.cuda.synchronize()
                    res[v] = parts[chunks].sum(0)
                times[v].append(event_ms(fn, 10))
        fl = 2.0 * rows * Cd * Cg * taps
        ref = res[variants[0]]
        for v in variants:
            m = statistics.median(times[v])
            err = float((res[v] - ref).abs().max() / (ref.abs().max() + 1e-30))
            print(f"wgrad {str(sh):40s} tile {'128x64' if v[0] == '1' else '64x64 '} chunks {v[1]:3d}  {m * 1e3:7.1f} us {fl / m / 1e9:6.1f} TF/s  "
                  f"maxdiff {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
