#!/usr/bin/env python3
"""Diagnostic (LAB build, VERDICT r3 item 3 iii): the stride-2 convolutions of the step on the one-tile-per-block ring kernel
with the filter taps in raster order (MMDYN_WS_TAPORDER=0) and with the four taps of one input-pixel class back to back (the product's order);
time per launch, interleaved rounds.  Run under rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum for the L2 hit rates."""
import os
import statistics
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
from mmdyn_hip import ops, _lib  # noqa: E402

HIP = ops.HipBackend(lib_path=_lib.LAB_LIB_PATH)
SHAPES = [(1, 4, 256, 16, 64, 8, 128, 2, -1), (1, 4, 256, 32, 32, 16, 64, 2, -1), (1, 1, 256, 16, 64, 8, 128, 2, -1),
          (1, 1, 256, 32, 32, 16, 64, 2, -1)]


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
    os.environ["MMDYN_WSP"] = "0"                       # the one-tile-per-block ring kernel
    os.environ["MMDYN_WS_TILE"] = "64,64"               # ... for every shape (the 262144-row launch is not served by rule)
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    for sh in SHAPES:
        mode, G, Bg, Hi, Cin, Ho, N, stride, offset = sh
        Bt = G * Bg
        A = torch.randn(Bt * Hi * Hi, Cin, device=dev)
        Bp = torch.randn(16, N, Cin, device=dev) * 0.1
        rows = Bt * Ho * Ho
        C = torch.empty(rows, N, device=dev)
        T = HIP.igemm_stat_tiles(mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N)
        st = torch.empty(G, T, 2, N, device=dev)
        fn = lambda: HIP.igemm_nt(A, Bp, None, C, None, st, None, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, N, stride, offset, 0, 1)
        times, res = {"0": [], "1": []}, {}
        for rnd in range(5):
            for flag in ("0", "1"):
                os.environ["MMDYN_WS_TAPORDER"] = flag
                if rnd == 0:
                    for _ in range(3):
                        fn()
                    torch.cuda.synchronize()
                    res[flag] = C.clone()
                times[flag].append(event_ms(fn, reps))
        fl = 2.0 * rows * N * Cin * 16
        m0, m1 = statistics.median(times["0"]), statistics.median(times["1"])
        err = float((res["0"] - res["1"]).abs().max() / (res["0"].abs().max() + 1e-30))
        print(f"{str(sh):44s} raster {m0 * 1e3:7.1f} us {fl / m0 / 1e9:6.1f} TF/s | class-major {m1 * 1e3:7.1f} us {fl / m1 / 1e9:6.1f} TF/s "
              f"| x{m0 / m1:5.2f}  maxdiff {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
