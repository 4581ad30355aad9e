#!/usr/bin/env python3
"""Diagnostic (LAB build): the k4 s1 p0 transposed convolution on bf16 operands -- pair walk against one pixel per block
(MMDYN_S1P0_SPLIT=2 / 4) on the launch sizes of the bf16s step; interleaved rounds after a clock warm-up."""
import os
import statistics
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
from mmdyn_hip import ops, _lib  # noqa: E402

HIP = ops.HipBackend(lib_path=_lib.LAB_LIB_PATH)
HIP.precision = "bf16s"
TCONV_S1P0 = 4


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
    for _ in range(40):
        x @ x
    torch.cuda.synchronize()
    for G, Bg in ((1, 128), (4, 128), (1, 256), (4, 256), (1, 64), (4, 64)):
        Bt = G * Bg
        A = torch.randn(Bt * 25, 256, device=dev).to(torch.bfloat16)
        Bp = (torch.randn(16, 128, 256, device=dev) * 0.1).to(torch.bfloat16)
        bias = torch.randn(128, device=dev)
        C = torch.empty(Bt * 64, 128, device=dev, dtype=torch.bfloat16)
        T = HIP.igemm_stat_tiles(TCONV_S1P0, G, Bg, 5, 5, 256, 8, 8, 128)
        stats = torch.zeros(G, T, 2, 128, device=dev)
        fn = lambda: HIP.igemm_nt(A, Bp, bias, C, None, stats, None, TCONV_S1P0, G, Bg, 5, 5, 256, 8, 8, 128, 128, 1, 0, 0, 1)
        times, res = {"2": [], "4": []}, {}
        for rnd in range(6):
            for flag in ("2", "4"):
                os.environ["MMDYN_S1P0_SPLIT"] = flag
                if rnd == 0:
                    fn()
                    torch.cuda.synchronize()
                    res[flag] = (C.float().clone(), stats.sum(1))
                times[flag].append(event_ms(fn, 10))
        fl = 2.0 * Bt * 400 * 256 * 128
        for flag in ("2", "4"):
            m = statistics.median(times[flag])
            dc = float((res[flag][0] - res["2"][0]).abs().max())
            ds = float((res[flag][1] - res["2"][1]).abs().max() / res["2"][1].abs().max())
            print(f"s1p0 bf16 G={G} Bg={Bg}  {'pair walk' if flag == '2' else 'one pixel per block'}  {m * 1e3:7.1f} us {fl / m / 1e9:6.1f} TF/s  "
                  f"maxdiff C {dc:.1e} stats {ds:.1e}", flush=True)


if __name__ == "__main__":
    main()
