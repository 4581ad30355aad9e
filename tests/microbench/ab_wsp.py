#!/usr/bin/env python3
"""Diagnostic: the LARGE fp32 implicit-GEMM launches of one bs=256 train step (and the per-GPU shares of the other configs),
each timed alone on the chip on the one-tile-per-block kernels (MMDYN_WSP=0: igemm_ws.hip / igemm_nt.hip by their rule) and on the
persistent, stream-K-scheduled ring kernel (igemm_wsp.hip), interleaved rounds in ONE process (LAB build of the library).
usage: ab_wsp.py [tile]      tile = "128,64" | "128,128" forces one persistent tile (default: the kernel's rule)"""
import os
import statistics
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
from mmdyn_hip import ops, _lib  # noqa: E402

HIP = ops.HipBackend(lib_path=_lib.LAB_LIB_PATH)
# mode,G,Bg,Hi,Cin,Ho,N,stride,offset, kind ('plain' | 'stats' | 'bnbwd' | 'actbwd')
SHAPES = [
    (4, 4, 256, 5, 256, 8, 128, 1, 0, "stats"),         # decoder layer 1: the k4 s1 p0 transposed convolution (a)
    (4, 4, 128, 5, 256, 8, 128, 1, 0, "stats"),         # ... at the bs 128 share
    (1, 1, 1024, 8, 128, 5, 256, 1, 0, "actbwd"),       # decoder layer-1 input gradient (b)
    (1, 4, 256, 32, 32, 16, 64, 2, -1, "bnbwd"),        # (c)
    (2, 4, 256, 8, 128, 16, 64, 1, 0, "stats"),         # (d)
    (1, 4, 256, 16, 64, 8, 128, 2, -1, "bnbwd"),        # (e)
    (1, 1, 256, 8, 128, 5, 256, 1, 0, "stats"),         # encoder conv4
    (1, 1, 256, 16, 64, 8, 128, 2, -1, "stats"),        # encoder conv3
    (2, 1, 256, 8, 128, 16, 64, 1, 0, "bnbwd"),
    (1, 1, 256, 32, 32, 16, 64, 2, -1, "stats"),
    (1, 4, 128, 16, 64, 8, 128, 2, -1, "bnbwd"),        # bs 128 share
    (2, 4, 128, 8, 128, 16, 64, 1, 0, "stats"),
]


def event_ms(fn, reps):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


TILE = sys.argv[1] if len(sys.argv) > 1 and "," in sys.argv[1] else ""
if len(sys.argv) > 1 and sys.argv[1].startswith("S"):       # "S4": the persistent side with a four-slot ring (LAB experiment)
    os.environ["MMDYN_WSP_S"] = sys.argv[1][1:]


def main():
    dev = "cuda"
    print("persistent tile:", TILE or "rule")
    tot = {"0": 0.0, "1": 0.0}
    for sh in SHAPES:
        mode, G, Bg, Hi, Cin, Ho, N, stride, offset, kind = sh
        Bt = G * Bg
        A = torch.randn(Bt * Hi * Hi, Cin, device=dev)
        Bp = torch.randn(16, N, Cin, device=dev) * 0.1
        rows = Bt * Ho * Ho
        C = torch.empty(rows, N, device=dev)
        y = torch.randn(rows, N, device=dev)
        mean, rstd = torch.randn(G, N, device=dev), torch.rand(G, N, device=dev) + 0.5
        gamma, beta = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
        res, times = {}, {"0": [], "1": []}

        def launch():
            T = HIP.igemm_stat_tiles(mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N) if kind in ("stats", "bnbwd") else 0
            st = torch.empty(G, T, 2, N, device=dev) if T else None
            if kind == "bnbwd":
                return lambda: HIP.igemm_nt_dgrad_bn(A, Bp, C, st, y, mean, rstd, gamma, beta, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N,
                                                     stride, offset)
            if kind == "actbwd":
                return lambda: HIP.igemm_nt_dgrad_act(A, Bp, C, y, 1, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, stride, offset)
            return lambda: HIP.igemm_nt(A, Bp, None, C, None, st, None, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, N, stride, offset, 0, 1)
        for rnd in range(5):
            for flag in ("0", "1"):
                os.environ["MMDYN_WSP"] = flag
                if TILE and flag == "1":
                    os.environ["MMDYN_WSP_TILE"] = TILE if N % int(TILE.split(",")[1]) == 0 else "128,64"
                else:
                    os.environ.pop("MMDYN_WSP_TILE", None)
                fn = launch()
                if rnd == 0:
                    for _ in range(3):
                        fn()
                    torch.cuda.synchronize()
                    res[flag] = C.clone()
                times[flag].append(event_ms(fn, 10))
        fl = 2.0 * rows * N * Cin * (16 if mode == 1 else 4) if mode != 4 else 2.0 * Bt * Hi * Hi * N * 16 * Cin
        m0, m1 = statistics.median(times["0"]), statistics.median(times["1"])
        tot["0"] += m0
        tot["1"] += m1
        err = float((res["0"] - res["1"]).abs().max() / (res["0"].abs().max() + 1e-30))
        print(f"{str(sh):52s} one-tile {m0 * 1e3:7.1f} us {fl / m0 / 1e9:6.1f} TF/s | persistent {m1 * 1e3:7.1f} us {fl / m1 / 1e9:6.1f} TF/s "
              f"| x{m0 / m1:5.2f}  maxdiff {err:.1e}", flush=True)
    print(f"sum one-tile {tot['0']:.3f} ms, persistent {tot['1']:.3f} ms")


if __name__ == "__main__":
    main()
