#!/usr/bin/env python3
"""Diagnostic: the fp32 implicit-GEMM launches of one bs=256 train step, each timed alone on the chip on the
register-staged kernels (igemm_nt.hip, MMDYN_IGEMM_WS=0) and on the wave-specialised LDS-DMA ring kernels (igemm_ws.hip),
interleaved rounds in ONE process (LAB build of the library: it reads the switch per launch)."""
import os
import statistics
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
from mmdyn_hip import ops, _lib  # noqa: E402

HIP = ops.HipBackend(lib_path=_lib.LAB_LIB_PATH)
# mode,G,Bg,Hi,Cin,Ho,N,stride,offset, kind ('plain' | 'stats' | 'bnbwd'), act, splitk
SHAPES = [
    (0, 1, 65536, 1, 1024, 1, 128, 1, 0, "plain", 0, 1),      # the dense GEMMs of tests/microbench/ws_ring_gemm.hip
    (0, 1, 262144, 1, 512, 1, 64, 1, 0, "plain", 0, 1),
    (0, 1, 65536, 1, 1024, 1, 128, 1, 0, "stats", 0, 1),
    (1, 1, 1024, 8, 128, 5, 256, 1, 0, "plain", 0, 1),
    (1, 4, 256, 32, 32, 16, 64, 2, -1, "bnbwd", 0, 1),
    (1, 4, 256, 16, 64, 8, 128, 2, -1, "bnbwd", 0, 1),
    (2, 4, 256, 8, 128, 16, 64, 1, 0, "stats", 0, 1),
    (1, 1, 256, 8, 128, 5, 256, 1, 0, "stats", 0, 1),
    (0, 1, 6400, 1, 256, 1, 2048, 1, 0, "plain", 0, 1),
    (1, 1, 256, 16, 64, 8, 128, 2, -1, "stats", 0, 1),
    (2, 1, 256, 8, 128, 16, 64, 1, 0, "bnbwd", 0, 1),
    (1, 1, 256, 32, 32, 16, 64, 2, -1, "stats", 0, 1),
    (0, 1, 1024, 1, 256, 1, 6400, 1, 0, "plain", 1, 1),
    (0, 1, 1024, 1, 6400, 1, 256, 1, 0, "plain", 0, 8),
    (0, 1, 256, 1, 6400, 1, 512, 1, 0, "plain", 0, 16),
    (0, 1, 256, 1, 512, 1, 6400, 1, 0, "plain", 0, 1),
    (0, 1, 1024, 1, 512, 1, 512, 1, 0, "plain", 0, 2),
]


def event_ms(fn, reps):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


TILE = sys.argv[1] if len(sys.argv) > 1 else ""


def main():
    dev = "cuda"
    print("ws tile:", TILE or "rule")
    tot = {"0": 0.0, "1": 0.0}
    for sh in SHAPES:
        mode, G, Bg, Hi, Cin, Ho, N, stride, offset, kind, act, splitk = sh
        Bt = G * Bg
        taps = 1 if mode == 0 else 16
        A = torch.randn(Bt * Hi * Hi, Cin, device=dev)
        Bp = torch.randn(taps, N, Cin, device=dev) * 0.1
        rows = Bt * Ho * Ho
        C = torch.empty(rows, N, device=dev)
        Ca = torch.empty(rows, N, device=dev) if act else None
        wsb = torch.empty(splitk, rows, N, device=dev) if splitk > 1 else None
        y = torch.randn(rows, N, device=dev)
        mean, rstd = torch.randn(G, N, device=dev), torch.rand(G, N, device=dev) + 0.5
        gamma, beta = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
        res = {}

        def launch():
            T = HIP.igemm_stat_tiles(mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N) if kind != "plain" else 0
            st = torch.empty(G, T, 2, N, device=dev) if T else None
            if kind == "bnbwd":
                return lambda: HIP.igemm_nt_dgrad_bn(A, Bp, C, st, y, mean, rstd, gamma, beta, mode, G, Bg, Hi, Hi, Cin, Ho, Ho,
                                                     N, stride, offset)
            return lambda: HIP.igemm_nt(A, Bp, None, C, Ca, st, wsb, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, N, stride, offset,
                                        act, splitk)
        times = {"0": [], "1": []}
        for rnd in range(5):
            for flag in ("0", "1"):
                os.environ["MMDYN_IGEMM_WS"] = flag
                if TILE:
                    os.environ["MMDYN_WS_TILE"] = TILE if (TILE != "128,128" or N % 128 == 0) else "128,64"
                fn = launch()
                if rnd == 0:
                    for _ in range(3):
                        fn()
                    torch.cuda.synchronize()
                    res[flag] = C.clone()
                times[flag].append(event_ms(fn, 10))
        fl = 2.0 * rows * N * Cin * (1 if mode == 0 else (16 if mode == 1 else 4))
        m0, m1 = statistics.median(times["0"]), statistics.median(times["1"])
        tot["0"] += m0
        tot["1"] += m1
        err = float((res["0"] - res["1"]).abs().max() / (res["0"].abs().max() + 1e-30)) if splitk == 1 else float("nan")
        print(f"{str(sh):58s} regstage {m0 * 1e3:7.1f} us {fl / m0 / 1e9:6.1f} TF/s | ws {m1 * 1e3:7.1f} us {fl / m1 / 1e9:6.1f} TF/s "
              f"| x{m0 / m1:5.2f}  maxdiff {err:.1e}", flush=True)
    print(f"sum regstage {tot['0']:.3f} ms, ws {tot['1']:.3f} ms")
    # ---- weight-gradient GEMMs (MMDYN_WGRAD_WS=1: the ring form, LAB build only) ----
    WGRAD = [  # mode,Bt,Hr,Cd,Hi,Cg,stride,offset
        (1, 1024, 5, 256, 8, 128, 1, 0), (1, 1024, 8, 128, 16, 64, 2, -1), (1, 256, 5, 256, 8, 128, 1, 0),
        (1, 256, 8, 128, 16, 64, 2, -1), (0, 1024, 1, 6400, 1, 256, 1, 0), (0, 1024, 1, 512, 1, 512, 1, 0),
        (0, 256, 1, 512, 1, 6400, 1, 0), (0, 1024, 1, 512, 1, 256, 1, 0)]
    tot = {"0": 0.0, "1": 0.0}
    for sh in WGRAD:
        mode, Bt, Hr, Cd, Hi, Cg, stride, offset = sh
        taps = 16 if mode == 1 else 1
        rows = Bt * Hr * Hr
        Dm = torch.randn(rows, Cd, device=dev)
        Gm = torch.randn(Bt * Hi * Hi, Cg, device=dev)
        chunks = HIP.wgrad_chunks(mode, rows, Cd, Cg)
        part = torch.empty(chunks, taps, Cd, Cg, device=dev)
        fn = lambda: HIP.wgrad_tn(Dm, Gm, part, mode, Bt, Hr, Hr, Cd, Hi, Hi, Cg, stride, offset, chunks)
        times, res = {"0": [], "1": []}, {}
        for rnd in range(5):
            for flag in ("0", "1"):
                os.environ["MMDYN_WGRAD_WS"] = flag
                if rnd == 0:
                    for _ in range(3):
                        fn()
                    torch.cuda.synchronize()
                    res[flag] = part.sum(0)
                times[flag].append(event_ms(fn, 10))
        fl = 2.0 * rows * Cd * Cg * taps
        m0, m1 = statistics.median(times["0"]), statistics.median(times["1"])
        tot["0"] += m0
        tot["1"] += m1
        err = float((res["0"] - res["1"]).abs().max() / (res["0"].abs().max() + 1e-30))
        print(f"wgrad {str(sh):44s} chunks {chunks:3d} regstage {m0 * 1e3:7.1f} us {fl / m0 / 1e9:6.1f} TF/s | ws {m1 * 1e3:7.1f} us "
              f"{fl / m1 / 1e9:6.1f} TF/s | x{m0 / m1:5.2f}  maxdiff {err:.1e}", flush=True)
    print(f"wgrad sum regstage {tot['0']:.3f} ms, ws {tot['1']:.3f} ms")


if __name__ == "__main__":
    main()
