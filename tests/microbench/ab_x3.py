#!/usr/bin/env python3
"""Diagnostic: fp32 implicit-GEMM launches of one bs=256 train step, each timed alone on the chip on the product's fp32 kernels
(native fp32 matrix cores) and on the three-term-split variant of the register-staged kernel (igemm_nt.hip X3: bf16 matrix cores,
six products, fp32 accumulate), interleaved rounds in ONE process (LAB build of the library).  For the plain convolutions the
error of both against an fp64 convolution of the same data is printed.
usage: ab_x3.py [BM,BN]      forces one X3 block tile (default: 128x128 where N % 128 == 0, else 128x64)"""
import os
import statistics
import sys
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
from mmdyn_hip import ops, _lib  # noqa: E402

HIP = ops.HipBackend(lib_path=_lib.LAB_LIB_PATH)
# mode,G,Bg,Hi,Cin,Ho,N,stride,offset, kind ('plain' | 'stats' | 'bnbwd' | 'actbwd')
SHAPES = [
    (4, 4, 256, 5, 256, 8, 128, 1, 0, "stats"),         # decoder layer 1: the k4 s1 p0 transposed convolution (a)
    (1, 1, 1024, 8, 128, 5, 256, 1, 0, "actbwd"),       # decoder layer-1 input gradient (b)
    (1, 4, 256, 32, 32, 16, 64, 2, -1, "bnbwd"),        # (c)
    (2, 4, 256, 8, 128, 16, 64, 1, 0, "stats"),         # (d)
    (1, 4, 256, 16, 64, 8, 128, 2, -1, "bnbwd"),        # (e)
    (1, 4, 256, 16, 64, 8, 128, 2, -1, "plain"),        # (e) without the epilogue: error against fp64
    (1, 1, 256, 8, 128, 5, 256, 1, 0, "stats"),         # encoder conv4
    (1, 1, 256, 16, 64, 8, 128, 2, -1, "stats"),        # encoder conv3
    (2, 1, 256, 8, 128, 16, 64, 1, 0, "bnbwd"),
    (1, 1, 256, 32, 32, 16, 64, 2, -1, "stats"),
    (2, 4, 256, 16, 64, 32, 32, 1, 0, "stats"),         # 64 -> 32 channel up-sampling layer (native: the patch-resident kernel);
    (2, 1, 256, 16, 64, 32, 32, 1, 0, "actbwd"),        #   on the split only with MMDYN_X3_N32=256|128
    (0, 1, 6400, 1, 256, 1, 2048, 1, 0, "plain"),       # FC level
    (0, 1, 1024, 1, 256, 1, 6400, 1, 0, "plain"),
    (0, 1, 256, 1, 512, 1, 6400, 1, 0, "plain"),
    (0, 3, 1024, 1, 512, 1, 512, 1, 0, "plain"),
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


def main():
    dev = "cuda"
    print("x3 tile:", TILE or "rule")
    os.environ["MMDYN_X3_MIN_BLOCKS"] = os.environ.get("MMDYN_X3_MIN_BLOCKS", "512")
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
                os.environ["MMDYN_X3"] = flag
                if TILE and flag == "1" and N % int(TILE.split(",")[1]) == 0:
                    os.environ["MMDYN_X3_TILE"] = TILE
                else:
                    os.environ.pop("MMDYN_X3_TILE", None)
                fn = launch()
                if rnd == 0:
                    for _ in range(3):
                        fn()
                    torch.cuda.synchronize()
                    res[flag] = C.clone()
                times[flag].append(event_ms(fn, 10))
        fl = 2.0 * rows * N * Cin * ({0: 1, 1: 16, 2: 4}[mode]) if mode != 4 else 2.0 * Bt * Hi * Hi * N * 16 * Cin
        m0, m1 = statistics.median(times["0"]), statistics.median(times["1"])
        tot["0"] += m0
        tot["1"] += m1
        err = float((res["0"] - res["1"]).norm() / (res["0"].norm() + 1e-30))
        note = ""
        if kind == "plain" and mode in (0, 1):
            if mode == 1:
                w = Bp.view(4, 4, N, Cin).permute(2, 3, 0, 1).double()
                ref = F.conv2d(A.view(Bt, Hi, Hi, Cin).permute(0, 3, 1, 2).double(), w, stride=stride, padding=-offset)
                ref = ref.permute(0, 2, 3, 1).reshape(rows, N)
            else:
                ref = A.double() @ Bp[0].double().t()
            e0 = float((res["0"].double() - ref).norm() / ref.norm())
            e1 = float((res["1"].double() - ref).norm() / ref.norm())
            note = f"  rel-L2 vs fp64: native {e0:.2e}, x3 {e1:.2e}"
        print(f"{str(sh):52s} native {m0 * 1e3:7.1f} us {fl / m0 / 1e9:6.1f} TF/s | x3 {m1 * 1e3:7.1f} us {fl / m1 / 1e9:6.1f} TF/s "
              f"| x{m0 / m1:5.2f}  rel diff {err:.1e}{note}", flush=True)
    print(f"sum native {tot['0']:.3f} ms, x3 {tot['1']:.3f} ms")


if __name__ == "__main__":
    main()
