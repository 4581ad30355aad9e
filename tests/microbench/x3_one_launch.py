#!/usr/bin/env python3
"""Diagnostic: ONE implicit-GEMM launch with BatchNorm partial sums, native fp32 against the three-term split (LAB library),
with guard bands around the output and the partial-sum buffer: values, sums and stray writes."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
from mmdyn_hip import ops, _lib  # noqa: E402

HIP = ops.HipBackend(lib_path=_lib.LAB_LIB_PATH)


def main():
    mode, G, Bg, Hi, Cin, Ho, N, stride, offset = [int(v) for v in (sys.argv[1:10] if len(sys.argv) > 9 else "1 1 130 32 32 16 64 2 -1".split())]
    dev = "cuda"
    torch.manual_seed(0)
    Bt, rows = G * Bg, G * Bg * Ho * Ho
    A = torch.randn(Bt * Hi * Hi, Cin, device=dev)
    Bp = torch.randn(16, N, Cin, device=dev) * 0.1
    out = {}
    for flag in ("0", "1"):
        os.environ["MMDYN_X3"] = flag
        T = HIP.igemm_stat_tiles(mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N)
        guard = 4096
        Cbuf = torch.full((rows * N + 2 * guard,), 777.0, device=dev)
        Sbuf = torch.full((G * T * 2 * N + 2 * guard,), 777.0, device=dev)
        C = Cbuf[guard:guard + rows * N].view(rows, N)
        st = Sbuf[guard:guard + G * T * 2 * N].view(G, T, 2, N)
        HIP.igemm_nt(A, Bp, None, C, None, st, None, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, N, stride, offset, 0, 1)
        torch.cuda.synchronize()
        stray = int((Cbuf[:guard] != 777.0).sum() + (Cbuf[-guard:] != 777.0).sum() + (Sbuf[:guard] != 777.0).sum() + (Sbuf[-guard:] != 777.0).sum())
        unwritten = int((C == 777.0).sum()), int((st == 777.0).sum())
        out[flag] = (C.clone(), st.double().sum(1))
        print(f"X3={flag}: T={T} stray guard writes {stray}, unwritten C / stats elements {unwritten}")
    (C0, s0), (C1, s1) = out["0"], out["1"]
    print("C rel diff", float((C0 - C1).norm() / C0.norm()), "max abs", float((C0 - C1).abs().max()))
    print("stat sums rel diff", float((s0 - s1).norm() / s0.norm()))
    ref = C0.double().view(G, -1, N)
    print("stats vs direct sums of C: native", float((s0[:, 0] - ref.sum(1)).norm() / ref.sum(1).norm()),
          "x3", float((s1[:, 0] - C1.double().view(G, -1, N).sum(1)).norm() / ref.sum(1).norm()),
          "squares x3", float((s1[:, 1] - (C1.double().view(G, -1, N) ** 2).sum(1)).norm() / (ref ** 2).sum(1).norm()))


if __name__ == "__main__":
    main()
