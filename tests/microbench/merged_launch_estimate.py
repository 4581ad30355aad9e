#!/usr/bin/env python3
"""Diagnostic: what would ONE launch per layer for both modalities buy?

Every MFMA launch of the fp32 bs-256 step exists twice (visual lane, tactile lane; the head / pose GEMMs ten times).  A
grouped launch over both problems has the same geometry with twice the blocks, so its duration is that of the same
launch with the group count (or the batch) doubled.  This times every launch of the step alone on the chip, back to
back, as issued today and as a merged launch, with the LDS-tiled kernels and with the direct-fragment kernels
(MMDYN_D16=1), and prints the two sums.  Geometry only: weights / statistics pointers per group do not change the time.
"""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
from mmdyn_hip import ops  # noqa: E402

# (signature, launches per step)   mode,G,Bg,Hi,Wi,Cin,Ho,Wo,N,ldc,stride,offset,act,splitk
IGEMM = [
    ((4, 4, 256, 5, 5, 256, 8, 8, 128, 128, 1, 0, 0, 1), 2),
    ((1, 1, 1024, 8, 8, 128, 5, 5, 256, 256, 1, 0, 0, 1), 2),
    ((2, 4, 256, 16, 16, 64, 32, 32, 32, 32, 1, 0, 0, 1), 2),
    ((1, 4, 256, 16, 16, 64, 8, 8, 128, 128, 2, -1, 0, 1), 2),
    ((1, 4, 256, 32, 32, 32, 16, 16, 64, 64, 2, -1, 0, 1), 2),
    ((2, 4, 256, 8, 8, 128, 16, 16, 64, 64, 1, 0, 0, 1), 2),
    ((1, 1, 256, 8, 8, 128, 5, 5, 256, 256, 1, 0, 0, 1), 2),
    ((0, 1, 1024, 1, 1, 512, 1, 1, 512, 512, 1, 0, 0, 2), 10),
    ((0, 1, 6400, 1, 1, 256, 1, 1, 2048, 2048, 1, 0, 0, 1), 2),
    ((2, 1, 256, 8, 8, 128, 16, 16, 64, 64, 1, 0, 0, 1), 4),
    ((1, 1, 256, 16, 16, 64, 8, 8, 128, 128, 2, -1, 0, 1), 2),
    ((1, 1, 256, 32, 32, 32, 16, 16, 64, 64, 2, -1, 0, 1), 2),
    ((2, 1, 256, 16, 16, 64, 32, 32, 32, 32, 1, 0, 0, 1), 2),
    ((0, 1, 1024, 1, 1, 256, 1, 1, 6400, 6400, 1, 0, 1, 1), 2),
    ((0, 1, 256, 1, 1, 512, 1, 1, 6400, 6400, 1, 0, 0, 1), 2),
    ((0, 1, 256, 1, 1, 6400, 1, 1, 512, 512, 1, 0, 0, 16), 2),
    ((0, 1, 1024, 1, 1, 6400, 1, 1, 256, 256, 1, 0, 0, 8), 2),
]
# mode,Bt,Hr,Wr,Cd,Hi,Wi,Cg,stride,offset,chunks
WGRAD = [
    ((1, 1024, 5, 5, 256, 8, 8, 128, 1, 0, 24), 2),
    ((1, 1024, 8, 8, 128, 16, 16, 64, 2, -1, 24), 2),
    ((1, 1024, 16, 16, 64, 32, 32, 32, 2, -1, 192), 2),
    ((1, 256, 5, 5, 256, 8, 8, 128, 1, 0, 24), 2),
    ((1, 256, 8, 8, 128, 16, 16, 64, 2, -1, 24), 2),
    ((1, 256, 16, 16, 64, 32, 32, 32, 2, -1, 192), 2),
    ((0, 1024, 1, 1, 6400, 1, 1, 256, 1, 0, 8), 2),
    ((0, 1024, 1, 1, 512, 1, 1, 512, 1, 0, 8), 5),
    ((0, 256, 1, 1, 512, 1, 1, 6400, 1, 0, 4), 2),
]


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
    return s.elapsed_time(e) / reps


def igemm_time(sh):
    mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, ldc, stride, offset, act, splitk = sh
    Bt = G * Bg
    taps = 1 if mode == 0 else 16
    A = torch.randn(Bt * Hi * Wi, Cin, device="cuda")
    Bp = torch.randn(taps, N, Cin, device="cuda") * 0.1
    C = torch.empty(Bt * Ho * Wo, N, device="cuda")
    ws = torch.empty(splitk, Bt * Ho * Wo, N, device="cuda") if splitk > 1 else None
    ms = timeit(lambda: ops.B.igemm_nt(A, Bp, None, C, None, None, ws, *sh))
    if splitk > 1:
        ms += timeit(lambda: ops.B.splitk_reduce(ws, None, C, None, splitk, Bt * Ho * Wo, N, act))
    fl = 2.0 * Bt * (Hi * Wi if mode == 4 else Ho * Wo) * N * Cin * (1 if mode == 0 else (4 if mode == 2 else 16))
    return ms, fl


def wgrad_time(sh):
    mode, Bt, Hr, Wr, Cd, Hi, Wi, Cg, stride, offset, chunks = sh
    taps = 16 if mode == 1 else 1
    D = torch.randn(Bt * Hr * Wr, Cd, device="cuda")
    Gt = torch.randn(Bt * Hi * Wi, Cg, device="cuda")
    part = torch.empty(chunks, taps, Cd, Cg, device="cuda")
    ms = timeit(lambda: ops.B.wgrad_tn(D, Gt, part, *sh))
    return ms, 2.0 * Bt * Hr * Wr * Cd * Cg * taps


def merged_igemm(sh, n):
    sh = list(sh)
    if sh[1] > 1:
        sh[1] *= n          # more groups
    else:
        sh[2] *= n          # one group: more samples (same tiles as a second group)
    return tuple(sh)


def merged_wgrad(sh, n):
    sh = list(sh)
    sh[1] *= n              # twice the rows in twice the slabs = the second problem's blocks
    sh[10] *= n
    return tuple(sh)


def main():
    tag = "d16" if os.environ.get("MMDYN_D16") else "lds"
    tot = {"now": 0.0, "merged": 0.0}
    flops = 0.0
    for sh, cnt in IGEMM:
        n = 2 if cnt % 2 == 0 and cnt < 10 else cnt
        t1, fl = igemm_time(sh)
        t2, _ = igemm_time(merged_igemm(sh, n))
        tot["now"] += cnt * t1
        tot["merged"] += (cnt // n) * t2
        flops += cnt * fl
        print(f"igemm {str(sh):60s} x{cnt:2d}  {t1 * 1e3:7.1f} us {fl / t1 / 1e9:6.1f} TF/s | merged x{n:2d} {t2 * 1e3:7.1f} us "
              f"{n * fl / t2 / 1e9:6.1f} TF/s", flush=True)
    for sh, cnt in WGRAD:
        n = 2 if cnt % 2 == 0 else cnt
        t1, fl = wgrad_time(sh)
        t2, _ = wgrad_time(merged_wgrad(sh, n))
        tot["now"] += cnt * t1
        tot["merged"] += (cnt // n) * t2
        flops += cnt * fl
        print(f"wgrad {str(sh):60s} x{cnt:2d}  {t1 * 1e3:7.1f} us {fl / t1 / 1e9:6.1f} TF/s | merged x{n:2d} {t2 * 1e3:7.1f} us "
              f"{n * fl / t2 / 1e9:6.1f} TF/s", flush=True)
    print(f"[{tag}] MFMA launches of one step, alone on the chip: as issued {tot['now']:.3f} ms ({flops / tot['now'] / 1e9:.1f} TF/s)"
          f"   merged per layer {tot['merged']:.3f} ms ({flops / tot['merged'] / 1e9:.1f} TF/s)")


if __name__ == "__main__":
    main()
