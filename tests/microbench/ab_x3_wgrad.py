#!/usr/bin/env python3
"""Diagnostic (LAB build): the fp32 weight-gradient GEMMs of one bs=256 train step on the native fp32 matrix cores and on the
three-term-split variant (wgrad_tn_kernel X3: bf16 matrix cores, six products), each alone on the chip, interleaved rounds; the
error of both against an fp64 product of the same data for the dense shapes."""
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


# mode, Bt, Hr, Cd, Hi, Cg, stride, offset   (rows = Bt*Hr*Hr of D [rows][Cd]; Gt [Bt*Hi*Hi][Cg])
SHAPES = [
    (1, 1024, 5, 256, 8, 128, 1, 0),
    (1, 1024, 16, 64, 32, 32, 2, -1),
    (1, 1024, 8, 128, 16, 64, 2, -1),
    (1, 256, 5, 256, 8, 128, 1, 0),
    (1, 256, 16, 64, 32, 32, 2, -1),
    (1, 256, 8, 128, 16, 64, 2, -1),
    (0, 1024, 1, 6400, 1, 256, 1, 0),
    (0, 256, 1, 512, 1, 6400, 1, 0),
    (0, 1024, 1, 512, 1, 512, 1, 0),
]


def main():
    dev = "cuda"
    tot = {"0": 0.0, "1": 0.0}
    for sh in SHAPES:
        mode, Bt, Hr, Cd, Hi, Cg, stride, offset = sh
        rows, taps = Bt * Hr * Hr, (16 if mode == 1 else 1)
        Dm = torch.randn(rows, Cd, device=dev)
        Gm = torch.randn(Bt * Hi * Hi, Cg, device=dev)
        chunks = HIP.wgrad_chunks(mode, rows, Cd, Cg)
        part = torch.empty(chunks, taps, Cd, Cg, device=dev)
        times, res = {"0": [], "1": []}, {}
        for rnd in range(5):
            for flag in ("0", "1"):
                os.environ["MMDYN_X3_WGRAD"] = flag
                fn = lambda: HIP.wgrad_tn(Dm, Gm, part, mode, Bt, Hr, Hr, Cd, Hi, Hi, Cg, stride, offset, chunks)
                if rnd == 0:
                    for _ in range(3):
                        fn()
                    torch.cuda.synchronize()
                    res[flag] = part.double().sum(0)
                times[flag].append(event_ms(fn, 10))
        fl = 2.0 * rows * Cd * Cg * taps
        m0, m1 = statistics.median(times["0"]), statistics.median(times["1"])
        tot["0"] += m0
        tot["1"] += m1
        err = float((res["0"] - res["1"]).norm() / (res["0"].norm() + 1e-30))
        note = ""
        if mode == 0:
            ref = Dm.double().t() @ Gm.double()
            note = (f"  rel-L2 vs fp64: native {float((res['0'][0] - ref).norm() / ref.norm()):.2e}, "
                    f"x3 {float((res['1'][0] - ref).norm() / ref.norm()):.2e}")
        print(f"wgrad {str(sh):40s} chunks {chunks:3d} native {m0 * 1e3:7.1f} us {fl / m0 / 1e9:6.1f} TF/s | x3 {m1 * 1e3:7.1f} us "
              f"{fl / m1 / 1e9:6.1f} TF/s | x{m0 / m1:5.2f}  rel diff {err:.1e}{note}", flush=True)
    print(f"sum native {tot['0']:.3f} ms, x3 {tot['1']:.3f} ms")


if __name__ == "__main__":
    main()
