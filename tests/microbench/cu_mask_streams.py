#!/usr/bin/env python3
"""Experiment: the two-stream igemm microbenchmark with each stream restricted to half of the CUs
(hipExtStreamCreateWithCUMask), in two mask layouts, against two unrestricted streams."""
import ctypes
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
from mmdyn_hip import ops  # noqa: E402

SHAPES = [
    (1, 1, 1024, 8, 8, 128, 5, 5, 256, 256, 1, 0, 0, 1),
    (1, 4, 256, 16, 16, 64, 8, 8, 128, 128, 2, -1, 0, 1),
    (2, 4, 256, 8, 8, 128, 16, 16, 64, 64, 1, 0, 0, 1),
]


def masked_stream(hip, words):
    st = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(words), arr)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(st.value)


def flops(sh):
    mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N = sh[:9]
    return 2.0 * G * Bg * Ho * Wo * N * {0: 1, 1: 16, 2: 4}[mode] * Cin


def main():
    torch.cuda.init()
    torch.zeros(1, device="cuda")
    hip = ctypes.CDLL("libamdhip64.so")
    layouts = {
        "none": None,
        "halves": ([0xFFFFFFFF] * 4 + [0] * 4, [0] * 4 + [0xFFFFFFFF] * 4),
        "even/odd": ([0x55555555] * 8, [0xAAAAAAAA] * 8),
        "alt-bytes": ([0x00FF00FF] * 8, [0xFF00FF00] * 8),
    }
    for name, masks in layouts.items():
        if masks is None:
            streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        else:
            streams = [masked_stream(hip, masks[0]), masked_stream(hip, masks[1])]
        for sh in SHAPES:
            mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N = sh[:9]
            Bt, taps = G * Bg, (1 if mode == 0 else 16)
            bufs = [(torch.randn(Bt * Hi * Wi * Cin, device="cuda"), torch.randn(taps, N, Cin, device="cuda") * 0.1,
                     torch.empty(Bt * Ho * Wo, N, device="cuda")) for _ in range(2)]
            reps = 30

            def run():
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for st in streams:
                    st.wait_event(a)
                for _ in range(reps):
                    for k, st in enumerate(streams):
                        with torch.cuda.stream(st):
                            ops.B.igemm_nt(bufs[k][0], bufs[k][1], None, bufs[k][2], None, None, None, *sh)
                for st in streams:
                    torch.cuda.current_stream().wait_stream(st)
                b.record()
                torch.cuda.synchronize()
                return a.elapsed_time(b) * 1e-3
            run()
            t = run()
            print(f"{name:10s} {sh[:9]}: {2 * flops(sh) * reps / t / 1e12:6.1f} TF/s aggregate", flush=True)


if __name__ == "__main__":
    main()
