#!/usr/bin/env python3
"""Diagnostic: shader clock and package power (rocm-smi) while ONE implicit-GEMM launch shape runs back to back, native fp32
arithmetic, the three-term split done in the MFMA waves, and the plane-ring kernel on operands split beforehand (LAB library).
usage: clock_under_load.py"""
import os
import subprocess
import sys
import threading
import time
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
from mmdyn_hip import ops, _lib  # noqa: E402

HIP = ops.HipBackend(lib_path=_lib.LAB_LIB_PATH)


def smi():
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
    except Exception as e:      # noqa: BLE001
        return f"rocm-smi failed: {e}"
    keep = [l.strip() for l in out.splitlines() if ("sclk" in l or "Power" in l or "mclk" in l)]
    return " | ".join(keep[:4])


def main():
    dev = "cuda"
    mode, G, Bg, Hi, Cin, Ho, N, stride, offset = 1, 1, 1024, 8, 128, 5, 256, 1, 0        # decoder layer-1 input gradient
    Bt = G * Bg
    A = torch.randn(Bt * Hi * Hi, Cin, device=dev)
    Bp = torch.randn(16, N, Cin, device=dev) * 0.1
    rows = Bt * Ho * Ho
    C = torch.empty(rows, N, device=dev)
    fl = 2.0 * rows * N * Cin * 16
    print("idle:", smi())
    Ap, Bpp = ops.Planes(A.shape[0], Cin, dev), ops.Planes(16 * N, Cin, dev)
    HIP.split_planes(A, Ap)
    HIP.split_planes(Bp, Bpp)
    for flag in ("0", "1", "planes"):
        os.environ["MMDYN_X3"] = "0" if flag == "0" else "1"
        HIP.fp32_split = flag != "0"
        a, b = (Ap, Bpp) if flag == "planes" else (A, Bp)
        fn = lambda: HIP.igemm_nt(a, b, None, C, None, None, None, mode, G, Bg, Hi, Hi, Cin, Ho, Ho, N, N, stride, offset, 0, 1)
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        stop = [False]
        samples = []

        def sampler():
            time.sleep(1.0)
            while not stop[0]:
                samples.append(smi())
                time.sleep(1.0)
        th = threading.Thread(target=sampler)
        th.start()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        n = 0
        t0 = time.time()
        while time.time() - t0 < 5.0:
            for _ in range(200):
                fn()
            n += 200
            torch.cuda.synchronize()
        e.record()
        torch.cuda.synchronize()
        stop[0] = True
        th.join()
        ms = s.elapsed_time(e) / n
        print(f"{ {'0': 'native fp32', '1': 'split in the MFMA waves', 'planes': 'plane-ring kernel'}[flag] }: {ms * 1e3:.1f} us per launch, {fl / ms / 1e9:.1f} TFLOP/s (fp32-equivalent)")
        for x in samples[:4]:
            print("    ", x)


if __name__ == "__main__":
    main()
