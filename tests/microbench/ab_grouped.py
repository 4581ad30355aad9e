#!/usr/bin/env python3
"""Diagnostic: the heads GEMMs of the three encoders (1024 x 512 x 512 each) launched one by one and as ONE grouped launch
(mmdyn_igemm_nt_grouped / mmdyn_wgrad_tn_grouped), with the three- and the four-slot ring (LAB build: MMDYN_WS_DENSE_S)."""
import os
import statistics
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
from mmdyn_hip import ops, _lib, layers  # noqa: E402

HIP = ops.HipBackend(lib_path=_lib.LAB_LIB_PATH)
ops.set_backend(HIP)


def event_ms(fn, reps=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        fn()
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    dev = "cuda"
    G, rows, K, N = 3, 1024, 512, 512
    x, W, b = torch.randn(G * rows, K, device=dev), torch.randn(G, N, K, device=dev) * 0.05, torch.randn(G, N, device=dev)
    y = torch.empty(G * rows, N, device=dev)
    dy = torch.randn(G * rows, N, device=dev)
    chunks = HIP.wgrad_chunks(ops.DENSE, rows, N, K)
    part = torch.empty(chunks, G, N, K, device=dev)
    part1 = torch.empty(chunks, 1, N, K, device=dev)
    fl = 2.0 * G * rows * K * N
    for S in ("3", "4"):
        os.environ["MMDYN_WS_DENSE_S"] = S
        one = lambda: [HIP.igemm_nt(x[g * rows:(g + 1) * rows], W[g], b[g], y[g * rows:(g + 1) * rows], None, None, None, ops.DENSE, 1,
                                    rows, 1, 1, K, 1, 1, N, N, 1, 0, 0, 1) for g in range(G)]
        grp = lambda: HIP.igemm_nt_grouped(x, W, b, y, None, None, G, rows, K, N, 0)
        t1 = statistics.median(event_ms(one) for _ in range(5))
        t2 = statistics.median(event_ms(grp) for _ in range(5))
        print(f"ring slots {S}: forward, three launches {t1 * 1e3:6.1f} us ({fl / t1 / 1e9:5.1f} TF/s) | grouped {t2 * 1e3:6.1f} us ({fl / t2 / 1e9:5.1f} TF/s)")
    w1 = lambda: [HIP.wgrad_tn(dy[g * rows:(g + 1) * rows], x[g * rows:(g + 1) * rows], part1, ops.DENSE, rows, 1, 1, N, 1, 1, K, 1, 0,
                               chunks) for g in range(G)]
    wg = lambda: HIP.wgrad_tn_grouped(dy, x, part, G, rows, N, K, chunks)
    t1 = statistics.median(event_ms(w1) for _ in range(5))
    t2 = statistics.median(event_ms(wg) for _ in range(5))
    print(f"weight gradient ({chunks} chunks), three launches {t1 * 1e3:6.1f} us ({fl / t1 / 1e9:5.1f} TF/s) | grouped {t2 * 1e3:6.1f} us ({fl / t2 / 1e9:5.1f} TF/s)")
    for ch in (4, 16):
        p2 = torch.empty(ch, G, N, K, device=dev)
        t = statistics.median(event_ms(lambda: HIP.wgrad_tn_grouped(dy, x, p2, G, rows, N, K, ch)) for _ in range(5))
        print(f"weight gradient grouped with {ch} chunks: {t * 1e3:6.1f} us ({fl / t / 1e9:5.1f} TF/s)")


if __name__ == "__main__":
    main()
