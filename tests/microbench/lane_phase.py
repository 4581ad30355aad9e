#!/usr/bin/env python3
"""Diagnostic: two streams each running the chain  GEMM -> BatchNorm+Swish apply -> GEMM -> ...  (the shape of a lane of
the train step).  Does it matter whether the two lanes are IN phase (GEMM next to GEMM, apply next to apply) or in
ANTI-phase (one lane's GEMM next to the other's apply)?  Captured into one HIP graph per lane, replayed concurrently."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multimodal-dynamics_amd"))
from mmdyn_hip import ops  # noqa: E402

SH = (1, 4, 256, 16, 16, 64, 8, 8, 128, 128, 2, -1, 0, 1)      # CONV 16x16x64 -> 8x8x128 on 4 x 256 samples: ~150 us


class Lane:
    def __init__(self, n_apply):
        mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N = SH[:9]
        Bt = G * Bg
        self.A = torch.randn(Bt * Hi * Wi, Cin, device="cuda")
        self.Bp = torch.randn(16, N, Cin, device="cuda") * 0.1
        self.C = torch.empty(Bt * Ho * Wo, N, device="cuda")
        self.a = torch.empty_like(self.C)
        self.mean, self.rstd = torch.zeros(G, N, device="cuda"), torch.ones(G, N, device="cuda")
        self.gamma, self.beta = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
        self.G, self.rows, self.N, self.n_apply = G, Bg * Ho * Wo, N, n_apply

    def gemm(self):
        ops.B.igemm_nt(self.A, self.Bp, None, self.C, None, None, None, *SH)

    def apply(self):
        for _ in range(self.n_apply):
            ops.B.bn_swish_fwd(self.C, self.mean, self.rstd, self.gamma, self.beta, self.a, self.G, self.rows, self.N)

    def chain(self, pairs, start_with_apply):
        if start_with_apply:                      # a lead of ~half a GEMM
            for _ in range(max(1, 4 // self.n_apply)):
                self.apply()
        for _ in range(pairs):
            self.gemm()
            self.apply()


def capture(lane, stream, pairs, start_with_apply):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        lane.chain(pairs, start_with_apply)
    return g


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    pairs = 24
    main_s = torch.cuda.current_stream()
    for n_apply in (1, 2, 3, 4, 5, 6, 8, 3, 1):    # memory-bound time per GEMM: ~9 % of the GEMM's per apply
        la, lb = Lane(n_apply), Lane(n_apply)
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        ga = capture(la, sa, pairs, False)
        gb_in = capture(lb, sb, pairs, False)
        gb_anti = capture(lb, sb, pairs, True)
        g_only = capture(la, sa, pairs, False)

        def solo():
            with torch.cuda.stream(sa):
                sa.wait_stream(main_s)
                g_only.replay()
                main_s.wait_stream(sa)

        def both(gb):
            def run():
                ev = main_s.record_event()
                for s, g in ((sa, ga), (sb, gb)):
                    s.wait_event(ev)
                    with torch.cuda.stream(s):
                        g.replay()
                main_s.wait_stream(sa)
                main_s.wait_stream(sb)
            return run

        t1 = timed(solo)
        t_in, t_anti = timed(both(gb_in)), timed(both(gb_anti))
        # apply-only and gemm-only durations of one lane, for reference
        tg = timed(lambda: [la.gemm() for _ in range(pairs)])
        te = timed(lambda: [la.apply() for _ in range(pairs)])
        print(f"apply x{n_apply}: one lane {t1:6.3f} ms (gemms {tg:6.3f} + applies {te:6.3f});  two lanes in phase {t_in:6.3f} ms, "
              f"lane B half a GEMM ahead {t_anti:6.3f} ms;  serial 2 lanes {2 * t1:6.3f}, gemm floor {2 * tg:6.3f}", flush=True)


if __name__ == "__main__":
    main()
