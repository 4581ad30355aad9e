#!/usr/bin/env python3
"""Diagnostic on a rocprofv3 --kernel-trace CSV of bench.py (two lanes, graph replay): per step, how long at least one
MFMA kernel (igemm_nt / wgrad_tn / conv3) is running, how long only other kernels run, how long nothing runs, and
which non-MFMA kernels account for the MFMA-free time.
CAVEAT (measured): under the profiler the two lanes' graphs start up to 1.6 ms apart, so the trace overstates the
MFMA-free time; without a profiler attached the lanes start and end within 50 us of each other
(tests/microbench/lane_timeline.py, HIP events) and the step is close to the serial sum of its MFMA kernels.
usage: trace_overlap.py <kernel_trace.csv> [steps_to_skip]"""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
    ev.sort()
    # step boundaries: the Adam kernel ends a step
    adam_ends = [e for s, e, n in ev if "adam_kernel" in n]
    if len(adam_ends) < skip + 3:
        print("not enough steps in trace", len(adam_ends))
        return
    # the window: the longest run of consecutive steps whose length is within 1.25x of the median step (bench.py's timed graph
    # replays; the eager profiling passes at the end of the run and the steps around a synchronisation fall outside)
    gaps = [adam_ends[i + 1] - adam_ends[i] for i in range(skip, len(adam_ends) - 1)]
    med = sorted(gaps)[len(gaps) // 2]
    best, cur = (0, 0), None
    for i, gp in enumerate(gaps + [10 ** 18]):
        if gp <= 1.25 * med:
            cur = i if cur is None else cur
        else:
            if cur is not None and i - cur > best[1] - best[0]:
                best = (cur, i)
            cur = None
    t0, t1 = adam_ends[skip + best[0]], adam_ends[skip + best[1]]
    nsteps = best[1] - best[0]
    win = [(max(s, t0), min(e, t1), n) for s, e, n in ev if e > t0 and s < t1]
    is_mfma = lambda n: (("igemm_" in n and "fixup" not in n) or "wgrad_tn" in n or "wgrad_b16" in n or "wgrad_p3" in n
                         or "conv3_" in n or "tconv_patch" in n)
    pts = []
    for s, e, n in win:
        pts.append((s, 1, n))
        pts.append((e, -1, n))
    pts.sort(key=lambda p: (p[0], p[1]))
    active = collections.Counter()
    n_m = n_o = 0
    last = t0
    t_m = t_o = t_idle = 0
    blame = collections.Counter()
    for t, d, n in pts:
        dt = t - last
        if dt > 0:
            if n_m > 0:
                t_m += dt
            elif n_o > 0:
                t_o += dt
                for k, c in active.items():
                    if c > 0:
                        blame[k] += dt / sum(1 for c2 in active.values() if c2 > 0)
            else:
                t_idle += dt
        last = t
        short = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:48]
        if is_mfma(n):
            n_m += d
        else:
            n_o += d
            active[short] += d
    tot = (t1 - t0) / nsteps / 1e6
    print(f"{nsteps} steps, {tot:.3f} ms/step: MFMA kernel running {t_m / nsteps / 1e6:.3f} ms, only other kernels "
          f"{t_o / nsteps / 1e6:.3f} ms, idle {t_idle / nsteps / 1e6:.3f} ms")
    for k, v in blame.most_common(16):
        print(f"   {k:50s} {v / nsteps / 1e3:8.1f} us/step")


if __name__ == "__main__":
    main()
