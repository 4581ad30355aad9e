#!/usr/bin/env python3
"""Diagnostic on a rocprofv3 --kernel-trace CSV of bench.py: launches per step and, per kernel, calls / average / total per step
inside the graph-replayed (two-lane) steps.  usage: trace_kernels.py <kernel_trace.csv> [steps_to_skip] [top]"""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
    adam = [e for s, e, n in ev if "adam_kernel" in n]
    t0, t1 = adam[skip], adam[-2]
    n = len(adam) - 2 - skip
    c, d = collections.Counter(), collections.Counter()
    for s, e, k in ev:
        if e > t0 and s < t1:
            k = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
            c[k] += 1
            d[k] += e - s
    print(f"{n} steps, {(t1 - t0) / n / 1e6:.3f} ms/step, launches per step {sum(c.values()) / n:.1f}")
    for k, v in d.most_common(top):
        print(f"{k:62s} {c[k] / n:6.1f} x {v / c[k] / 1e3:8.1f} us = {v / n / 1e3:8.1f} us/step")


if __name__ == "__main__":
    main()
