"""Ordered kernel names of the last replayed step of a rocprofv3 --kernel-trace CSV (who launches what, in which order).
usage: python trace_sequence.py <kernel_trace.csv> [marker kernel = adam_kernel] [which step, counted in marker launches = -1]
(bench.py: the last two steps of a run are its profiled eager passes with spin-kernel pads; a replayed step is e.g. step 4)"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    marker = sys.argv[2] if len(sys.argv) > 2 else "adam_kernel"
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    if len(idx) < 2:
        print("fewer than two marker launches")
        return
    k = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    lo, hi = idx[k - 1] + 1, idx[k] + 1
    t0 = int(rows[lo]["Start_Timestamp"])
    for r in rows[lo:hi]:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        print("%9.1f us  %7.1f us  q%-3s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3,
                                              (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                              r.get("Queue_Id", "?"), name[:110]))


if __name__ == "__main__":
    main()
