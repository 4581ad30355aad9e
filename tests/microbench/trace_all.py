"""All kernel launches of a rocprofv3 --kernel-trace CSV between two points of the run, in order (start, duration, queue, name).
usage: python trace_all.py <kernel_trace.csv> <first row from the end, e.g. 120> [count]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2])
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else n
sel = rows[-n:][:cnt]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    print("%9.1f us  %7.1f us  q%-3s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                          r.get("Queue_Id", "?"), name[:110]))
