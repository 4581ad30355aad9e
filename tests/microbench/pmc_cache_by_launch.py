#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc passes of cache counters per LAUNCH SHAPE: kernel name + grid size + workgroup size + LDS bytes
identify one launch shape of the step (bench.py --no-graph --single-lane, so launches do not overlap).  Prints, per
shape: launches per step, average duration, L2 (TCC) hit rate, L1 -> L2 read requests, their average latency in cycles
(TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ), and the fabric-side read requests.  Every counter directory is one
rocprofv3 run (the TCC block has four counter slots per pass: MI355X_MICROARCH.md, rocprofv3 PMC slots).
usage: pmc_cache_by_launch.py <steps_in_trace> <out.json> <dir> [<dir> ...]"""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_]+)(<[^(]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name


def main():
    steps, out = int(sys.argv[1]), sys.argv[2]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    dur = collections.defaultdict(lambda: [0.0, 0])
    for d in sys.argv[3:]:
        seen = set()
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                key = (short(r["Kernel_Name"]), int(r["Grid_Size"]), int(r["Workgroup_Size"]), int(r.get("LDS_Block_Size", 0) or 0))
                agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[key][r["Counter_Name"]] += 1
                did = (r["Dispatch_Id"], d)
                if did not in seen:
                    seen.add(did)
                    dur[key][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
                    dur[key][1] += 1
    rows = []
    for key, c in agg.items():
        n = max(max(cnt[key].values()), 1)
        per = {k: v / cnt[key][k] for k, v in c.items()}
        hit, miss = per.get("TCC_HIT_sum"), per.get("TCC_MISS_sum")
        req, lat = per.get("TCP_TCC_READ_REQ_sum"), per.get("TCP_TCC_READ_REQ_LATENCY_sum")
        rows.append({"kernel": key[0], "grid": key[1], "wg": key[2], "lds": key[3], "launches_per_step": n / steps,
                     "us": dur[key][0] / max(dur[key][1], 1),
                     "l2_hit_rate": (hit / (hit + miss)) if hit is not None and miss is not None and hit + miss > 0 else None,
                     "tcc_hit": hit, "tcc_miss": miss, "tcp_tcc_read_req": req,
                     "avg_read_latency_cycles": (lat / req) if req and lat is not None else None,
                     "ea_rdreq": per.get("TCC_EA0_RDREQ_sum"), "ea_rdreq_dram": per.get("TCC_EA0_RDREQ_DRAM_sum"),
                     "tcc_req": per.get("TCC_REQ_sum")})
    rows.sort(key=lambda r: -r["us"] * r["launches_per_step"])
    print(f"{'kernel':70s} {'grid':>8s} {'n/step':>6s} {'us':>7s} {'L2 hit':>7s} {'rd req':>10s} {'lat cyc':>8s} {'EA rd':>10s} {'EA dram':>10s}")
    for r in rows[:60]:
        f = lambda v, p: ("-" if v is None else format(v, p))
        print(f"{r['kernel'][:70]:70s} {r['grid']:8d} {r['launches_per_step']:6.1f} {r['us']:7.1f} {f(r['l2_hit_rate'], '7.3f')} "
              f"{f(r['tcp_tcc_read_req'], '10.0f')} {f(r['avg_read_latency_cycles'], '8.0f')} {f(r['ea_rdreq'], '10.0f')} {f(r['ea_rdreq_dram'], '10.0f')}")
    json.dump({"steps": steps, "launch_shapes": rows}, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
