#!/usr/bin/env python3
"""hbm_traffic_by_kernel.json (tests/microbench/pmc_by_kernel.py) -> traffic_<workload>.json: the HBM-side bytes of one
train step of one bench.py workload -- whole step and the implicit-GEMM launches (the dominant kernel family) -- WITH
provenance: collection date and the sha-256 over the kernel sources it was taken on (bench.py reports null once they
have changed).
usage: make_traffic_json.py <hbm_traffic_by_kernel.json> <out.json> <workload key> <bench args...>"""
import datetime
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sources_sha256():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "multimodal-dynamics_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    d = json.load(open(sys.argv[1]))
    # (igemm_wsp_fixup_kernel is the second half of a persistent launch with split tiles: its bytes count, its launches do not)
    fam = ("igemm_nt_kernel", "igemm_ws_kernel", "igemm_wsp_kernel", "igemm_wsp3_kernel", "igemm_wsp_fixup_kernel")
    rows = [r for r in d["kernels"] if r["kernel"].startswith(fam)]
    launches = sum(r["launches_per_step"] for r in rows if not r["kernel"].startswith("igemm_wsp_fixup_kernel"))
    fetch = sum(r["fetch_MB_per_step"] for r in rows) * 1e6
    write = sum(r["write_MB_per_step"] for r in rows) * 1e6
    out = {
        "source_note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --no-alt --steps 2 "
                       "--warmup 1 --no-graph --no-cpu-baseline <args> (five steps in the trace: the warm-up, the two timed ones, the two "
                       "profiled passes; per-step figures = totals / the number of adam_kernel launches); FETCH_SIZE doubled (gfx950 correction, "
                       "MI355X_MICROARCH.md HBM section); KiB units; aggregated by tests/microbench/pmc_by_kernel.py "
                       "(profiles/collect_r5.sh traffic)",
        "workload": sys.argv[3],
        "bench_args": sys.argv[4:],
        "kernel": "igemm_wsp3_kernel / igemm_wsp_kernel (+ their fix-up launch) + igemm_ws_kernel + igemm_nt_kernel (every implicit-GEMM launch of the step; the 3-channel layers run "
                  "conv3_nt_kernel, the 32-channel up-sampling layers tconv_patch_kernel: booked apart, as in bench.py's launch count)",
        "sources_sha256": sources_sha256(),
        "collected": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"),
        "launches_per_step": launches,
        "hbm_bytes_per_launch": (fetch + write) / max(launches, 1),
        "whole_step": {"fetch_bytes": d["fetch_MB_per_step"] * 1e6, "write_bytes": d["write_MB_per_step"] * 1e6,
                       "launches": sum(r["launches_per_step"] for r in d["kernels"])},
    }
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps(out, indent=1))
