#!/usr/bin/env python3
"""Static audit of the compiler's assembly of the MFMA kernels: for every kernel that contains v_mfma, count
(a) global loads immediately followed by `s_waitcnt vmcnt(0)` near / inside the K-loop (a serialised fetch),
(b) scratch instructions (spilled or memory-resident private arrays), (c) branches around the MFMA block.
usage: isa_audit.py <file.hip> [more.hip ...]   (run from the repo root; needs hipcc)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def audit(src):
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                        "-S", "--cuda-device-only", src, "-o", f.name], check=True, stderr=subprocess.DEVNULL)
        txt = open(f.name).read().split("\n")
    kern, cur = {}, None
    for line in txt:
        m = re.match(r"^(_Z\S+):\s", line + " ")
        if m and "kernel" in m.group(1):
            cur = m.group(1)
            kern[cur] = []
        elif cur is not None:
            kern[cur].append(line)
            if "s_endpgm" in line:
                cur = None
    flagged = 0
    for name, ls in kern.items():
        idx = [i for i, l in enumerate(ls) if "v_mfma" in l]
        if not idx:
            continue
        first, last = idx[0], idx[-1]
        seq = []
        for i, l in enumerate(ls):
            t = l.strip().split()
            if not t:
                continue
            op = t[0]
            if op.startswith("global_load") or (op == "s_waitcnt" and "vmcnt" in l) or op.startswith("s_cbranch") or op.startswith("scratch"):
                seq.append((i, op, l.strip()))
        serial = sum(1 for a, b in zip(seq, seq[1:]) if a[1].startswith("global_load") and b[1] == "s_waitcnt"
                     and "vmcnt(0)" in b[2] and first <= a[0] <= last)
        scratch = sum(1 for s in seq if s[1].startswith("scratch"))
        branches = sum(1 for s in seq if s[1].startswith("s_cbranch") and first < s[0] < last)
        if serial or scratch or branches > 1:
            flagged += 1
            print(f"  {name[:100]}: load->vmcnt(0) {serial}, scratch {scratch}, branches inside the MFMA span {branches}")
    print(f"{os.path.basename(src)}: {len(kern)} kernels, {flagged} flagged")


if __name__ == "__main__":
    for s in sys.argv[1:]:
        audit(s)
