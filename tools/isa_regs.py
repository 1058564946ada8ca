#!/usr/bin/env python3
"""Registers / scratch / occupancy / LDS of every kernel in a --save-temps (or -S) listing.  Usage: python tools/isa_regs.py file.s [substring]"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read().split("\n")
key = sys.argv[2] if len(sys.argv) > 2 else ""
cur, info = None, {}
for l in s:
    m = re.match(r"^(_Z\w+):", l)
    if m:
        cur = m.group(1); info[cur] = {}
    for k in ("NumVgprs", "NumAgprs", "ScratchSize", "Occupancy", "LDSByteSize"):
        m2 = re.match(r"^; %s: (\d+)" % k, l)
        if m2 and cur:
            info[cur][k] = int(m2.group(1))
names = [k for k, v in info.items() if v and "kernel" in k]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
rows = []
for k, d in zip(names, dem):
    d = re.sub(r"\(.*", "", d.replace("(anonymous namespace)::", "")).replace("void ", "")
    if key in d and "rocprim" not in d:
        v = info[k]
        rows.append((d, v.get("NumVgprs"), v.get("NumAgprs"), v.get("ScratchSize"), v.get("Occupancy"), v.get("LDSByteSize")))
print("%-75s %5s %5s %7s %4s %6s" % ("kernel", "vgpr", "agpr", "scratch", "occ", "lds"))
for r in sorted(rows):
    print("%-75s %5s %5s %7s %4s %6s" % r)
