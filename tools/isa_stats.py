#!/usr/bin/env python3
"""ISA statistics of one kernel of the library: registers / occupancy from the kernel descriptor comments and the instruction mix of
its largest basic block (the steady loop).  Usage: hipcc --save-temps ... ; python tools/isa_stats.py <file.s> <substring of the mangled name> [--mem]"""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
s = open(path).read().split("\n")
start = next(i for i, l in enumerate(s) if l.endswith(":") or ": " in l if re.match(r"^_Z\w+:", l) and key in l)
end = next(i for i in range(start, len(s)) if s[i].startswith(".Lfunc_end"))
body = s[start:end]
print(body[0][:120])
for l in s[end:end + 80]:
    if re.search(r"NumVgprs|NumAgprs|NumSgprs|Occupancy|LDSByteSize|ScratchSize|TotalNumVgprs", l):
        print("  ", l.strip().lstrip("; "))
labels = [i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)] + [len(body)]
blocks = sorted(((labels[k + 1] - labels[k], labels[k]) for k in range(len(labels) - 1)), reverse=True)
for rank in range(int(sys.argv[4]) if len(sys.argv) > 4 else 1):
    n, st = blocks[rank]
    blk = [l.strip() for l in body[st:st + n]]
    ins = [l for l in blk if l and not l.startswith((".", ";", "//"))]
    cnt = collections.Counter(l.split()[0] for l in ins)
    valu = sum(v for k, v in cnt.items() if k.startswith("v_"))
    print("block #%d: %d instructions, %d VALU, %d DS, %d VMEM, %d SALU/other" % (rank, len(ins), valu, sum(v for k, v in cnt.items() if k.startswith("ds_")),
          sum(v for k, v in cnt.items() if k.startswith(("global_", "buffer_", "flat_"))), sum(v for k, v in cnt.items() if k.startswith("s_"))))
    print("  " + ", ".join("%d %s" % (v, k) for k, v in cnt.most_common(40)))
    if "--mem" in sys.argv:
        for l in ins:
            if re.match(r"(global_|buffer_|flat_|s_waitcnt|s_sleep|s_cbranch|s_barrier|ds_)", l):
                print("     ", l[:110])
