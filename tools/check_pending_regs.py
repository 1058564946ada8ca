#!/usr/bin/env python3
"""Safety check for hand-tracked loads (csrc/const_long.hip.h: vm_load_* / vm_wait): between a `global_load_*` written by inline asm
and the `s_waitcnt vmcnt` that validates it, no instruction may READ (or copy, or spill) the destination register -- the compiler does not
know the load is pending.  Scans the ISA of a kernel (hipcc --save-temps): for every global_load whose destination is later named in
an `s_waitcnt`-carrying inline-asm use... simplification that is sufficient here: for every global_load_{dword,ubyte} destination
register, walk forward until the first s_waitcnt vmcnt and report any instruction that has the register among its SOURCE operands.
Usage: python tools/check_pending_regs.py <file.s> <kernel name substring>"""
import re
import sys

s = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(s) if re.match(r"^_Z\w+:", l) and key in l)
end = next(i for i in range(start, len(s)) if s[i].startswith(".Lfunc_end"))
body = [l.split(";")[0].strip() for l in s[start:end]]
body = [l for l in body if l and not l.startswith(".") or l.startswith(".LBB")]
bad = 0
for i, l in enumerate(body):
    m = re.match(r"global_load_(dword|ubyte)\s+(v\d+),", l)
    if not m:
        continue
    reg = m.group(2)
    for j in range(i + 1, len(body)):
        t = body[j]
        if t.startswith("s_waitcnt") and "vmcnt" in t:
            break
        if t.startswith(".LBB") or t.startswith("s_cbranch") or t.startswith("s_branch"):
            continue
        ops = t.split(None, 1)
        if len(ops) < 2:
            continue
        args = [a.strip() for a in ops[1].split(",")]
        srcs = args[1:] if not ops[0].startswith(("global_store", "ds_write", "scratch_store", "buffer_store")) else args
        for a in srcs:
            regs = set()
            mm = re.match(r"v\[(\d+):(\d+)\]", a)
            if mm:
                regs = {"v%d" % k for k in range(int(mm.group(1)), int(mm.group(2)) + 1)}
            elif re.match(r"v\d+$", a.split()[0] if a else ""):
                regs = {a.split()[0]}
            if reg in regs:
                print("line %d: %s reads %s, pending since line %d (%s)" % (j, t, reg, i, l))
                bad += 1
print("pending-register reads: %d" % bad)
sys.exit(1 if bad else 0)
