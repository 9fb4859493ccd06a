#!/usr/bin/env python3
"""Loop-level view of one kernel's ISA (round 5, VERDICT r4 item 1: which loops carry the AGPR / lane spill traffic).

usage: isa_loops.py file.s <substring of the mangled kernel name> [min_instructions]
Lists every backward branch of the function (target label above the branch) = a loop, innermost first, with the number of
instructions in the span and how many of them are spill traffic (v_accvgpr_read/write = VGPR spills parked in AGPRs,
v_readlane / v_writelane = SGPR spills parked in VGPR lanes), MFMAs, fp64 VALU, LDS and global memory instructions."""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(tuple(":",)) or (l.startswith("_Z") and key in l and ": ;" in l))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.match(r"^\s+s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))


def is_inst(l):
    return l.startswith("\t") and not l.lstrip().startswith((".", ";")) and l.strip()


def stats(a, b):
    c = dict(n=0, acc=0, lane=0, mfma=0, f64=0, ds=0, glob=0, salu=0, trans=0, wait=0, nop=0)
    for l in body[a:b + 1]:
        if not is_inst(l):
            continue
        op = l.split()[0]
        c["n"] += 1
        if op.startswith("v_accvgpr"):
            c["acc"] += 1
        elif op.startswith(("v_readlane", "v_writelane")):
            c["lane"] += 1
        elif op.startswith("v_mfma"):
            c["mfma"] += 1
        elif "_f64" in op:
            c["f64"] += 1
            if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_div")):
                c["trans"] += 1
        elif op.startswith("ds_"):
            c["ds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            c["glob"] += 1
        elif op.startswith("s_waitcnt"):
            c["wait"] += 1
        elif op.startswith("s_nop"):
            c["nop"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
    return c


print(f"{key}: {sum(1 for l in body if is_inst(l))} instructions, {len(loops)} backward branches; whole function: {stats(0, len(body) - 1)}")
loops.sort(key=lambda ab: ab[1] - ab[0])
for a, b in loops:
    inner = [(x, y) for (x, y) in loops if x >= a and y <= b and (x, y) != (a, b)]
    c = stats(a, b)
    if c["n"] < minlen:
        continue
    print(f"  lines {start + a + 1:7d}-{start + b + 1:7d}  inst {c['n']:5d}  accvgpr {c['acc']:4d}  lane {c['lane']:4d}  mfma {c['mfma']:3d}  f64 {c['f64']:4d}"
          f"  ds {c['ds']:4d}  glob {c['glob']:3d}  salu {c['salu']:4d}  wait {c['wait']:3d}  nop {c['nop']:3d}  nested {len(inner)}")
