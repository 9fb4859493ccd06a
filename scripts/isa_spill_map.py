#!/usr/bin/env python3
"""Where a kernel's spill traffic comes from, by SOURCE LINE (round 5).

usage: isa_spill_map.py file_g.s <kernel-name substring> [top]
file_g.s = the ISA compiled with -gline-tables-only (codegen unchanged, `.loc file line col` directives added).
Every instruction is attributed to the last .loc in front of it; prints, per source file:line (with the enclosing function
name when the line tables name an inlined-at chain is not available, the plain line), the instruction count and the spill
instructions (v_accvgpr_* = VGPR spills parked in AGPRs; v_readlane / v_writelane = SGPR spills parked in VGPR lanes)."""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split("\n")
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
    if m:
        files[int(m.group(1))] = m.group(2)
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and ": ;" in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
cur = ("?", 0)
n = collections.Counter()
acc = collections.Counter()
lane = collections.Counter()
for l in lines[start:end]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    if not l.startswith("\t") or l.lstrip().startswith((".", ";")) or not l.strip():
        continue
    op = l.split()[0]
    n[cur] += 1
    if op.startswith("v_accvgpr"):
        acc[cur] += 1
    elif op.startswith(("v_readlane", "v_writelane")):
        lane[cur] += 1
print(f"{key}: {sum(n.values())} instructions, {sum(acc.values())} v_accvgpr, {sum(lane.values())} lane moves")
print("-- by v_accvgpr --")
for k, v in acc.most_common(top):
    print(f"  {k[0]}:{k[1]:5d}  accvgpr {v:4d}  lane {lane[k]:4d}  inst {n[k]:5d}")
print("-- by lane moves --")
for k, v in lane.most_common(top):
    print(f"  {k[0]}:{k[1]:5d}  lane {v:4d}  accvgpr {acc[k]:4d}  inst {n[k]:5d}")
