"""Rough register-pressure profile of one kernel from its final ISA: for every VGPR the span between its first and last
mention in listing order, and per line the number of spans that cover it.  Ignores control flow (a loop-carried value counts
from its first to its last mention), so it over-approximates -- good enough to see WHERE a kernel's pressure peaks.
usage: python scripts/probes/isa_pressure.py file.s kernel-name-substring [window]"""
import re, sys
txt = open(sys.argv[1]).read()
name = sys.argv[2]
W = int(sys.argv[3]) if len(sys.argv) > 3 else 200
m = re.search(r"\n(_Z\S*%s\S*):" % re.escape(name), txt)
body = txt[m.end():]
body = body[:body.index("s_endpgm")]
lines = body.split("\n")
def regs(c):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", c): out |= set(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", c): out.add(int(a))
    return out
first, last = {}, {}
for i, l in enumerate(lines):
    c = l.split(";")[0]
    for r in regs(c):
        first.setdefault(r, i); last[r] = i
# spans that are re-defined: split at every definition that is not also a use (approximation: dst = first operand)
events = [0] * (len(lines) + 1)
for i, l in enumerate(lines):
    pass
defs = {}
for i, l in enumerate(lines):
    c = l.split(";")[0].strip()
    if not c or c.endswith(":") or c.startswith("."): continue
    ops = c.split(None, 1)
    if len(ops) < 2: continue
    parts = ops[1].split(",")
    d = regs(parts[0]); u = regs(",".join(parts[1:]))
    if ops[0].startswith(("global_store", "ds_write", "scratch_store", "buffer_store", "s_", "global_load_lds")): u |= d; d = set()
    for r in d - u: defs.setdefault(r, []).append(i)
    for r in u | d: last[r] = i
live = [0] * len(lines)
for r in first:
    cuts = sorted(set([first[r]] + defs.get(r, [])))
    # a span from each definition to the last use before the next definition
    uses = [i for i, l in enumerate(lines) if re.search(r"\bv%d\b" % r, l.split(";")[0]) or any(int(a) <= r <= int(b) for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", l.split(";")[0]))]
    for k, c0 in enumerate(cuts):
        c1 = cuts[k + 1] if k + 1 < len(cuts) else len(lines)
        us = [u for u in uses if c0 <= u < c1]
        if us:
            for i in range(c0, us[-1] + 1): live[i] += 1
for i in range(0, len(lines), W):
    seg = live[i:i + W]
    nm = sum("v_mfma" in l for l in lines[i:i + W])
    nb = sum("s_barrier" in l for l in lines[i:i + W])
    print(f"lines {i:5d}-{i + W:5d}: max live {max(seg):3d}  mfma {nm:3d}  barriers {nb}")
