"""Condensed issue order of a loop body: python3 scratch/isa_loop_order.py <file.s> <kernel-substring> <first> <last> (line offsets printed by isa_loop_mix.py)"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and sys.argv[2] in l)
body = lines[start + int(sys.argv[3]):start + int(sys.argv[4]) + 1]
out = []
for l in body:
    m = re.match(r"^\s+([a-z_0-9]+)\s*(.*)", l)
    if not m or l.lstrip().startswith((".", ";")): continue
    op = m.group(1)
    if op.startswith("s_waitcnt"): out.append("WAIT[" + m.group(2).split(";")[0].strip() + "]")
    elif op.startswith("v_mfma"): out.append("M")
    elif op.startswith("ds_read"): out.append("r")
    elif op.startswith("ds_write"): out.append("w")
    elif op.startswith(("buffer_load", "global_load")): out.append("G")
    elif op.startswith("s_barrier"): out.append("BAR")
    elif op.startswith("v_"): out.append("v")
    elif op.startswith("s_"): out.append("s")
res = []; prev = None; n = 0
for o in out + [None]:
    if o == prev: n += 1
    else:
        if prev: res.append(prev + (f"x{n}" if n > 1 else ""))
        prev = o; n = 1
print(" ".join(res))
