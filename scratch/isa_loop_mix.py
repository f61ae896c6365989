"""Instruction mix of the loops of one kernel in a -save-temps .s file: for every backward branch, the instructions between its target label and
the branch, by issue class.  usage: python3 scratch/isa_loop_mix.py <file.s> <mangled-name-substring>"""
import re, sys, collections
lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and sys.argv[2] in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\w+):", l))}
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "lds_read"
    if op.startswith("ds_write") or op.startswith("ds_store"): return "lds_write"
    if op.startswith("ds_"): return "lds_other"
    if op.startswith(("buffer_load", "global_load", "flat_load", "scratch_load")): return "vmem_load"
    if op.startswith(("buffer_store", "global_store", "flat_store", "scratch_store", "global_atomic", "buffer_atomic")): return "vmem_store"
    if op.startswith("v_accvgpr"): return "acc_move"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    return "other"
for i, l in enumerate(body):
    m = re.match(r"^\s+(s_cbranch_\w+|s_branch)\s+(\.LBB\w+)", l)
    if m and m.group(2) in labels and labels[m.group(2)] < i:
        c = collections.Counter()
        for k in range(labels[m.group(2)], i + 1):
            mm = re.match(r"^\s+([a-z_0-9]+)", body[k])
            if mm and not body[k].lstrip().startswith((".", ";")): c[cls(mm.group(1))] += 1
        if c["mfma"]:
            tot = sum(c.values())
            print(f"loop {m.group(2)} (lines {labels[m.group(2)]}..{i}): {tot} instructions, " + ", ".join(f"{k} {v}" for k, v in c.most_common()))
            print(f"    per MFMA: valu {c['valu'] / c['mfma']:.1f}, lds_read {c['lds_read'] / c['mfma']:.2f}, salu {c['salu'] / c['mfma']:.1f}, all non-MFMA {(tot - c['mfma']) / c['mfma']:.1f}")
