"""Instruction mix of the hottest loop (the backward-branch body holding the most MFMAs) of each kernel in a
hipcc -S file.  usage: python profiles/isa_loop.py file.s [name-substring]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""


def klass(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")): return "vmem"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_accvgpr"): return "accmov"
    if op.startswith("v_"): return "valu"
    return "other"


for m in re.finditer(r"^(_Z\S+):.*?\n(.*?)s_endpgm", s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue
    lines = body.split("\n")
    labels = {}
    for i, l in enumerate(lines):
        lm = re.match(r"^(\.LBB\d+_\d+):", l)
        if lm: labels[lm.group(1)] = i
    best = None
    for i, l in enumerate(lines):
        bm = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if bm and bm.group(1) in labels and labels[bm.group(1)] < i:
            seg = lines[labels[bm.group(1)]:i + 1]
            n_mfma = sum("v_mfma" in x for x in seg)
            if best is None or n_mfma > best[0]:
                best = (n_mfma, seg)
    if not best: continue
    c = collections.Counter()
    ops = collections.Counter()
    for l in best[1]:
        t = l.strip().split()
        if not t or t[0].startswith((";", ".")) or t[0].endswith(":"): continue
        c[klass(t[0])] += 1
        if klass(t[0]) in ("valu", "salu"): ops[t[0]] += 1
    print(name[:120])
    print("   loop lines", len(best[1]), dict(c))
    print("   top valu/salu:", ops.most_common(12))
