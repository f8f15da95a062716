"""Summarise memory/LDS/MFMA instruction mix per kernel from a hipcc -save-temps .s file.
usage: python profiles/isa_summary.py file.s [name-substring]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
pat = r"\b(global_load_\w+|global_store_\w+|global_atomic_\w+|buffer_load_\w+|buffer_store_\w+|ds_bpermute_b32|ds_read\w*|ds_write\w*|scratch_\w+|v_mfma_\w+|s_barrier|v_readlane_b32)"
for m in re.finditer(r"^(_Z\S+):.*?\n(.*?)s_endpgm", s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue
    c = collections.Counter(re.findall(pat, body))
    vg = re.search(r"\.vgpr_count:\s+(\d+)", s[s.find(name, m.end()):]) if False else None
    print(name[:110])
    print("   ", dict(sorted(c.items())))
