#!/usr/bin/env python3
"""Instructions between the PA_MARK comments of a kernel's ISA listing (hipcc -S -DPA_ISA_MARKS): the static cost of the
sections of a scheduler iteration. usage: isa_sections.py <file.s> <kernel name substring>"""
import collections
import sys

lines = open(sys.argv[1]).read().split("\n")
st = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sys.argv[2] in l and ":" in l.split(";")[0])
end = next(i for i in range(st, len(lines)) if "s_endpgm" in lines[i])
cur, acc, order = "entry", collections.defaultdict(collections.Counter), ["entry"]
for l in lines[st + 1:end]:
    t = l.strip()
    if "PA_MARK" in t:
        cur = t.split("PA_MARK")[1].strip()
        if cur not in order:
            order.append(cur)
        continue
    if not t or t.startswith((".", ";", "//")) or t.split(";")[0].strip().endswith(":"):
        continue
    op = t.split()[0]
    kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem"
    acc[cur][kind] += 1
for k in order:
    c = acc[k]
    print("after %-16s total %5d  valu %5d salu %5d lds %4d vmem %4d" % (k, sum(c.values()), c["valu"], c["salu"], c["lds"], c["vmem"]))
