#!/usr/bin/env python3
"""Instruction histogram per kernel of a hipcc -save-temps .s file (tuning aid)."""
import collections
import sys

lines = open(sys.argv[1]).read().split("\n")
starts = [i for i, l in enumerate(lines) if l.startswith("_Z") and ":" in l.split(";")[0] and "@" in l]
for st in starts:
    name = lines[st].split(":")[0]
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    end = next(i for i in range(st, len(lines)) if "s_endpgm" in lines[i] or "s_setpc_b64 s[30:31]" in lines[i])
    ins = [l.strip() for l in lines[st + 1:end] if l.strip() and not l.strip().startswith((".", ";", "//")) and not l.split(";")[0].strip().endswith(":")]
    c = collections.Counter(l.split()[0] for l in ins)
    v = sum(n for k, n in c.items() if k.startswith("v_"))
    s = sum(n for k, n in c.items() if k.startswith("s_"))
    mem = {k: n for k, n in sorted(c.items()) if k.startswith(("global_", "flat_", "ds_", "scratch_", "buffer_"))}
    print("%-60s total %5d valu %5d salu %5d" % (name[:60], len(ins), v, s), mem)
