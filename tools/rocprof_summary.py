#!/usr/bin/env python3
"""Condense rocprofv3 output (rocpd .db or csv directory) into the small text summaries kept under profiles/."""
import csv
import glob
import sqlite3
import sys
from collections import defaultdict


def from_db(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), max(vgpr_count), max(sgpr_count),"
                       " max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name order by 3 desc").fetchall()
    return rows


def from_csv(d):
    acc = defaultdict(list)
    meta = {}
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            meta[r["Kernel_Name"]] = (r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Grid_Size"), r.get("Workgroup_Size"))
    return [(k, len(v), sum(v), sum(v) / len(v), min(v), max(v)) + meta[k] for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))]


def main():
    src = sys.argv[1]
    rows = from_db(src) if src.endswith(".db") else from_csv(src)
    total = sum(r[2] for r in rows) or 1
    print("# rocprofv3 --kernel-trace --stats summary of: %s" % " ".join(sys.argv[2:]))
    print("%-60s %6s %12s %12s %12s %12s %6s %5s %5s %7s %9s %5s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "sgpr", "lds", "grid", "wg"))
    for r in rows:
        name = r[0].replace("pa::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
        print("%-60s %6d %12.1f %12.1f %12.1f %12.1f %6.2f %5s %5s %7s %9s %5s" % (name, r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total, *r[6:11]))


if __name__ == "__main__":
    main()
