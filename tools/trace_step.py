#!/usr/bin/env python
"""Timeline of one replayed step from a rocprofv3 --kernel-trace CSV, anchored at the gather kernel:
start / end / duration in us, queue, kernel.   python tools/trace_step.py trace.csv [k-th last step] [anchor]
(anchor: the kernel that starts a step; gather_rows for the row-sharded engine)"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 8
anchor = sys.argv[3] if len(sys.argv) > 3 else "embed_fwd_kernel"
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
a, b = idx[-back], idx[-back + 1]
t0 = int(rows[a]["Start_Timestamp"])


def short(n):
    n = re.sub(r"^void ", "", n).replace("xdfm_cin::(anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    m = re.match(r"[\w:]+(<[^()]*>)?", n)
    return m.group(0)[:44]


for r in rows[a - 9:b - 8]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f %8.1f %6.1f q=%s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), short(r["Kernel_Name"])))
