#!/usr/bin/env python
"""Timeline of one training step from a rocprofv3 kernel-trace CSV (start/end in us relative to
the step's embed_fwd kernel, queue id, kernel).   python tools/step_timeline.py trace.csv [k-th last]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3


def short(n):
    n = re.sub(r"^void ", "", n)
    n = n.replace("xdfm_cin::(anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    m = re.match(r"[\w:]+(<[^()]*>)?", n)
    return m.group(0)[:48]


idx = [i for i, r in enumerate(rows) if "embed_fwd" in r["Kernel_Name"]]
a, b = idx[-back], idx[-back + 1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    print("%8.1f %8.1f  q=%s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                     r.get("Queue_Id", "?"), short(r["Kernel_Name"])))
