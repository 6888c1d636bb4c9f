#!/usr/bin/env python
"""Per-kernel durations from a rocprofv3 --kernel-trace CSV, keeping launches of the same
kernel apart by their position inside a step (e.g. CIN layer 1 / layer 2 share a kernel).

    python tools/trace_summary.py <*_kernel_trace.csv> [period]
"""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("xdfm_cin::(anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    m = re.match(r"[\w:]+(<[^()]*>)?", name)
    return m.group(0) if m and m.group(0) else name


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    per = collections.defaultdict(list)
    for r in rows:
        per[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    # launches per step = count / steps; steps inferred from the most common small count
    counts = collections.Counter(len(v) for v in per.values())
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else counts.most_common(1)[0][0]
    print("steps inferred:", steps)
    out = []
    for k, v in per.items():
        n = len(v)
        if n % steps == 0 and n // steps > 1 and n // steps <= 8:
            p = n // steps
            tail = v[-p * (steps // 2):]
            for ph in range(p):
                sel = tail[ph::p]
                out.append((sum(sel) / len(sel) / 1e3, "%s [launch %d/%d]" % (k, ph + 1, p), len(sel)))
        else:
            tail = v[len(v) // 2:]
            out.append((sum(tail) / len(tail) / 1e3, k, len(tail)))
    for us, k, n in sorted(out, reverse=True):
        print("%9.2f us  n=%-5d %s" % (us, n, k))


if __name__ == "__main__":
    main()
