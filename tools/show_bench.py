#!/usr/bin/env python
"""Print the headline and the per-kernel table of a bench.py JSON line."""
import json
import sys

for path in sys.argv[1:]:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    print(path, d["value"], d["unit"], d["ms_per_step"], "ms/step; kernels", d["kernel_time_ms_per_step"], "ms")
    r = d["roofline"]
    print("  roofline:", r["kernel"], r["achieved"], r["unit"], "frac", r["frac"], "avg_us", r["avg_us"])
    for k, v in d["kernels"].items():
        if v["avg_us"] * v["per_step"] >= 12:
            print("   %-45s %8.2f us x%-3g %s" % (k, v["avg_us"], v["per_step"],
                                                 " ".join("%s=%s" % kv for kv in v.items() if kv[0] not in ("avg_us", "per_step"))))
