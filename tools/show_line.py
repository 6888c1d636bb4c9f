#!/usr/bin/env python
"""One-line summary of bench.py JSON lines."""
import json
import sys

for path in sys.argv[1:]:
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
        print(path, d["value"], d["unit"], d["ms_per_step"], "ms/step; host enqueue", d["host_enqueue_ms_per_step"], "|",
              d["launch"])
    except Exception as e:  # noqa: BLE001
        print(path, "no line:", e)
