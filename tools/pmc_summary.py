#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 --pmc passes (counter_collection.csv files).

    python tools/pmc_summary.py <dir with */*/*_counter_collection.csv> [substring filter] [--json out.json]

--json writes {kernel name: HBM bytes per launch} (read + write, corrected as below) for the kernels
that have both a FETCH_SIZE and a WRITE_SIZE pass: bench.py loads profiles/traffic.json from it.

FETCH_SIZE / WRITE_SIZE are reported in bytes per launch with the corrections of
/opt/skills/guides/MI355X_MICROARCH.md (TCC counters count 32 B... see HBM section):
    HBM read bytes  = FETCH_SIZE * 1024 * 2   (kB units, gfx950 reports half)
    HBM write bytes = WRITE_SIZE * 1024
"""
import collections
import csv
import glob
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("xdfm_cin::(anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    m = re.match(r"[\w:]+(<[^()]*>)?", name)
    name = m.group(0) if m and m.group(0) else name
    return name[:90]


def main():
    args = list(sys.argv[1:])
    json_out = None
    if "--json" in args:
        i = args.index("--json")
        json_out = args[i + 1]
        del args[i:i + 2]
    root = args[0]
    filt = args[1] if len(args) > 1 else ""
    traffic = {}
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for path in glob.glob(root + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"]) + " lds=%s" % r.get("LDS_Block_Size", "?")
            if filt not in k:
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k in sorted(acc, key=lambda k: -sum(dur[k])):
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        n = max(len(v) for v in acc[k].values())
        out = ["%-60s n=%d" % (k, n)]
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            traffic[k.split(" lds=")[0]] = round(c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024)
        if "FETCH_SIZE" in c:
            out.append("hbm_read_MB=%.2f" % (c["FETCH_SIZE"] * 1024 * 2 / 1e6))
        if "WRITE_SIZE" in c:
            out.append("hbm_write_MB=%.2f" % (c["WRITE_SIZE"] * 1024 / 1e6))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE"):
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs
            # (32 cycles per v_mfma_f32_32x32x16_bf16): share of SIMD cycles with the matrix core busy
            out.append("mfma_busy_pct=%.1f" % (100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)))
        for n_, v in sorted(c.items()):
            if n_ not in ("FETCH_SIZE", "WRITE_SIZE"):
                out.append("%s=%.4g" % (n_, v))
        print("  ".join(out))
    if json_out:
        import json

        with open(json_out, "w") as f:
            json.dump({"unit": "HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, kB units -> bytes)",
                       "source": root, "kernels": traffic}, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
