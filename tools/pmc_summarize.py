#!/usr/bin/env python3
"""Per-kernel HBM traffic from the two rocprofv3 --pmc passes of tools/pmc_bench.sh.

Prints {kernel: {"fetch_kb_raw", "write_kb", "launches"}} (averages per launch).  FETCH_SIZE is reported RAW: on gfx950
it counts 64-byte requests in 32-byte units, i.e. HBM bytes fetched = 2 x fetch_kb_raw KiB (MI355X_MICROARCH.md,
HBM / rocprofv3 section); bench.py applies that correction.  WRITE_SIZE is in KiB as is.
"""
import csv, glob, json, os, sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(lambda: {"fetch": 0.0, "write": 0.0, "nf": 0, "nw": 0})
for ctr, key, cnt in (("FETCH_SIZE", "fetch", "nf"), ("WRITE_SIZE", "write", "nw")):
    files = glob.glob(os.path.join(out, f"pmc_{ctr}", "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != ctr:
                continue
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            acc[name][key] += float(r["Counter_Value"])
            acc[name][cnt] += 1
res = {k: {"fetch_kb_raw": v["fetch"] / max(v["nf"], 1), "write_kb": v["write"] / max(v["nw"], 1), "launches": max(v["nf"], v["nw"])}
       for k, v in acc.items()}
json.dump(res, sys.stdout, indent=0)
