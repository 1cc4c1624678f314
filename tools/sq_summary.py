#!/usr/bin/env python3
""" per-kernel averages of the SQ counters of one rocprofv3 --pmc pass (phihip kernels only): python tools/sq_summary.py <dir> """
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
per = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            if "phihip::" not in name:
                continue
            name = re.sub(r"\(.*$", "", name.replace("void ", "").replace("phihip::", ""))
            per[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for name, ctrs in per.items():
    e = {"launches": max(len(v) for v in ctrs.values())}
    for c, v in ctrs.items():
        e[c] = sum(v) / len(v)
    wc = e.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if c in e:
                e[c + "_share_of_wave_cycles"] = round(e[c] / wc, 4)
    out[name] = e
json.dump(dict(sorted(out.items(), key=lambda kv: -kv[1]["launches"])), sys.stdout, indent=1)
print()
