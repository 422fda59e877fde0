#!/usr/bin/env python3
"""mean per dispatch of the counters rocprofv3 --pmc left under <out>/<dir>/ (one directory per pass), per kernel:
    python3 tools/pmc_mean.py gpurun_out/x pf pw [--match walk2,k_env]"""
import csv, glob, sys
from collections import defaultdict
args = [a for a in sys.argv[1:] if not a.startswith("--")]
match = [m for a in sys.argv[1:] if a.startswith("--match=") for m in a[8:].split(",")] or ["walk2", "k_env", "k_slot", "k_step"]
out = args[0]
for d in args[1:]:
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"][:48]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        if any(m in k for m in match):
            print(d, k, {c: round(sum(x) / len(x)) for c, x in v.items()}, "dispatches", len(next(iter(v.values()))))
