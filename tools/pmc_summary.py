#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel name, mean counter value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0][-40:]
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name in sorted(acc):
    print(name)
    for c in sorted(acc[name]):
        v = acc[name][c]
        print("   %-28s n=%4d mean=%16.1f" % (c, len(v), sum(v) / len(v)))
