#!/usr/bin/env python
"""Aggregate a rocprofv3 --pmc counter_collection.csv: per kernel name, mean of each counter per dispatch."""
import csv
import collections
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = re.sub(r"\(.*", "", r.get("Kernel_Name", ""))[:70]
    agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in agg.items():
    print(name)
    for c, v in sorted(cs.items()):
        print("   %-28s n=%3d mean=%.4g" % (c, len(v), sum(v) / len(v)))
