#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel name."""
import csv, sys, collections, re
rows = collections.defaultdict(lambda: collections.defaultdict(list))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        rows[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in sorted(rows.items()):
    if name.startswith("at::") or name.startswith("__amd") or name.startswith("Cijk_"):
        continue
    print(name)
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} {sum(v)/len(v):16.1f}   (n={len(v)})")
