#!/usr/bin/env python3
"""Average kernel durations (us) from a rocprofv3 --kernel-trace CSV, skipping the first `skip` calls of each kernel."""
import csv, sys, collections, re
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 5
d = collections.defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "")
        name = re.sub(r"\(.*", "", name)
        d[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if n.startswith("at::") or n.startswith("__amd") or n.startswith("Cijk_"):   # (torch's own kernels: the synthetic frames are generated on the device)
        continue
    vv = v[skip:] if len(v) > skip else v
    avg = sum(vv) / len(vv)
    tot += avg
    print(f"{avg:9.2f} us  x{len(v):4d}  {n}")
print(f"{tot:9.2f} us  sum of averages")
