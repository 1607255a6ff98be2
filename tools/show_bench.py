#!/usr/bin/env python3
"""Print the interesting numbers of a bench.py JSON line:  python tools/show_bench.py gpurun_out/r4/bench_a.log"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("step us", round(d["ms_per_step"] * 1e3, 2), "value", round(d["value"] / 1e6, 1), "M frames/s; kern", {k: round(v, 1) for k, v in d["kernel_avg_us"].items()},
      "roof", round(d["roofline"]["avg_launch_us"], 2), round(d["roofline"]["frac"], 4), "launches", d.get("launches_per_step", {}).get("total"))
print("scaling", {k: round(v["ms_per_step"], 4) for k, v in d.get("scaling_table", {}).get("rows", {}).items()})
r = d.get("roofline_align_feature")
if r:
    print("K1 head frac", round(r["frac"], 3), "us", round(r["avg_launch_us"], 1), "copy", round(r["copy_GBps"]), "read", round(r["read_GBps"]), "k1/copy",
          round(r["k1_over_copy"], 3), "k1/read", round(r["k1_over_read"], 3))
    print("clocks", r.get("clock_state_at_start"))
    for k, v in r["cases"].items():
        f, g = v["features_only"], v["generator_outputs"]
        print(" ", k, "feat us", round(f["avg_launch_us"], 1), "[", round(f["min_launch_us"], 1), round(f["max_launch_us"], 1), "] frac", round(f["frac"], 3), "copy",
              round(f["copy_GBps"]), "read", round(f["read_GBps"]), "k1/copy", round(f["k1_over_copy"], 3), "| gen us", round(g["avg_launch_us"], 1), "frac",
              round(g["frac"], 3), f.get("kernel_ab_us") or f.get("xcd_placement_ab_us") or "", g.get("kernel_ab_us") or g.get("xcd_placement_ab_us") or "", "rows:", f.get("row_major_output") or "", f.get("launch_decomposition") or "")
for k, v in (d.get("other_configs") or {}).items():
    print(" ", k, "us/step", v.get("us_per_step") and round(v["us_per_step"], 1), "frac", (v.get("roofline") or v.get("roofline_c5_step") or {}).get("frac"),
          {a: round(b, 1) for a, b in (v.get("call_avg_us") or {}).items()}, v.get("error") or "")
print("cpu", (d.get("cpu_baseline") or {}).get("value"))
