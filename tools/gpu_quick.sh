#!/bin/bash
# Runs ON THE GPU BOX: the probes (kernel times by batch size; stamped phase breakdown) and one full bench line.
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${1:-dev}"
O="$R/gpurun_out/${CVF_ROUND:-r4}"
mkdir -p "$O"
cd "$R"
timeout -k 10 120 tools/build/ef16_time > "$O/ef16_time_$TAG.log" 2>&1 || { echo "time probe failed"; tail -5 "$O/ef16_time_$TAG.log"; exit 1; }
cat "$O/ef16_time_$TAG.log"
timeout -k 10 120 tools/build/ef16_probe > "$O/ef16_probe_$TAG.log" 2>&1 || { echo "probe failed"; tail -5 "$O/ef16_probe_$TAG.log"; exit 1; }
sed -n 6,32p "$O/ef16_probe_$TAG.log"
if [ "$2" != "nobench" ]; then
timeout -k 10 500 python bench.py --cpu-seconds 0 > "$O/bench_$TAG.log" 2> "$O/bench_$TAG.err" || { echo "bench failed"; tail -5 "$O/bench_$TAG.err"; exit 1; }
python - "$O/bench_$TAG.log" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("step us", d["ms_per_step"]*1e3, "value", d["value"], "kern", d["kernel_avg_us"], "roof", d["roofline"]["avg_launch_us"], d["roofline"]["frac"])
print("scaling", {k:(v["ms_per_step"]) for k,v in d.get("scaling_table",{}).get("rows",{}).items()})
r=d.get("roofline_align_feature",{})
for k,v in r.get("cases",{}).items(): print(k, "feat-only", v["features_only"]["avg_launch_us"], round(v["features_only"]["frac"],3), "gen", v["generator_outputs"]["avg_launch_us"], round(v["generator_outputs"]["frac"],3))
PY
fi
