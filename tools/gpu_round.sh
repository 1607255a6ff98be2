#!/bin/bash
# Runs ON THE GPU BOX: one development round - GPU tests, recorded parity errors, the stamped probe, a full bench line.
# Every step bounded by `timeout`; steps joined so that a failed GPU step starts no further one.
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${1:-dev}"
O="$R/gpurun_out/${CVF_ROUND:-r4}"
mkdir -p "$O"
cd "$R"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$O/gpu_tests_$TAG.log" 2>&1; rc=$?
tail -4 "$O/gpu_tests_$TAG.log"
[ $rc -ne 0 ] && { echo "gpu tests failed rc=$rc"; grep -n "Error\|FAILED\|assert" "$O/gpu_tests_$TAG.log" | head -30; exit 1; }
timeout -k 10 600 python tools/parity_errors.py "$TAG" > "$O/parity_$TAG.log" 2>&1 || { echo "parity_errors failed"; tail -20 "$O/parity_$TAG.log"; exit 1; }
tail -2 "$O/parity_$TAG.log"
cp "$R/gpurun_out/${TAG}_parity_errors.json" "$O/" 2>/dev/null
if [ -x tools/build/ef16_probe ]; then timeout -k 10 120 tools/build/ef16_probe > "$O/ef16_probe_$TAG.log" 2>&1 || { echo "probe failed"; tail -5 "$O/ef16_probe_$TAG.log"; exit 1; }; head -12 "$O/ef16_probe_$TAG.log"; fi
timeout -k 10 500 python bench.py > "$O/bench_$TAG.log" 2> "$O/bench_$TAG.err" || { echo "bench failed"; tail -5 "$O/bench_$TAG.err"; exit 1; }
python - "$O/bench_$TAG.log" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("step us", d["ms_per_step"]*1e3, "value", d["value"], "kern", d["kernel_avg_us"], "roof", d["roofline"]["avg_launch_us"], d["roofline"]["frac"])
print("scaling", {k:(v["ms_per_step"]) for k,v in d.get("scaling_table",{}).get("rows",{}).items()})
r=d.get("roofline_align_feature",{})
for k,v in r.get("cases",{}).items(): print(k, "feat-only", v["features_only"]["avg_launch_us"], round(v["features_only"]["frac"],3), "gen", v["generator_outputs"]["avg_launch_us"], round(v["generator_outputs"]["frac"],3))
print("cpu", d.get("cpu_baseline",{}).get("value"))
PY
