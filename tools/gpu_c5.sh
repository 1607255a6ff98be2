#!/bin/bash
# Runs ON THE GPU BOX: the large-molecule parity tests, the stamped probe of the derivative kernel, the config-5 workload at two batch sizes.
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${1:-dev}"
O="$R/gpurun_out/${CVF_ROUND:-r4}"
mkdir -p "$O"
cd "$R"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "large or config5 or streaming or fused_metric" > "$O/c5_tests_$TAG.log" 2>&1 || { echo "tests failed"; tail -30 "$O/c5_tests_$TAG.log"; exit 1; }
tail -2 "$O/c5_tests_$TAG.log"
WAVES="${WAVES:-8}" bash tools/run_ml_probe.sh ml_$TAG.log || exit 1
for W in ${WAVES:-8}; do
for B in 2000 16000; do
  CVF_METRIC_WAVES=$W timeout -k 10 300 python bench.py --workload c5 --batch $B --cpu-seconds 0 > "$O/c5_${TAG}_w${W}_b$B.log" 2> "$O/c5_${TAG}_w${W}_b$B.err" || { echo "bench failed"; tail -5 "$O/c5_${TAG}_w${W}_b$B.err"; exit 1; }
  python - "$O/c5_${TAG}_w${W}_b$B.log" $W $B <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("W",sys.argv[2],"B",sys.argv[3],"step us", round(d["ms_per_step"]*1e3,1), {k:round(v,1) for k,v in (d.get("kernel_avg_us") or {}).items()})
PY
done
done
