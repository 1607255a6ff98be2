#!/bin/bash
# Runs ON THE GPU BOX: `bash tools/gpu_tests.sh <tag> [pytest -k expression]` - the GPU tests, verbose and unbuffered into
# gpurun_out/<round>/tests_<tag>.log (gpurun kills a call that writes nothing for 7 minutes), failures and the slowest tests at the end.
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${1:-dev}"
O="$R/gpurun_out/${CVF_ROUND:-r4}"
mkdir -p "$O"
cd "$R"
if [ -n "$2" ]; then K=(-k "$2"); else K=(); fi
( while true; do date +%T >> "$O/.heartbeat_$TAG"; sleep 60; done ) &   # (a single test may think for minutes: the chunked fp64 oracle at 16 000 frames x 5000 atoms)
HB=$!
PYTHONUNBUFFERED=1 timeout -k 10 "${CVF_TEST_TIMEOUT:-1100}" python -u -m pytest tests -m gpu -v --durations=15 -p no:cacheprovider "${K[@]}" > "$O/tests_$TAG.log" 2>&1
rc=$?
kill $HB 2>/dev/null
grep -n "FAILED\|ERROR" "$O/tests_$TAG.log" | head -40
grep -n "^E  " "$O/tests_$TAG.log" | head -60
tail -25 "$O/tests_$TAG.log"
exit $rc
