#!/bin/bash
# Runs ON THE GPU BOX: tools/parity_errors.py with a heartbeat (the chunked fp64 oracle at 16 000 frames x 5000 atoms thinks for minutes);
# the JSON lands in profiles/ of the box's copy and is copied to gpurun_out/<round>/ so that it travels back.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${1:-r4}"
O="$R/gpurun_out/$TAG"
mkdir -p "$O"
cd "$R"
( while true; do date +%T >> "$O/.hb_parity"; sleep 60; done ) &
HB=$!
timeout -k 10 1000 python -u tools/parity_errors.py "$TAG" > "$O/parity_errors.log" 2>&1
rc=$?
kill $HB 2>/dev/null
cp "profiles/${TAG}_parity_errors.json" "$O/${TAG}_parity_errors.json"
tail -3 "$O/parity_errors.log"
exit $rc
