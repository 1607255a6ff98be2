#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/run_profiles.sh'): the round's rocprofv3 passes over bench.py and
# tools/bench_k1.py, each pass on its own (counters never together with --stats of another domain), every pass
# bounded by `timeout`.  Outputs land in gpurun_out/<tag>/ (tag: $CVF_PROFILE_TAG, default r2); tools/make_profile_summary.sh turns them into profiles/<tag>_*.
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${CVF_PROFILE_TAG:-r4}"
O="$R/gpurun_out/$TAG"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift; echo "== $name"; timeout -k 10 240 "$@" > "$O/bench_$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$O/bench_$name.log"; exit 1; }; }
rm -rf "$O/kt" "$O/pmc_fetch" "$O/pmc_write" "$O/pmc_sq" "$O/kt_k1"
run kt rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -o bench -- python3 "$R/bench.py" --steps 200 --warmup 20 --cpu-seconds 0 --no-extras
run pmc_fetch rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -o bench -- python3 "$R/bench.py" --steps 20 --warmup 5 --cpu-seconds 0 --no-extras
run pmc_write rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -o bench -- python3 "$R/bench.py" --steps 20 --warmup 5 --cpu-seconds 0 --no-extras
run pmc_sq rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$O/pmc_sq" -o bench -- python3 "$R/bench.py" --steps 20 --warmup 5 --cpu-seconds 0 --no-extras
for b in 2000 16000; do rm -rf "$O/kt_c5_$b"; run kt_c5_$b rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_c5_$b" -o c5 -- python3 "$R/bench.py" --workload c5 --batch $b --cpu-seconds 0; done
echo "== kt_k1"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_k1" -o k1 -- python3 "$R/tools/bench_k1_c5.py" > "$O/bench_k1.log" 2>&1 || { echo "bench_k1_c5 pass failed"; tail -5 "$O/bench_k1.log"; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do echo "== pmc_k1_$c"; timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$O/pmc_k1_$c" -o k1 -- python3 "$R/tools/bench_k1_c5.py" 20000 > "$O/bench_k1_$c.log" 2>&1 || { echo "pmc k1 pass failed"; exit 1; }; done
cd "$R"
echo "== full bench"; timeout -k 10 500 python3 bench.py > "$O/bench_full.log" 2> "$O/bench_full.err" || { echo "bench failed"; tail -5 "$O/bench_full.err"; exit 1; }
tail -1 "$O/bench_full.log" | cut -c1-600
for wl in c2 c5 regae transfer; do timeout -k 10 300 python3 bench.py --workload $wl > "$O/bench_$wl.log" 2>&1 && tail -1 "$O/bench_$wl.log" | cut -c1-400; done
grep '^{"case"' "$O/bench_k1.log"
hipcc -O3 --offload-arch=gfx950 tools/stream_probe.hip -o /tmp/stream_probe && COPY_ONLY=1 timeout -k 10 120 /tmp/stream_probe > "$O/copy_probe.log" 2>&1; cat "$O/copy_probe.log"
