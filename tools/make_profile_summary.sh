#!/bin/bash
# Collects the round's rocprofv3 outputs (gpurun_out/$TAG/*) into the tracked summaries under profiles/.
set -e
cd "$(dirname "$0")/.."
TAG="${CVF_PROFILE_TAG:-r4}"
cp gpurun_out/$TAG/kt/bench_kernel_stats.csv profiles/${TAG}_bench_kernel_stats.csv
[ -f gpurun_out/$TAG/kt_k1/k1_kernel_stats.csv ] && cp gpurun_out/$TAG/kt_k1/k1_kernel_stats.csv profiles/${TAG}_k1_roofline_kernel_stats.csv
for b in 2000 16000; do [ -f gpurun_out/$TAG/kt_c5_$b/c5_kernel_stats.csv ] && cp gpurun_out/$TAG/kt_c5_$b/c5_kernel_stats.csv profiles/${TAG}_c5_batch${b}_kernel_stats.csv; done
(cd gpurun_out/$TAG && {
  echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 200 --warmup 20 --cpu-seconds 0   (avg us per launch, first 30 launches skipped)"
  python3 ../../tools/kstats.py kt/bench_kernel_trace.csv 30; echo
  echo "# rocprofv3 --kernel-trace --pmc FETCH_SIZE  (separate pass; KB per launch, uncorrected: gfx950 reports 1/2 of wide streaming reads)"
  python3 ../../tools/pmc_summary.py pmc_fetch/bench_counter_collection.csv; echo
  echo "# rocprofv3 --kernel-trace --pmc WRITE_SIZE (separate pass; KB per launch)"
  python3 ../../tools/pmc_summary.py pmc_write/bench_counter_collection.csv; echo
  echo "# rocprofv3 --kernel-trace --pmc SQ_* (separate pass)"
  python3 ../../tools/pmc_summary.py pmc_sq/bench_counter_collection.csv; echo
  echo "# rocprofv3 --kernel-trace --stats -- python3 tools/bench_k1.py  (align+feature kernel alone, whole-shard launches)"
  python3 ../../tools/kstats.py kt_k1/k1_kernel_trace.csv 3; grep '^{"case"' bench_k1.log; echo
  for b in 2000 16000; do
    if [ -f kt_c5_$b/c5_kernel_trace.csv ]; then
      echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload c5 --batch $b --cpu-seconds 0   (config-5 shape: 5000 atoms, d_r 384, k 6; avg us per launch, first 10 launches skipped)"
      python3 ../../tools/kstats.py kt_c5_$b/c5_kernel_trace.csv 10; echo
    fi
  done
  echo "# python bench.py (full line incl. cpu_baseline)"; tail -1 bench_full.log
} > ../../profiles/${TAG}_bench_summary.txt)
CVF_TAG=$TAG python3 - <<'PY'
import csv, json, collections, re, os
TAG = os.environ["CVF_TAG"]
def mean_counter(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter: continue
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); name = re.sub(r"[<(].*", "", name).replace("void ", "")
        d[name].append(float(r["Counter_Value"]))
    return {k: sum(v)/len(v) for k, v in d.items()}
f = mean_counter(f"gpurun_out/{TAG}/pmc_fetch/bench_counter_collection.csv", "FETCH_SIZE")
w = mean_counter(f"gpurun_out/{TAG}/pmc_write/bench_counter_collection.csv", "WRITE_SIZE")
out = {"_source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes) -- python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-extras; mean per launch, KiB",
       "_correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE counts 128-B read requests as 64 B for 16-B-per-lane streaming loads (MI355X_MICROARCH.md, HBM); other access widths are uncalibrated, Infinity-Cache hits are included",
       "kernels": {k: {"FETCH_SIZE_KiB": f[k], "WRITE_SIZE_KiB": w.get(k, 0.0)} for k in f if not k.startswith("at::") and not k.startswith("__amd") and not k.startswith("Cijk")}}
# the align+feature kernel alone at the config-5 shape (tools/bench_k1_c5.py 20000: two launch flavours per pass - features only,
# then with the generator-mode extras)
try:
    def per_launch(path, counter):
        return [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and "k1_large_" in r["Kernel_Name"]]
    fk = per_launch(f"gpurun_out/{TAG}/pmc_k1_FETCH_SIZE/k1_counter_collection.csv", "FETCH_SIZE")
    wk = per_launch(f"gpurun_out/{TAG}/pmc_k1_WRITE_SIZE/k1_counter_collection.csv", "WRITE_SIZE")
    half = len(fk) // 2
    out["kernels"]["k1_large_kernel"] = {"FETCH_SIZE_KiB": sum(fk[:half]) / half, "WRITE_SIZE_KiB": sum(wk[:half]) / half,
                                               "frames_per_launch": 20000, "flavour": "features only (config-5 shape, 5000 atoms, d_r 384)"}
    out["kernels"]["k1_large_kernel+extras"] = {"FETCH_SIZE_KiB": sum(fk[half:]) / (len(fk) - half), "WRITE_SIZE_KiB": sum(wk[half:]) / (len(wk) - half),
                                                      "frames_per_launch": 20000, "flavour": "+ rotation rows and slot copy"}
except Exception as exc:
    print("no K1 PMC passes:", exc)
json.dump(out, open(f"profiles/{TAG}_pmc_traffic.json", "w"), indent=1)
print({k: (round(v["FETCH_SIZE_KiB"]), round(v["WRITE_SIZE_KiB"])) for k, v in out["kernels"].items()})
PY
