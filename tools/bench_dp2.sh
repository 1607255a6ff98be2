#!/bin/bash
# Runs ON THE GPU BOX: the driver's N > 1 command shape rehearsed with TWO ranks on the one GPU (gloo rendezvous), the cross-rank sums
# folded into the finishing launch / slab reduction over the peer-to-peer windows (CVF_COMM=p2p), then as separate launches.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; O="$R/gpurun_out/${CVF_ROUND:-r4}"; mkdir -p "$O"; cd "$R"
for mode in fused separate group; do
  export CVF_BENCH_ONE_GPU=1 CVF_BENCH_BACKEND=gloo
  case $mode in fused) export CVF_COMM=p2p CVF_FUSED_COMM=1;; separate) export CVF_COMM=p2p CVF_FUSED_COMM=0;; group) export CVF_COMM=rccl;; esac
  timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 5 --frames-total 200000 --global-batch 40000 --no-extras --cpu-seconds 0 2> "$O/bench_dp2_$mode.err" | tail -1 > "$O/bench_dp2_$mode.log" || { tail -5 "$O/bench_dp2_$mode.err"; exit 1; }
  python - "$O/bench_dp2_$mode.log" $mode <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
print(sys.argv[2], "n", d["n_gpus"], d["scaling"], "us/step", round(d["ms_per_step"] * 1e3, 1), "launches", d["launches_per_step"]["total"], d["launches_per_step"]["cross_rank_sums"],
      "graphs", d["hip_graph"], {k: round(v, 1) for k, v in d["kernel_avg_us"].items()})
PY
done
