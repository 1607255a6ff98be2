#!/bin/bash
# Runs ON THE GPU BOX: the config-5 workload at 16 000 frames under the backward kernel's sharing switch (CVF_BWD_ZS).
mkdir -p gpurun_out/r3
for z in 1 2 3 4; do for b in ${BATCHES:-16000}; do
  CVF_BWD_ZS=$z timeout -k 10 200 python bench.py --workload c5 --batch $b --cpu-seconds 0 2>/dev/null | tail -1 > gpurun_out/r3/c5_zs_${z}_$b.log
  python - gpurun_out/r3/c5_zs_${z}_$b.log $z $b <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read())
print("zs", sys.argv[2], "B", sys.argv[3], "us/step", round(d["ms_per_step"]*1e3,1), "backward", round(d["kernel_avg_us"]["cvf_ef_backward"],1))
PY
done; done
