#!/bin/bash
# Runs ON THE GPU BOX: the config-5 workload with and without the side-stream alignment of the next batch (CVF_PIPELINE).
mkdir -p gpurun_out/r3
for p in 0 1; do for b in 2000 16000; do
  CVF_PIPELINE=$p timeout -k 10 200 python bench.py --workload c5 --batch $b --cpu-seconds 0 2>/dev/null | tail -1 > gpurun_out/r3/c5_pipe_${p}_$b.log
  python - gpurun_out/r3/c5_pipe_${p}_$b.log $p $b <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read())
print("pipeline", sys.argv[2], "B", sys.argv[3], "us/step", round(d["ms_per_step"]*1e3,1), "final loss", d.get("final_loss"))
PY
done; done
