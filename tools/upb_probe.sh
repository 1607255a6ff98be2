#!/bin/bash
# Runs ON THE GPU BOX: the headline step with one unit per workgroup against the default (several units per workgroup: fewer workgroups to start)
for v in 1 2 3 5 ""; do
  if [ -n "$v" ]; then export CVF_EF16_UPB=$v; else unset CVF_EF16_UPB; fi
  python bench.py --steps 200 --warmup 20 --no-extras --cpu-seconds 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('upb', '${v:-default}', 'step', round(d['ms_per_step']*1e3,2), {k: round(v,1) for k,v in d['kernel_avg_us'].items()}, 'b2b', round(d['roofline']['avg_launch_us'],2), 'loss', d['final_loss'])"
done
