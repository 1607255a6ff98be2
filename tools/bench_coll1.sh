#!/bin/bash
# Runs ON THE GPU BOX: the headline step with and without the data-parallel code path forced on in a one-rank nccl group
# (two real RCCL all-reduces per step inside the captured graphs, unfused Adam) - the fixed cost the multi-GPU step pays on
# top of the single-GPU one before any inter-GPU latency.
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29877
for b in 20000 100000; do
  for mode in plain coll abi p2p fused; do     # coll: RCCL launches; p2p: the windows as separate launches; fused: four launches
    unset CVF_FORCE_COLLECTIVES CVF_COMM CVF_FUSED_COMM
    [ $mode != plain ] && export CVF_FORCE_COLLECTIVES=1
    [ $mode = coll ] && export CVF_COMM=rccl
    [ $mode = abi ] && export CVF_COMM=abi
    [ $mode = p2p ] && export CVF_COMM=p2p CVF_FUSED_COMM=0
    [ $mode = fused ] && export CVF_COMM=p2p
    python bench.py --steps 40 --warmup 5 --no-extras --cpu-seconds 0 --batch $b --frames $((b*5)) 2>/dev/null | tail -1 > /tmp/line.json
    python3 -c "import json; d=json.load(open('/tmp/line.json')); print('$mode', $b, 'us/step', round(d['ms_per_step']*1000,2), 'graphs', d['hip_graph'], 'launches', d['launches_per_step']['total'])"
  done
done
