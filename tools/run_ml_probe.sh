#!/bin/bash
# Runs ON THE GPU BOX: tools/build/metric_large_probe (built here with the phase stamps, see the header of
# tools/metric_large_probe.hip) at two batch sizes; output to gpurun_out/r3/$1.
mkdir -p gpurun_out/r3
for W in ${WAVES:-8 16}; do
for B in 2000 16000; do
  echo "== CVF_METRIC_WAVES=$W B=$B"
  CVF_METRIC_WAVES=$W timeout -k 10 120 tools/build/metric_large_probe $B || exit 1
done
done > gpurun_out/r3/$1 2>&1
cat gpurun_out/r3/$1
