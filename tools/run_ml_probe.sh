#!/bin/bash
# Runs ON THE GPU BOX: builds tools/metric_large_probe.hip with the phase stamps; output to gpurun_out/r2/$1.
mkdir -p gpurun_out/r2
hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCVF_STAMPS -Iinclude -Icolvars-finder_amd/csrc -Wno-pass-failed \
    tools/metric_large_probe.hip colvars-finder_amd/csrc/stats.hip colvars-finder_amd/csrc/k1_large.hip -o /tmp/metric_large_probe 2>gpurun_out/r2/ml_build.err || { tail -5 gpurun_out/r2/ml_build.err; exit 1; }
timeout -k 10 120 /tmp/metric_large_probe > gpurun_out/r2/$1 2>&1 || { tail -5 gpurun_out/r2/$1; exit 1; }
cat gpurun_out/r2/$1
