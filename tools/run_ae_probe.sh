#!/bin/bash
# Runs ON THE GPU BOX: builds tools/ae_probe.hip with the phase stamps and writes its output to gpurun_out/r2/$1.
mkdir -p gpurun_out/r2
hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCVF_STAMPS -Iinclude -Icolvars-finder_amd/csrc -Wno-pass-failed \
    tools/ae_probe.hip colvars-finder_amd/csrc/stats.hip colvars-finder_amd/csrc/ef_mfma.hip -o /tmp/ae_probe 2>/dev/null || exit 1
timeout -k 10 120 /tmp/ae_probe > gpurun_out/r2/$1 2>&1 || exit 1
head -60 gpurun_out/r2/$1
