#!/bin/bash
# Runs ON THE GPU BOX: builds tools/ef16_probe.hip with the phase stamps and writes its output to gpurun_out/r2/$1.
set -o pipefail
mkdir -p gpurun_out/r2
hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCVF_STAMPS -DCVF_STAMP_WPB=4 -Iinclude -Icolvars-finder_amd/csrc -Wno-pass-failed \
    tools/ef16_probe.hip colvars-finder_amd/csrc/stats.hip -o /tmp/ef16_probe 2>/dev/null || exit 1
timeout -k 10 120 /tmp/ef16_probe > gpurun_out/r2/$1 2>&1 || exit 1
head -${2:-34} gpurun_out/r2/$1
