#!/bin/bash
# Runs ON THE GPU BOX: tools/build/ef16_probe (built in the container: hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCVF_STAMPS
# -DCVF_STAMP_WPB=4 -DCVF_DEV_SHAPES -Iinclude -Icolvars-finder_amd/csrc tools/ef16_probe.hip -Lcolvars-finder_amd/colvarsfinder -lcvf_hip
# -Wl,-rpath,'$ORIGIN/../../colvars-finder_amd/colvarsfinder' -o tools/build/ef16_probe) -> gpurun_out/r4/$1
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 120 tools/build/ef16_probe > gpurun_out/r4/$1 2>&1 || { tail -5 gpurun_out/r4/$1; exit 1; }
head -${2:-34} gpurun_out/r4/$1
