mkdir -p gpurun_out/r1
hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCVF_STAMPS -Iinclude -Icolvars-finder_amd/csrc -Wno-pass-failed tools/front_probe.hip colvars-finder_amd/csrc/stats.hip colvars-finder_amd/csrc/k1_align.hip colvars-finder_amd/csrc/k1_large.hip colvars-finder_amd/csrc/metric_large.hip -o /tmp/front_probe || exit 1
timeout -k 10 120 /tmp/front_probe > gpurun_out/r1/front_probe.log 2>&1 || exit 1
cat gpurun_out/r1/front_probe.log
