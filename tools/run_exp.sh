mkdir -p gpurun_out/r1
rm -f gpurun_out/r1/k1_exp.log
hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCVF_STAMPS -Iinclude -Icolvars-finder_amd/csrc -Wno-pass-failed tools/k1_probe.hip colvars-finder_amd/csrc/stats.hip colvars-finder_amd/csrc/k1_large.hip colvars-finder_amd/csrc/metric_large.hip -o /tmp/k1_probe || exit 1
echo "== streaming kernel" >> gpurun_out/r1/k1_exp.log
timeout -k 10 120 /tmp/k1_probe 2>&1 | grep -A3 "^B=" >> gpurun_out/r1/k1_exp.log || exit 1
echo "== CVF_K1_NOSTREAM=1" >> gpurun_out/r1/k1_exp.log
CVF_K1_NOSTREAM=1 timeout -k 10 120 /tmp/k1_probe 2>&1 | grep -A3 "^B=" >> gpurun_out/r1/k1_exp.log || exit 1
cat gpurun_out/r1/k1_exp.log
