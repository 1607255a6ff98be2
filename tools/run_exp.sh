mkdir -p gpurun_out/r1
for e in 0 2; do
hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCVF_STAMPS -DCVF_K1_EXP=$e -Iinclude -Icolvars-finder_amd/csrc -Wno-pass-failed tools/k1_probe.hip colvars-finder_amd/csrc/stats.hip colvars-finder_amd/csrc/k1_large.hip colvars-finder_amd/csrc/metric_large.hip -o /tmp/k1_probe$e || exit 1
echo "== EXP $e" >> gpurun_out/r1/k1_exp.log
timeout -k 10 120 /tmp/k1_probe$e 2>&1 | grep -A10 "B=1000000" >> gpurun_out/r1/k1_exp.log || exit 1
done
cat gpurun_out/r1/k1_exp.log
