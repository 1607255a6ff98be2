// Developer tool: phase timing of the AutoEncoderTask step kernel (ae16_kernel, config-2 shape) with s_memtime stamps.
//   hipcc -O3 -std=c++17 -fno-slp-vectorize --offload-arch=gfx950 -DCVF_STAMPS -DCVF_STAMP_WPB=2 -Iinclude -Icolvars-finder_amd/csrc -Wno-pass-failed \
//       tools/ae_probe.hip -Lcolvars-finder_amd/colvarsfinder -lcvf_hip -Wl,-rpath,$ORIGIN/../../colvars-finder_amd/colvarsfinder -o tools/build/ae_probe
#include "../colvars-finder_amd/csrc/ae.hip"
#include <cstdio>
#include <random>
#include <vector>

int main() {
  const int dims[8] = {66, 20, 20, 20, 2, 10, 10, 66};
  const int L = 7;
  const int64_t B = 20000, n = 100000;
  cvf_mlp_desc m = {};
  m.n_nets = 1; m.n_layers = L;
  int pos = 0;
  for (int l = 0; l <= L; ++l) m.dims[l] = dims[l];
  for (int l = 0; l < L; ++l) {
    m.act[l] = (l == 3 || l == 6) ? 0 : 1;
    m.w_off[0][l] = pos; pos += dims[l] * dims[l + 1];
    m.b_off[0][l] = pos; pos += dims[l + 1];
  }
  m.n_params = pos;
  std::mt19937 rng(2);
  std::uniform_real_distribution<float> U(-0.3f, 0.3f);
  std::vector<float> theta(pos), feat((size_t)n * 66), w(B, 1.0f);
  for (auto& v : theta) v = U(rng);
  for (auto& v : feat) v = 3 * U(rng);
  std::vector<int64_t> idx(B);
  for (auto& v : idx) v = rng() % n;
  float *dth, *dfeat, *dw, *dscr, *dgrad; int64_t* didx; double* dout2;
  (void)hipMalloc(&dth, pos * 4); (void)hipMalloc(&dfeat, feat.size() * 4); (void)hipMalloc(&dw, B * 4); (void)hipMalloc(&didx, B * 8);
  (void)hipMalloc(&dscr, cvf_ae_scratch_floats(&m, B) * 4); (void)hipMalloc(&dgrad, pos * 4); (void)hipMalloc(&dout2, 16);
  (void)hipMemcpy(dth, theta.data(), pos * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dfeat, feat.data(), feat.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dw, w.data(), B * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(didx, idx.data(), B * 8, hipMemcpyHostToDevice);
  for (int it = 0; it < 5; ++it) {
    int rc = cvf_ae_step(&m, dth, dfeat, didx, B, dw, 1.0 / B, dscr, dout2, dgrad, nullptr, nullptr, nullptr);
    if (rc) { printf("failed: %s\n", cvf_last_error()); return 1; }
  }
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> st(64 * 4096);
  (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8);
  // ae16_kernel (the chain in registers): stamps 18..29
  const int ids[] = {18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29};
  const char* nm[] = {"", "setup (tables, weights -> LDS)", "frame index / weight / input vector", "first layer + image", "hidden layers + images", "last layer + error",
                      "zbar_{L-1} (registers)", "zbar of the hidden layers", "barrier", "gradient tiles of layers L-1 .. 1", "gradient tiles of layer 0", "barrier"};
  const int NS = sizeof(ids) / sizeof(ids[0]);
  for (int wave = 0; wave < CVF_STAMP_WPB; ++wave) {
    std::vector<double> acc(NS, 0.0); int cnt = 0;
    for (int b = 0; b < 313; ++b) {
      const unsigned long long* s = &st[(size_t)((b * CVF_STAMP_WPB + wave) % 4096) * 64];
      bool ok = s[18] != 0;
      for (int i = 1; i < NS; ++i) ok = ok && s[ids[i]] >= s[ids[i - 1]] && s[ids[i]] - s[ids[i - 1]] < 10000000ull;
      if (!ok) continue;
      for (int i = 1; i < NS; ++i) acc[i] += double(s[ids[i]] - s[ids[i - 1]]);
      ++cnt;
    }
    double tot = 0;
    printf("-- ae16_kernel wave %d\n", wave);
    for (int i = 1; i < NS; ++i) { printf("   %-38s %8.0f cycles\n", nm[i], acc[i] / (cnt ? cnt : 1)); tot += acc[i] / (cnt ? cnt : 1); }
    printf("   total %8.0f cycles over %d blocks\n", tot, cnt);
  }
  {   // launch time
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, nullptr);
    for (int it = 0; it < 50; ++it) cvf_ae_step(&m, dth, dfeat, didx, B, dw, 1.0 / B, dscr, dout2, dgrad, nullptr, nullptr, nullptr);
    (void)hipEventRecord(e1, nullptr);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("cvf_ae_step (step kernel + slab sum): %.1f us per call\n", 1e3 * ms / 50);
  }
  return 0;
}
