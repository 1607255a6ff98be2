// Developer tool (not part of the library): phase timing of ef_fwd_mfma_kernel with s_memtime stamps.
// Build+run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -DCVF_STAMPS -Iinclude -Icolvars-finder_amd/csrc \
//     tools/stamp_ef.hip colvars-finder_amd/csrc/stats.hip -o /tmp/stamp_ef && /tmp/stamp_ef
#include "../colvars-finder_amd/csrc/ef_mfma.hip"
#include <cstdio>
#include <vector>
#include <random>

int main(int argc, char** argv) {
  // usage: stamp_ef [D] [k] [B]   (defaults: the dipeptide benchmark shape; 384 6 2000 = the config-5 shape)
  const int D = argc > 1 ? atoi(argv[1]) : 66, k = argc > 2 ? atoi(argv[2]) : 3, H = 20, NHl = 3, B = argc > 3 ? atoi(argv[3]) : 20000;
  const int64_t T = (B + 63) / 64;
  cvf_mlp_desc m = {};
  m.n_nets = k; m.n_layers = NHl + 1;
  int dims[5] = {D, H, H, H, 1};
  for (int i = 0; i < 5; ++i) m.dims[i] = dims[i];
  int pos = 0;
  for (int n = 0; n < k; ++n)
    for (int l = 0; l < 4; ++l) {
      m.act[l] = l < 3;
      m.w_off[n][l] = pos; pos += dims[l] * dims[l + 1];
      m.b_off[n][l] = pos; pos += dims[l + 1];
    }
  m.n_params = pos;
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> U(-0.2f, 0.2f);
  std::vector<float> theta(pos), feat(T * D * 64);
  for (auto& v : theta) v = U(rng);
  for (auto& v : feat) v = 5 * U(rng);
  float *dth, *dpk, *dfeat, *dy, *dg;
  hipMalloc(&dth, pos * 4); hipMalloc(&dpk, cvf_ef_pack_floats(&m) * 4); hipMalloc(&dfeat, feat.size() * 4);
  hipMalloc(&dy, T * k * 64 * 4); hipMalloc(&dg, T * k * D * 64 * 4);
  hipMemcpy(dth, theta.data(), pos * 4, hipMemcpyHostToDevice);
  hipMemcpy(dfeat, feat.data(), feat.size() * 4, hipMemcpyHostToDevice);
  float* dsaved = nullptr;
  if (cvf_ef_saved_floats(&m, T) > 0) hipMalloc(&dsaved, cvf_ef_saved_floats(&m, T) * 4);
  cvf_ef_pack(&m, dth, dpk, nullptr);
  for (int it = 0; it < 5; ++it) cvf_ef_mlp_fwd(&m, dth, dpk, dfeat, T, dy, dg, dsaved, nullptr);
  hipDeviceSynchronize();
  std::vector<unsigned long long> st(64 * 4096);
  hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8);
  const char* names[8] = {"start", "loads issued", "chunk0 done", "layer0 done", "chain+y done", "dchain done", "g done", "-"};
  double acc[8] = {0};
  int n = 0;
  for (int b = 0; b < 2 * T && b < 4096; ++b) {
    if (b & 1) continue;
    const unsigned long long* s = &st[b * 64];
    if (s[7] == 0) continue;
    for (int i = 1; i < 8; ++i) acc[i] += double(s[i] - s[i - 1]);
    ++n;
  }
  double tot = 0;
  for (int i = 1; i < 8; ++i) { printf("%-18s %9.0f cycles\n", names[i], acc[i] / n); tot += acc[i] / n; }
  printf("total %9.0f cycles over %d waves\n", tot, n);
  // ---- backward kernel phases
  {
    std::vector<float> hw(B), hy(T * k * 64), hq(T * k * D * 64);
    for (auto& v : hw) v = 1.0f + U(rng);
    for (auto& v : hy) v = U(rng);
    for (auto& v : hq) v = 0.01f * U(rng);
    std::vector<double> hc(4 * k + k * k);
    for (auto& v : hc) v = 0.01 * U(rng);
    float *dw, *dq, *dslab; double* dcoef;
    hipMalloc(&dw, B * 4); hipMalloc(&dq, hq.size() * 4); hipMalloc(&dcoef, hc.size() * 8);
    hipMalloc(&dslab, cvf_ef_backward_slab_rows(T) * (size_t)pos * 4);
    hipMemcpy(dw, hw.data(), B * 4, hipMemcpyHostToDevice);
    hipMemcpy(dy, hy.data(), hy.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dcoef, hc.data(), hc.size() * 8, hipMemcpyHostToDevice);
    cvf_ef_cfg cfg = {};
    cfg.k = k; cfg.lag_idx = 0;
    (void)0;
    for (int it = 0; it < 5; ++it) {
      int rc = cvf_ef_backward(&cfg, &m, dth, dpk, B, dw, nullptr, dfeat, dy, dq, dcoef, dslab, nullptr, dsaved, nullptr);
      if (rc) { printf("bwd failed: %s\n", cvf_last_error()); return 1; }
    }
    hipDeviceSynchronize();
    {
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0, nullptr);
      for (int it = 0; it < 10; ++it) cvf_ef_backward(&cfg, &m, dth, dpk, B, dw, nullptr, dfeat, dy, dq, dcoef, dslab, nullptr, dsaved, nullptr);
      hipEventRecord(e1, nullptr);
      hipDeviceSynchronize();
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      printf("cvf_ef_backward D=%d k=%d B=%d: %.1f us/launch\n", D, k, B, 100.0f * ms);
    }
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8);
    const char* bn[11] = {"alpha", "fwd chain", "d+tangent", "last layer", "hbar init", "reverse l=2", "reverse l=1", "reverse l=0", "-", "flush", "end"};
    double ab[11] = {0};
    int nb = 0;
    for (int b = 0; b < 2 * T && b < 4096; ++b) {
      const unsigned long long* s = &st[b * 64];
      if (s[18] == 0 || s[8] == 0) continue;
      ab[0] += double(s[9] - s[8]); ab[1] += double(s[10] - s[9]); ab[2] += double(s[11] - s[10]); ab[3] += double(s[12] - s[11]);
      ab[4] += double(s[13] - s[12]); ab[5] += double(s[14] - s[13]); ab[6] += double(s[15] - s[14]); ab[7] += double(s[17] - s[15]);
      ab[9] += double(s[18] - s[17]);
      ++nb;
    }
    double tb = 0;
    for (int i = 0; i < 11; ++i) { if (ab[i] > 0) printf("bwd %-14s %9.0f cycles\n", bn[i], ab[i] / nb); tb += ab[i] / nb; }
    printf("bwd total %9.0f cycles over %d waves\n", tb, nb);
  }
  return 0;
}
