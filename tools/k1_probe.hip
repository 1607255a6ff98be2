// Developer tool (not part of the library): phase timing of k1_align_kernel<true, true, false> with s_memtime stamps and
// event-timed throughput at config-3 shape.  Build+run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCVF_STAMPS -Iinclude -Icolvars-finder_amd/csrc -Wno-pass-failed \
//       tools/k1_probe.hip colvars-finder_amd/csrc/stats.hip colvars-finder_amd/csrc/k1_large.hip colvars-finder_amd/csrc/metric_large.hip -o /tmp/k1_probe
#include "../colvars-finder_amd/csrc/k1_align.hip"
#include <cstdio>
#include <random>
#include <vector>

int main(int argc, char** argv) {
  const int N = 22, nc = 3 * N;
  std::mt19937 rng(3);
  std::normal_distribution<float> G(0.0f, 1.0f);
  std::vector<float> ref(nc);
  for (auto& v : ref) v = 2.0f * G(rng);
  float cm[3] = {0, 0, 0};
  for (int a = 0; a < N; ++a) for (int d = 0; d < 3; ++d) cm[d] += ref[3 * a + d] / N;
  std::vector<float> refc(nc);
  for (int a = 0; a < N; ++a) for (int d = 0; d < 3; ++d) refc[3 * a + d] = ref[3 * a + d] - cm[d];
  std::vector<int32_t> align(N), rec(6 * N);
  for (int a = 0; a < N; ++a) { align[a] = a; int32_t r[6] = {CVF_FEAT_POSITION, a, 0, 0, 0, 3 * a}; for (int i = 0; i < 6; ++i) rec[6 * a + i] = r[i]; }
  int32_t *dal, *drec; float* dref;
  (void)hipMalloc(&dal, N * 4); (void)hipMalloc(&drec, 6 * N * 4); (void)hipMalloc(&dref, nc * 4);
  (void)hipMemcpy(dal, align.data(), N * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(drec, rec.data(), 6 * N * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dref, refc.data(), nc * 4, hipMemcpyHostToDevice);
  cvf_pp_desc pp = {};
  pp.mode = CVF_PP_ALIGN; pp.n_coord = nc; pp.n_align = N; pp.n_rec = N; pp.d_r = nc; pp.has_position = 1;
  pp.flags = CVF_PP_ALIGN_CONTIG | CVF_PP_PURE_POSITION;
  pp.align_idx = dal; pp.ref_c = dref; pp.rec = drec;
  const int64_t sizes[6] = {20000, 100000, 150000, 250000, 500000, 1000000};
  for (int64_t B : sizes) {
    const int64_t T = (B + 63) / 64;
    std::vector<float> x((size_t)B * nc);
    for (size_t i = 0; i < x.size(); ++i) x[i] = ref[i % nc] + 0.3f * G(rng);
    float *dx, *dfeat, *daux;
    (void)hipMalloc(&dx, x.size() * 4); (void)hipMalloc(&dfeat, T * nc * 64 * 4); (void)hipMalloc(&daux, T * 18 * 64 * 4);
    (void)hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) cvf_align_feature_fwd(&pp, dx, B, dfeat, nullptr, daux, nullptr, nullptr);
    (void)hipEventRecord(e0, nullptr);
    const int reps = 20;
    for (int it = 0; it < reps; ++it) {
      int rc = cvf_align_feature_fwd(&pp, dx, B, dfeat, nullptr, daux, nullptr, nullptr);
      if (rc) { printf("failed: %s\n", cvf_last_error()); return 1; }
    }
    (void)hipEventRecord(e1, nullptr);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = 1e3 * ms / reps;
    printf("B=%lld: %.1f us/launch (back-to-back), %.0f GB/s algorithmic (532 B/frame)\n", (long long)B, us, 532.0 * B / us * 1e-3);
    {   // features only (no rotation / centroid / K^-1 rows)
      (void)hipEventRecord(e0, nullptr);
      for (int it = 0; it < reps; ++it) cvf_align_feature_fwd(&pp, dx, B, dfeat, nullptr, nullptr, nullptr, nullptr);
      (void)hipEventRecord(e1, nullptr);
      (void)hipDeviceSynchronize();
      (void)hipEventElapsedTime(&ms, e0, e1);
      printf("   features only: %.1f us/launch, %.0f GB/s algorithmic\n", 1e3 * ms / reps, 532.0 * B / (1e3 * ms / reps) * 1e-3);
      (void)hipEventRecord(e0, nullptr);
      for (int it = 0; it < reps; ++it) cvf_align_feature_fwd(&pp, dx, B, nullptr, dfeat, nullptr, nullptr, nullptr);
      (void)hipEventRecord(e1, nullptr);
      (void)hipDeviceSynchronize();
      (void)hipEventElapsedTime(&ms, e0, e1);
      printf("   row-major features only: %.1f us/launch, %.0f GB/s algorithmic\n", 1e3 * ms / reps, 532.0 * B / (1e3 * ms / reps) * 1e-3);
      cvf_align_feature_fwd(&pp, dx, B, dfeat, nullptr, daux, nullptr, nullptr);
      (void)hipDeviceSynchronize();
    }
    std::vector<unsigned long long> st(64 * 4096);
    (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8);
    const char* nm[7] = {"", "tile staged", "tables+barrier", "centroid+covariance", "rotation solve", "(align end)", "aux+features"};
    double acc[7] = {0}; int n = 0;
    for (int b = 0; b < 4096 && b < T; ++b) {
      const unsigned long long* s = &st[(b * 2) % 4096 * 64];
      if (s[6] == 0) continue;
      for (int i = 1; i < 7; ++i) acc[i] += double(s[i] - s[i - 1]);
      ++n;
    }
    double tot = 0;
    for (int i = 1; i < 7; ++i) { printf("   %-22s %8.0f cycles\n", nm[i], acc[i] / n); tot += acc[i] / n; }
    printf("   total %8.0f cycles over %d waves\n", tot, n);
    if (B == 20000) {   // the derivative kernel on the same tile, three nets
      const int k = 3;
      std::vector<float> g((size_t)T * k * nc * 64), av(nc, 1.0f);
      for (auto& v : g) v = G(rng);
      float *dg, *dq, *de, *da;
      (void)hipMalloc(&dg, g.size() * 4); (void)hipMalloc(&dq, g.size() * 4); (void)hipMalloc(&de, T * k * 64 * 4); (void)hipMalloc(&da, nc * 4);
      (void)hipMemcpy(dg, g.data(), g.size() * 4, hipMemcpyHostToDevice);
      (void)hipMemcpy(da, av.data(), nc * 4, hipMemcpyHostToDevice);
      for (int it = 0; it < 3; ++it) cvf_metric_apply(&pp, dx, B, daux, da, k, dg, dq, de, nullptr, nullptr, nullptr);
      (void)hipEventRecord(e0, nullptr);
      for (int it = 0; it < reps; ++it) {
        int rc = cvf_metric_apply(&pp, dx, B, daux, da, k, dg, dq, de, nullptr, nullptr, nullptr);
        if (rc) { printf("failed: %s\n", cvf_last_error()); return 1; }
      }
      (void)hipEventRecord(e1, nullptr);
      (void)hipDeviceSynchronize();
      (void)hipEventElapsedTime(&ms, e0, e1);
      printf("metric_apply B=%lld k=%d: %.1f us/launch\n", (long long)B, k, 1e3 * ms / reps);
      (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8);
      const char* mn[8] = {"", "tile staged", "tables+aux", "pass 1 (M, sum p)", "Z", "pass 2 (G,E,u)", "dR", "pass 3 (q)"};
      double am[8] = {0}; int nm2 = 0;
      for (int b = 0; b < 4096 && b < T; ++b) {
        const unsigned long long* s2 = &st[(b * 2) % 4096 * 64];
        if (s2[15] == 0) continue;
        for (int i = 1; i < 8; ++i) am[i] += double(s2[8 + i] - s2[8 + i - 1]);
        ++nm2;
      }
      double tm = 0;
      for (int i = 1; i < 8; ++i) { printf("   %-22s %8.0f cycles\n", mn[i], am[i] / nm2); tm += am[i] / nm2; }
      printf("   total %8.0f cycles over %d waves\n", tm, nm2);
      // fused epilogue (batch sums + loss tail)
      cvf_ef_cfg cfg = {};
      cfg.k = k; cfg.lag_idx = 0; cfg.sort_eigvals = 1; cfg.alpha = 10.0; cfg.beta = 1.0; cfg.dt = 1.0;
      for (int i = 0; i < k; ++i) cfg.eig_w[i] = 1.0 - 0.3 * i;
      std::vector<float> hw(B, 1.0f), hy((size_t)T * k * 64);
      for (auto& v : hy) v = G(rng);
      float *dw, *dy; double *dscr, *dstats, *dlv, *dcf;
      const int64_t nscr = cvf_metric_stats_scratch_doubles(B, k);
      (void)hipMalloc(&dw, B * 4); (void)hipMalloc(&dy, hy.size() * 4); (void)hipMalloc(&dscr, nscr * 8);
      (void)hipMalloc(&dstats, 64 * 8); (void)hipMalloc(&dlv, 64 * 8); (void)hipMalloc(&dcf, 128 * 8);
      (void)hipMemcpy(dw, hw.data(), B * 4, hipMemcpyHostToDevice);
      (void)hipMemcpy(dy, hy.data(), hy.size() * 4, hipMemcpyHostToDevice);
      (void)hipMemset(dscr, 0, nscr * 8);
      for (int it = 0; it < 3; ++it) cvf_metric_apply_stats(&pp, dx, B, daux, da, k, dg, dq, de, nullptr, nullptr, &cfg, dw, dy, dscr, dstats, dlv, dcf, nullptr);
      (void)hipEventRecord(e0, nullptr);
      for (int it = 0; it < reps; ++it) {
        int rc = cvf_metric_apply_stats(&pp, dx, B, daux, da, k, dg, dq, de, nullptr, nullptr, &cfg, dw, dy, dscr, dstats, dlv, dcf, nullptr);
        if (rc) { printf("failed: %s\n", cvf_last_error()); return 1; }
      }
      (void)hipEventRecord(e1, nullptr);
      (void)hipDeviceSynchronize();
      (void)hipEventElapsedTime(&ms, e0, e1);
      printf("metric_apply_stats B=%lld k=%d: %.1f us/launch\n", (long long)B, k, 1e3 * ms / reps);
    }
    (void)hipFree(dx); (void)hipFree(dfeat); (void)hipFree(daux);
  }
  return 0;
}
