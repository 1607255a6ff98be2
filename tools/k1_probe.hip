// Developer tool (not part of the library): phase timing of k1_align_kernel<true, true, false> with s_memtime stamps and
// event-timed throughput at config-3 shape.  Build+run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCVF_STAMPS -Iinclude -Icolvars-finder_amd/csrc -Wno-pass-failed \
//       tools/k1_probe.hip colvars-finder_amd/csrc/stats.hip colvars-finder_amd/csrc/k1_large.hip -o /tmp/k1_probe
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
  const int64_t sizes[3] = {20000, 100000, 1000000};
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
    (void)hipFree(dx); (void)hipFree(dfeat); (void)hipFree(daux);
  }
  return 0;
}
