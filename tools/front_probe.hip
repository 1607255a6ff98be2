// Developer tool: phase timing (s_memtime stamps) of the fused front launch (K1 + forward + derivative) at config 3.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCVF_STAMPS -Iinclude -Icolvars-finder_amd/csrc -Wno-pass-failed \
//       tools/front_probe.hip colvars-finder_amd/csrc/stats.hip colvars-finder_amd/csrc/k1_align.hip \
//       colvars-finder_amd/csrc/k1_large.hip colvars-finder_amd/csrc/metric_large.hip -o /tmp/front_probe
#include "../colvars-finder_amd/csrc/ef_mfma.hip"
#include <cstdio>
#include <random>
#include <vector>

int main(int argc, char** argv) {
  const int N = 22, nc = 3 * N, k = 3, D = nc, Hh = 20, NHl = 3;
  const int64_t B = argc > 1 ? atoll(argv[1]) : 20000;
  const int64_t T = (B + 63) / 64;
  std::mt19937 rng(3);
  std::normal_distribution<float> G(0.0f, 1.0f);
  std::uniform_real_distribution<float> U(-0.2f, 0.2f);
  std::vector<float> ref(nc), refc(nc);
  for (auto& v : ref) v = 2.0f * G(rng);
  float cm[3] = {0, 0, 0};
  for (int a = 0; a < N; ++a) for (int d = 0; d < 3; ++d) cm[d] += ref[3 * a + d] / N;
  for (int a = 0; a < N; ++a) for (int d = 0; d < 3; ++d) refc[3 * a + d] = ref[3 * a + d] - cm[d];
  std::vector<int32_t> align(N), rec(6 * N);
  for (int a = 0; a < N; ++a) { align[a] = a; int32_t r[6] = {CVF_FEAT_POSITION, a, 0, 0, 0, 3 * a}; for (int i = 0; i < 6; ++i) rec[6 * a + i] = r[i]; }
  auto up = [](const void* h, size_t n) { void* d; (void)hipMalloc(&d, n); (void)hipMemcpy(d, h, n, hipMemcpyHostToDevice); return d; };
  cvf_pp_desc pp = {};
  pp.mode = CVF_PP_ALIGN; pp.n_coord = nc; pp.n_align = N; pp.n_rec = N; pp.d_r = nc; pp.has_position = 1;
  pp.flags = CVF_PP_ALIGN_CONTIG | CVF_PP_PURE_POSITION;
  pp.align_idx = (const int32_t*)up(align.data(), N * 4); pp.ref_c = (const float*)up(refc.data(), nc * 4); pp.rec = (const int32_t*)up(rec.data(), 6 * N * 4);
  cvf_mlp_desc m = {};
  m.n_nets = k; m.n_layers = NHl + 1;
  int dims[5] = {D, Hh, Hh, Hh, 1}, pos = 0;
  for (int i = 0; i < 5; ++i) m.dims[i] = dims[i];
  for (int n = 0; n < k; ++n)
    for (int l = 0; l < 4; ++l) { m.act[l] = l < 3; m.w_off[n][l] = pos; pos += dims[l] * dims[l + 1]; m.b_off[n][l] = pos; pos += dims[l + 1]; }
  m.n_params = pos;
  std::vector<float> theta(pos), x((size_t)B * nc), w(B, 1.0f), av(nc, 1.0f);
  for (auto& v : theta) v = U(rng);
  for (size_t i = 0; i < x.size(); ++i) x[i] = ref[i % nc] + 0.3f * G(rng);
  float* dth = (float*)up(theta.data(), pos * 4); float* dx = (float*)up(x.data(), x.size() * 4);
  float* dw = (float*)up(w.data(), B * 4); float* da = (float*)up(av.data(), nc * 4);
  float *dpk, *dfeat, *daux, *dy, *dsaved, *dq, *de; double *dscr, *dstats, *dlv, *dcf;
  (void)hipMalloc(&dpk, cvf_ef_pack_floats(&m) * 4); (void)hipMalloc(&dfeat, T * nc * 64 * 4); (void)hipMalloc(&daux, T * 18 * 64 * 4);
  (void)hipMalloc(&dy, T * k * 64 * 4); (void)hipMalloc(&dsaved, cvf_ef_saved_floats(&m, T) * 4); (void)hipMalloc(&dq, T * k * nc * 64 * 4);
  (void)hipMalloc(&de, T * k * 64 * 4); (void)hipMalloc(&dscr, (4 * T * 64 + 8192) * 8); (void)hipMalloc(&dstats, 64 * 8); (void)hipMalloc(&dlv, 64 * 8); (void)hipMalloc(&dcf, 128 * 8);
  cvf_ef_pack(&m, dth, dpk, nullptr);
  cvf_ef_cfg cfg = {};
  cfg.k = k; cfg.sort_eigvals = 1; cfg.alpha = 10; cfg.beta = 1; cfg.dt = 1;
  for (int i = 0; i < k; ++i) cfg.eig_w[i] = 1.0 - 0.2 * i;
  if (!cvf_ef_align_fwd_metric_supported(&m, &pp)) { printf("not supported\n"); return 1; }
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it)
    if (cvf_ef_align_fwd_metric_stats(&m, dth, dpk, dfeat, &pp, dx, B, daux, da, dy, dsaved, dq, de, &cfg, dw, dscr, nullptr, nullptr, nullptr, nullptr)) { printf("failed: %s\n", cvf_last_error()); return 1; }
  (void)hipEventRecord(e0, nullptr);
  for (int it = 0; it < 20; ++it) cvf_ef_align_fwd_metric_stats(&m, dth, dpk, dfeat, &pp, dx, B, daux, da, dy, dsaved, dq, de, &cfg, dw, dscr, nullptr, nullptr, nullptr, nullptr);
  (void)hipEventRecord(e1, nullptr); (void)hipDeviceSynchronize();
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("front launch B=%lld: %.1f us (back-to-back)\n", (long long)B, 1e3 * ms / 20);
  std::vector<unsigned long long> st(64 * 4096);
  (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8);
  const int ids[] = {50, 51, 52, 53, 60, 61, 54, 55, 56, 57, 58, 59, 10, 11, 12, 13, 14, 15};
  const char* nm[] = {"", "stage tile + tables", "covariance (split) + exchange", "solve", "aux rows stored", "features", "barrier", "forward operand loads issued", "layer 0", "hidden layers, y, hand-off", "d chain", "barrier + g -> LDS", "barrier, y exchange", "pass 1", "Z", "pass 2", "sums + dR", "pass 3"};
  const int NS = sizeof(ids) / sizeof(ids[0]);
  std::vector<double> acc(NS, 0.0); int cnt = 0;
  for (int b = 0; b < T && b < 2000; ++b) {
    const unsigned long long* s = &st[(b * 2) % 4096 * 64];
    bool ok = s[50] != 0;
    for (int i = 1; i < NS; ++i) ok = ok && s[ids[i]] >= s[ids[i - 1]] && s[ids[i]] - s[ids[i - 1]] < 10000000ull;
    if (!ok) continue;
    for (int i = 1; i < NS; ++i) acc[i] += double(s[ids[i]] - s[ids[i - 1]]);
    ++cnt;
  }
  double tot = 0;
  for (int i = 1; i < NS; ++i) { printf("   %-32s %8.0f cycles\n", nm[i], acc[i] / cnt); tot += acc[i] / cnt; }
  printf("   total %8.0f cycles over %d blocks (wave 0)\n", tot, cnt);
  return 0;
}
