// Developer tool: phase timing of ae_mfma_kernel (config-2 shape) with s_memtime stamps.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCVF_STAMPS -Iinclude -Icolvars-finder_amd/csrc -Wno-pass-failed \
//       tools/ae_probe.hip colvars-finder_amd/csrc/stats.hip colvars-finder_amd/csrc/ef_mfma.hip -o /tmp/ae_probe
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
  const int ids[] = {18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 40};
  const char* nm[] = {"", "setup (zero LDS, weights)", "frame index/weight loads", "forward (7 layers)", "error + zbar_L", "barrier l=6", "outer l=6", "bwd-data l=6 + barrier", "outer l=5", "bwd-data l=5 + barrier", "outer l=4", "bwd-data l=4 + barrier", "outer l=3", "bwd-data l=3 + barrier", "outer l=2", "bwd-data l=2 + barrier", "outer l=1", "bwd-data l=1 + barrier", "outer l=0", "(end)"};
  const int NS = sizeof(ids) / sizeof(ids[0]);
  std::vector<double> acc(NS, 0.0); int cnt = 0;
  for (int b = 0; b < 313; ++b) {
    const unsigned long long* s = &st[(b * 2) % 4096 * 64];
    bool ok = s[18] != 0;
    for (int i = 1; i < NS; ++i) ok = ok && s[ids[i]] >= s[ids[i - 1]] && s[ids[i]] - s[ids[i - 1]] < 10000000ull;
    if (!ok) continue;
    for (int i = 1; i < NS; ++i) acc[i] += double(s[ids[i]] - s[ids[i - 1]]);
    ++cnt;
  }
  double tot = 0;
  for (int i = 1; i < NS; ++i) { printf("   %-28s %8.0f cycles\n", nm[i], acc[i] / cnt); tot += acc[i] / cnt; }
  printf("   total %8.0f cycles over %d blocks (wave 0)\n", tot, cnt);
  {
    double a[8] = {0}; int c2 = 0;
    for (int b = 0; b < 313; ++b) {
      const unsigned long long* q = &st[(b * 2) % 4096 * 64];
      if (q[47] == 0 || q[47] < q[20] || q[47] - q[20] > 10000000ull) continue;
      a[0] += double(q[41] - q[20]);
      for (int l = 1; l < 7; ++l) a[l] += double(q[41 + l] - q[40 + l]);
      ++c2;
    }
    for (int l = 0; l < 7; ++l) printf("   forward layer %d (%d -> %d): %8.0f cycles\n", l, dims[l], dims[l + 1], a[l] / c2);
  }
  return 0;
}
