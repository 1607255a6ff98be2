// Developer tool (not part of the library): the 16-frames-per-wave generator step (ef16_front.hip + ef16_back.hip: the front kernel and the four-wave
// backward kernel of ef_mfma.hip) at the config-3 shape - kernel times with HIP events at several batch sizes and, with
// -DCVF_STAMPS, s_memtime phase stamps per wave.
// Build + run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCVF_STAMPS -DCVF_STAMP_WPB=4 -Iinclude -Icolvars-finder_amd/csrc -Wno-pass-failed \
//       tools/ef16_probe.hip -Lcolvars-finder_amd/colvarsfinder -lcvf_hip -Wl,-rpath,\$ORIGIN/../../colvars-finder_amd/colvarsfinder -o tools/build/ef16_probe
#include "../colvars-finder_amd/csrc/ef16_front.hip"
#include "../colvars-finder_amd/csrc/ef16_back.hip"
// (everything else - cvf_ef_pack, the batch sums - comes from the library the probe is linked against)
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>


// launch-level view from (start, end) stamps of many blocks: s_memtime has a different base on every XCD, so the blocks are grouped
// by base (values within 1e8 cycles of each other) and the figures are per group: span = last end - first start, the blocks'
// start times after the group's first, their own durations
static void launch_view(const char* what, std::vector<std::pair<unsigned long long, unsigned long long>> se) {
  std::sort(se.begin(), se.end());
  size_t i = 0;
  int g = 0;
  double span_sum = 0, start_avg = 0, start_max = 0, dur = 0;
  size_t n = 0;
  while (i < se.size()) {
    size_t j = i;
    unsigned long long last_end = 0;
    while (j < se.size() && se[j].first - se[i].first < 100000000ull) { last_end = std::max(last_end, se[j].second); ++j; }
    span_sum += double(last_end - se[i].first);
    for (size_t t = i; t < j; ++t) { start_avg += double(se[t].first - se[i].first); start_max = std::max(start_max, double(se[t].first - se[i].first)); dur += double(se[t].second - se[t].first); ++n; }
    ++g;
    i = j;
  }
  printf("== %s: %zu blocks, %d clock domain(s) | span first start -> last end %.2f us | block start after the first: avg %.2f max %.2f us | block duration avg %.2f us (100 MHz chip-wide counter)\n",
         what, n, g, span_sum / g / 100.0, start_avg / n / 100.0, start_max / 100.0, dur / n / 100.0);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  const int k = 3, NA = 22, D = 66, H = 20, NHl = 3;
  const int Bmax = 160000;
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
  std::uniform_real_distribution<float> U(-1.0f, 1.0f);
  std::normal_distribution<float> G(0.0f, 1.0f);
  std::vector<float> theta(pos), x((size_t)Bmax * NA * 3), ref(NA * 3), a(NA * 3), w(Bmax);
  for (auto& v : theta) v = 0.2f * U(rng);
  for (auto& v : ref) v = 2.0f * G(rng);
  {  // centre the reference
    float c[3] = {0, 0, 0};
    for (int i = 0; i < NA; ++i) for (int d = 0; d < 3; ++d) c[d] += ref[3 * i + d] / NA;
    for (int i = 0; i < NA; ++i) for (int d = 0; d < 3; ++d) ref[3 * i + d] -= c[d];
  }
  for (auto& v : a) v = 0.1f + 0.5f * std::fabs(U(rng));
  for (auto& v : w) v = 1.0f + 0.5f * U(rng);
  for (size_t b = 0; b < (size_t)Bmax; ++b)
    for (int i = 0; i < NA * 3; ++i) x[b * NA * 3 + i] = ref[i] + 0.3f * G(rng) + (i % 3 == 0 ? 0.5f : -0.25f);
  std::vector<int> aidx(NA), rec(NA * 6);
  for (int i = 0; i < NA; ++i) {
    aidx[i] = i;
    rec[6 * i] = CVF_FEAT_POSITION; rec[6 * i + 1] = i; rec[6 * i + 2] = rec[6 * i + 3] = rec[6 * i + 4] = 0; rec[6 * i + 5] = 3 * i;
  }
  float *dth, *dpk, *dx, *dref, *da, *dw, *dfeat, *dy, *dsaved, *dq, *de, *dslab;
  int *daidx, *drec;
  double *dscr, *dstats, *dlv, *dcoef;
  const int64_t Tm = cvf_ntiles(Bmax);
  CK(hipMalloc(&dth, pos * 4)); CK(hipMalloc(&dpk, cvf_ef_pack_floats(&m) * 4)); CK(hipMalloc(&dx, x.size() * 4));
  CK(hipMalloc(&dref, ref.size() * 4)); CK(hipMalloc(&da, a.size() * 4)); CK(hipMalloc(&dw, w.size() * 4));
  CK(hipMalloc(&daidx, NA * 4)); CK(hipMalloc(&drec, NA * 6 * 4));
  CK(hipMalloc(&dfeat, Tm * D * 64 * 4)); CK(hipMalloc(&dy, Tm * k * 64 * 4)); CK(hipMalloc(&dsaved, cvf_ef16_saved_floats(&m, Tm) * 4));
  CK(hipMalloc(&dq, Tm * k * D * 64 * 4)); CK(hipMalloc(&de, Tm * k * 64 * 4));
  CK(hipMalloc(&dslab, cvf_ef_backward_slab_rows(Tm) * (size_t)pos * 4));
  CK(hipMalloc(&dscr, cvf_ef16_scratch_doubles(Bmax, k) * 8)); CK(hipMemset(dscr, 0, cvf_ef16_scratch_doubles(Bmax, k) * 8)); CK(hipMalloc(&dstats, 64 * 8)); CK(hipMalloc(&dlv, 64 * 8)); CK(hipMalloc(&dcoef, 128 * 8));
  CK(hipMemcpy(dth, theta.data(), pos * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dref, ref.data(), ref.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(daidx, aidx.data(), NA * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(drec, rec.data(), NA * 6 * 4, hipMemcpyHostToDevice));
  cvf_ef_pack(&m, dth, dpk, nullptr);
  cvf_pp_desc pp = {};
  pp.mode = CVF_PP_ALIGN; pp.n_coord = 3 * NA; pp.n_align = NA; pp.n_rec = NA; pp.d_r = D; pp.has_position = 1;
  pp.flags = CVF_PP_ALIGN_CONTIG | CVF_PP_PURE_POSITION;
  pp.align_idx = daidx; pp.ref_c = dref; pp.rec = drec;
  cvf_ef_cfg cfg = {};
  cfg.k = k; cfg.lag_idx = 0; cfg.sort_eigvals = 1; cfg.alpha = 20.0; cfg.beta = 1.0; cfg.dt = 1.0;
  cfg.eig_w[0] = 1.0; cfg.eig_w[1] = 0.75; cfg.eig_w[2] = 0.5;
  if (!cvf_ef16_supported(&m, &pp)) { printf("shape not supported\n"); return 1; }
  hipEvent_t e0, e1, e2;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  const int batches[] = {2500, 5000, 20000, 40000, 160000};
  for (int B : batches) {
    const int reps = 30;
    float tf = 0, tb = 0;
    for (int it = 0; it < reps + 5; ++it) {
      CK(hipEventRecord(e0));
      int rc = cvf_ef16_front(&m, dth, dpk, dfeat, &pp, dx, B, da, dy, dsaved, dq, de, &cfg, dw, dscr, dstats, dlv, dcoef, nullptr);
      if (rc) { printf("front failed: %s\n", cvf_last_error()); return 1; }
      CK(hipEventRecord(e1));
      rc = cvf_ef16_backward(&cfg, &m, dth, dpk, B, dw, dfeat, dy, dq, dcoef, dslab, nullptr, dsaved, nullptr);
      if (rc) { printf("back failed: %s\n", cvf_last_error()); return 1; }
      CK(hipEventRecord(e2));
      CK(hipEventSynchronize(e2));
      float a_, b_;
      CK(hipEventElapsedTime(&a_, e0, e1)); CK(hipEventElapsedTime(&b_, e1, e2));
      if (it >= 5) { tf += a_; tb += b_; }
    }
    double lv[4];
    CK(hipMemcpy(lv, dlv, 32, hipMemcpyDeviceToHost));
    printf("B=%6d  front(+finish) %7.1f us   back %7.1f us   loss %.6f\n", B, tf / reps * 1e3, tb / reps * 1e3, lv[0]);
  }
#ifdef CVF_STAMPS
  for (int B : {20000, 2500}) {
    std::vector<unsigned long long> st(64 * 4096, 0);
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), st.data(), st.size() * 8));
    cvf_ef16_front(&m, dth, dpk, dfeat, &pp, dx, B, da, dy, dsaved, dq, de, &cfg, dw, dscr, dstats, dlv, dcoef, nullptr);
    CK(hipDeviceSynchronize());
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8));
    {
      std::vector<std::pair<unsigned long long, unsigned long long>> se;
      for (int u = 0; u < 4 * (int)cvf_ntiles(B) && u * CVF_STAMP_WPB < 4096; ++u) {
        const unsigned long long* s_ = &st[(size_t)(u * CVF_STAMP_WPB) * 64];
        if (s_[40] && s_[41]) se.push_back({s_[40], s_[41]});
      }
      char nm[64]; snprintf(nm, sizeof nm, "front B=%d", B);
      launch_view(nm, se);
    }
    const char* fn[10] = {"stage", "kabsch(w0)+barrier", "weights req + features + barrier", "layer 0", "hidden + y + hand-off", "d chain + g", "pass 1", "pass 2", "pass 3", "stats row"};
    const int units = 4 * (int)cvf_ntiles(B);
    for (int wv = 0; wv < (CVF_STAMP_WPB < 2 ? CVF_STAMP_WPB : 2); ++wv) {
      double acc[10] = {0}, tot = 0;
      int n = 0;
      for (int u = 0; u < units && u * CVF_STAMP_WPB + wv < 4096; ++u) {
        const unsigned long long* s = &st[(size_t)(u * CVF_STAMP_WPB + wv) * 64];
        if (s[20] == 0 || s[29] == 0) continue;
        for (int i = 0; i < 9; ++i) acc[i] += double(s[21 + i] - s[20 + i]);
        if (s[30]) acc[9] += double(s[30] - s[29]);
        ++n;
      }
      printf("-- front B=%d wave %d (%d units sampled)\n", B, wv, n);
      for (int i = 0; i < 10; ++i) { printf("   %-34s %8.0f cycles\n", fn[i], acc[i] / n); tot += acc[i] / n; }
      printf("   total %8.0f cycles\n", tot);
      {  // pass 3 in detail
        double d3[5] = {0};
        int n3 = 0;
        for (int u = 0; u < units && u * CVF_STAMP_WPB + wv < 4096; ++u) {
          const unsigned long long* s = &st[(size_t)(u * CVF_STAMP_WPB + wv) * 64];
          if (s[28] == 0 || s[34] == 0) continue;
          d3[0] += double(s[31] - s[28]); d3[1] += double(s[32] - s[31]); d3[2] += double(s[33] - s[32]); d3[3] += double(s[34] - s[33]);
          d3[4] += double(s[29] - s[34]);
          ++n3;
        }
        printf("   pass 3 split: sums+dR %.0f | q loop %.0f | W1 q (loads, MFMA, stores) %.0f | q stores %.0f | feature copy %.0f\n",
               d3[0] / n3, d3[1] / n3, d3[2] / n3, d3[3] / n3, d3[4] / n3);
      }
    }
    std::fill(st.begin(), st.end(), 0ull);
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), st.data(), st.size() * 8));
    cvf_ef16_backward(&cfg, &m, dth, dpk, B, dw, dfeat, dy, dq, dcoef, dslab, nullptr, dsaved, nullptr);
    CK(hipDeviceSynchronize());
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8));
    {
      std::vector<std::pair<unsigned long long, unsigned long long>> se;
      double pro = 0;
      for (int t = 0; t < (int)cvf_ntiles(B) && t * CVF_STAMP_WPB < 4096; ++t) {
        const unsigned long long* s_ = &st[(size_t)(t * CVF_STAMP_WPB) * 64];
        if (s_[40] && s_[41]) { se.push_back({s_[40], s_[41]}); pro += double(s_[8] - s_[7]); }
      }
      char nm[64]; snprintf(nm, sizeof nm, "back B=%d (net 0)", B);
      launch_view(nm, se);
      printf("   prologue (kernel entry -> tile loop) avg %.0f cycles\n", pro / se.size());
    }
    const char* bn[9] = {"requests + alpha", "-", "tangent chain", "last layer", "reverse l=2", "reverse l=1", "reverse l=0", "flush", "-"};
    const int tiles = (int)cvf_ntiles(B);
    for (int wv = 0; wv < CVF_STAMP_WPB; wv += 3) {
      double ab[9] = {0}, tb = 0;
      int nb = 0;
      for (int t = 0; t < tiles && t * CVF_STAMP_WPB + wv < 4096; ++t) {
        const unsigned long long* s = &st[(size_t)(t * CVF_STAMP_WPB + wv) * 64];
        if (s[18] == 0 || s[8] == 0) continue;
        ab[0] += double(s[9] - s[8]); ab[2] += double(s[11] - s[9]); ab[3] += double(s[12] - s[11]);
        ab[4] += double(s[14] - s[12]); ab[5] += double(s[15] - s[14]); ab[6] += double(s[17] - s[15]); ab[7] += double(s[18] - s[17]);
        ++nb;
      }
      printf("-- back B=%d wave %d (%d tiles sampled)\n", B, wv, nb);
      for (int i = 0; i < 8; ++i) { printf("   %-22s %8.0f cycles\n", bn[i], ab[i] / nb); tb += ab[i] / nb; }
      printf("   total %8.0f cycles\n", tb);
    }
  }
#endif
  return 0;
}
