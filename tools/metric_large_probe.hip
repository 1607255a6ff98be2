// Developer tool: phase timing (s_memtime stamps) of the large-molecule derivative kernel (metric_large) at the
// config-5 shape (5000 atoms, 32 positions + 96 dihedrals + 96 distances).  Build on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCVF_STAMPS -Iinclude -Icolvars-finder_amd/csrc -Wno-pass-failed \
//       tools/metric_large_probe.hip colvars-finder_amd/csrc/stats.hip colvars-finder_amd/csrc/k1_large.hip -o /tmp/metric_large_probe
#include "../colvars-finder_amd/csrc/metric_large.hip"
int cvf_k1_large_launch(const cvf_pp_desc* pp, const float* x, int64_t B, float* feat_tiled, float* feat_rows,
                        float* aux_tiled, float* slot_xyz, hipStream_t s);
size_t cvf_k1_large_scratch_bytes(const cvf_pp_desc* pp, int64_t B);
#include <cstdio>
#include <random>
#include <vector>
#include <algorithm>

int main(int argc, char** argv) {
  const int N = 5000, nc = 3 * N;
  const int64_t B = argc > 1 ? atoll(argv[1]) : 2000;
  std::mt19937 rng(5);
  std::normal_distribution<float> G(0.0f, 1.0f);
  std::vector<float> ref(nc);
  for (auto& v : ref) v = 20.0f * G(rng);
  double cm[3] = {0, 0, 0};
  for (int a = 0; a < N; ++a) for (int d = 0; d < 3; ++d) cm[d] += ref[3 * a + d] / N;
  std::vector<float> refc(nc);
  for (int a = 0; a < N; ++a) for (int d = 0; d < 3; ++d) refc[3 * a + d] = ref[3 * a + d] - (float)cm[d];
  // feature records over random atoms
  std::vector<std::vector<int>> recs;   // type, atoms..., out
  std::uniform_int_distribution<int> U(0, N - 1);
  int out = 0;
  for (int i = 0; i < 32; ++i) { recs.push_back({CVF_FEAT_POSITION, U(rng), 0, 0, 0, out}); out += 3; }
  for (int i = 0; i < 96; ++i) { recs.push_back({CVF_FEAT_DIHEDRAL, U(rng), U(rng), U(rng), U(rng), out}); out += 2; }
  for (int i = 0; i < 96; ++i) { recs.push_back({CVF_FEAT_BOND, U(rng), U(rng), 0, 0, out}); out += 1; }
  const int d_r = out;
  std::vector<int> used;
  auto natoms = [](int t) { return t == CVF_FEAT_POSITION ? 1 : t == CVF_FEAT_BOND ? 2 : t == CVF_FEAT_ANGLE ? 3 : 4; };
  for (auto& r : recs) for (int i = 0; i < natoms(r[0]); ++i) used.push_back(r[1 + i]);
  std::sort(used.begin(), used.end());
  used.erase(std::unique(used.begin(), used.end()), used.end());
  std::vector<int32_t> atom_slot(N, -1), atom_align(N), align(N), rec, rec_slot, slot_atom(used.begin(), used.end());
  for (size_t i = 0; i < used.size(); ++i) atom_slot[used[i]] = (int)i;
  for (int a = 0; a < N; ++a) { atom_align[a] = a; align[a] = a; }
  for (auto& r : recs) for (int v : r) rec.push_back(v);
  std::stable_sort(recs.begin(), recs.end(), [](const std::vector<int>& a, const std::vector<int>& b) { return a[0] < b[0]; });
  int n_rec_slot = 0;
  for (size_t ri = 0; ri < recs.size(); ++ri) {   // batches of 64 of one type, padded with type -1 (as pp.py does)
    const auto& r = recs[ri];
    rec_slot.push_back(r[0]);
    for (int i = 0; i < 4; ++i) rec_slot.push_back(i < natoms(r[0]) ? atom_slot[r[1 + i]] : 0);
    rec_slot.push_back(r[5]);
    ++n_rec_slot;
    if (ri + 1 == recs.size() || recs[ri + 1][0] != r[0])
      while (n_rec_slot % 64) { const int32_t padr[6] = {-1, 0, 0, 0, 0, 0}; rec_slot.insert(rec_slot.end(), padr, padr + 6); ++n_rec_slot; }
  }
  auto up = [](const void* h, size_t n) { void* d; (void)hipMalloc(&d, n); (void)hipMemcpy(d, h, n, hipMemcpyHostToDevice); return d; };
  cvf_pp_desc pp = {};
  pp.mode = CVF_PP_ALIGN; pp.n_coord = nc; pp.n_align = N; pp.n_rec = (int)recs.size(); pp.d_r = d_r; pp.has_position = 1;
  pp.flags = CVF_PP_ALIGN_CONTIG | CVF_PP_SLOT_BATCHED;
  pp.align_idx = (const int32_t*)up(align.data(), N * 4); pp.ref_c = (const float*)up(refc.data(), nc * 4);
  pp.rec = (const int32_t*)up(rec.data(), rec.size() * 4);
  pp.atom_align = (const int32_t*)up(atom_align.data(), N * 4); pp.atom_slot = (const int32_t*)up(atom_slot.data(), N * 4);
  pp.rec_slot = (const int32_t*)up(rec_slot.data(), rec_slot.size() * 4); pp.slot_atom = (const int32_t*)up(slot_atom.data(), slot_atom.size() * 4);
  pp.n_slot = (int)used.size();
  pp.n_rec_slot = n_rec_slot;
  {  // row tables of the derivative kernel (as pp.py builds them): rows sorted by (slot, record, position)
    struct Pair { int slot, rec, pos; };
    std::vector<Pair> pairs;
    for (size_t i = 0; i < recs.size(); ++i)
      for (int j = 0; j < natoms(recs[i][0]); ++j) pairs.push_back({atom_slot[recs[i][1 + j]], (int)i, j});
    std::sort(pairs.begin(), pairs.end(), [](const Pair& a, const Pair& b) {
      return a.slot != b.slot ? a.slot < b.slot : a.rec != b.rec ? a.rec < b.rec : a.pos < b.pos; });
    std::vector<int32_t> slot_row(used.size() + 1, 0), mrec(recs.size() * 8, 0), rowof(recs.size() * 4, 0);
    for (size_t n = 0; n < pairs.size(); ++n) { slot_row[pairs[n].slot + 1]++; rowof[pairs[n].rec * 4 + pairs[n].pos] = (int)n; }
    for (size_t t = 0; t < used.size(); ++t) slot_row[t + 1] += slot_row[t];
    for (size_t i = 0; i < recs.size(); ++i) {
      int32_t* m = &mrec[i * 8];
      const int na = natoms(recs[i][0]);
      int sl[4] = {0, 0, 0, 0}, rw[4] = {0, 0, 0, 0}, ur[4] = {0, 0, 0, 0};
      for (int j = 0; j < na; ++j) { sl[j] = atom_slot[recs[i][1 + j]]; rw[j] = rowof[i * 4 + j]; ur[j] = slot_row[sl[j]]; }
      m[0] = (recs[i][0] + 1) | (recs[i][5] << 3);
      m[1] = sl[0] | (sl[1] << 16); m[2] = sl[2] | (sl[3] << 16); m[3] = ur[0] | (ur[1] << 16); m[4] = ur[2] | (ur[3] << 16);
      m[5] = (rw[0] - ur[0]) | ((rw[1] - ur[1]) << 8) | ((rw[2] - ur[2]) << 16) | ((rw[3] - ur[3]) << 24);
    }
    pp.mrec = (const int32_t*)up(mrec.data(), mrec.size() * 4); pp.slot_row = (const int32_t*)up(slot_row.data(), slot_row.size() * 4);
    pp.n_mrec = (int)recs.size(); pp.n_ref = (int)pairs.size();
  }
  const int64_t T = (B + 63) / 64;
  float *dx, *dfeat, *daux, *dslot;
  const size_t xb = (size_t)B * nc * 4;
  (void)hipMalloc(&dx, xb); (void)hipMalloc(&dfeat, T * d_r * 64 * 4); (void)hipMalloc(&daux, T * 18 * 64 * 4);
  (void)hipMalloc(&dslot, cvf_k1_large_scratch_bytes(&pp, B));
  {
    std::vector<float> one(nc);
    for (int i = 0; i < nc; ++i) one[i] = ref[i] + 0.5f * G(rng);
    for (int64_t b = 0; b < B; ++b) (void)hipMemcpy(dx + b * nc, one.data(), nc * 4, hipMemcpyHostToDevice);
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int reps = 10;
  float ms = 0;
  std::vector<unsigned long long> st(64 * 4096);
  if (cvf_k1_large_launch(&pp, dx, B, dfeat, nullptr, daux, dslot, nullptr)) { printf("K1 failed: %s\n", cvf_last_error()); return 1; }
  {  // the derivative kernel of large molecules on the same frames, k = 6 nets
    const int k = 6;
    std::vector<float> g((size_t)T * k * d_r * 64), av(nc, 1.0f);
    for (auto& v : g) v = G(rng);
    float *dg, *dq, *de, *da; double* ddense;
    (void)hipMalloc(&dg, g.size() * 4); (void)hipMalloc(&dq, g.size() * 4); (void)hipMalloc(&de, T * k * 64 * 4); (void)hipMalloc(&da, nc * 4);
    (void)hipMalloc(&ddense, cvf_metric_dense_doubles(&pp) * 8);
    (void)hipMemcpy(dg, g.data(), g.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(da, av.data(), nc * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(metric_dense_kernel, dim3(1), dim3(256), 0, 0, pp, da, ddense);
    for (int it = 0; it < 2; ++it) cvf_metric_large_launch(&pp, B, daux, da, k, dslot, ddense, dg, dq, de, nullptr, nullptr, nullptr);
    (void)hipEventRecord(e0, nullptr);
    for (int it = 0; it < reps; ++it) {
      int rc = cvf_metric_large_launch(&pp, B, daux, da, k, dslot, ddense, dg, dq, de, nullptr, nullptr, nullptr);
      if (rc) { printf("failed: %s\n", cvf_last_error()); return 1; }
    }
    (void)hipEventRecord(e1, nullptr);
    (void)hipDeviceSynchronize();
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("metric_large B=%lld k=%d: %.1f us/launch\n", (long long)B, k, 1e3 * ms / reps);
    (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8);
    const char* mn[7] = {"", "prologue (slot constants, aux)", "phase A: geometry + rows", "block sum", "dense + phase B: slots", "block sum", "phase C: q"};
    for (int wv = 0; wv < CVF_STAMP_WPB; ++wv) {
      double am[7] = {0}; int c2 = 0;
      for (int b = 0; b < 2000; ++b) {
        const unsigned long long* q = &st[(size_t)((b * CVF_STAMP_WPB + wv) % 4096) * 64];
        bool ok = q[20] != 0;
        for (int i = 21; i <= 26; ++i) ok = ok && q[i] >= q[i - 1] && q[i] - q[i - 1] < 10000000ull;
        if (!ok) continue;
        for (int i = 1; i < 7; ++i) am[i] += double(q[20 + i] - q[19 + i]);
        ++c2;
      }
      printf(" wave %d (%d blocks sampled)\n", wv, c2);
      double tot = 0;
      for (int i = 1; i < 7; ++i) { printf("   %-36s %8.0f cycles\n", mn[i], am[i] / c2); tot += am[i] / c2; }
      printf("   %-36s %8.0f cycles\n", "total", tot);
      {   // phase B in detail: stamp 30 after the dense part, 31.. after each iteration of the slot loop
        double d[9] = {0}; int c3 = 0;
        for (int b = 0; b < 2000; ++b) {
          const unsigned long long* q = &st[(size_t)((b * CVF_STAMP_WPB + wv) % 4096) * 64];
          if (q[30] == 0 || q[30] < q[23] || q[36] < q[30] || q[36] - q[23] > 10000000ull) continue;
          d[0] += double(q[30] - q[23]);
          for (int i = 0; i < 6; ++i) d[1 + i] += double(q[31 + i] - q[30 + i]);
          ++c3;
        }
        printf("   phase B: dense %.0f | iterations", d[0] / c3);
        for (int i = 0; i < 6; ++i) printf(" %.0f", d[1 + i] / c3);
        printf("\n");
      }
    }
  }
  return 0;
}
