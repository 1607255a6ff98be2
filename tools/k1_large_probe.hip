// Developer tool: phase timing (s_memtime stamps) and throughput of the large-molecule align+feature kernel at the
// config-5 shape (5000 atoms, 32 positions + 96 dihedrals + 96 distances).  Build on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCVF_STAMPS -Iinclude -Icolvars-finder_amd/csrc -Wno-pass-failed \
//       tools/k1_large_probe.hip colvars-finder_amd/csrc/stats.hip -o /tmp/k1_large_probe
#include "../colvars-finder_amd/csrc/k1_large.hip"
#include <cstdio>
#include <random>
#include <vector>
#include <algorithm>

int main(int argc, char** argv) {
  const int N = 5000, nc = 3 * N;
  const int64_t B = argc > 1 ? atoll(argv[1]) : 20000;
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
  {
    const size_t ldsc = ((size_t)kGroup * pp.n_slot * 3 + (size_t)pp.d_r * kGroup) * sizeof(float);
    int nb = -1;
    (void)hipFuncSetAttribute((const void*)k1_large_slice_kernel<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsc);
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k1_large_slice_kernel<3, false>, 64 * kGroup, ldsc);
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, (const void*)k1_large_slice_kernel<3, false>);
    printf("occupancy query: %d blocks/CU (err %d), dynamic LDS %zu B, static %zu B, regs %d\n", nb, (int)e, ldsc, fa.sharedSizeBytes, fa.numRegs);
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int it = 0; it < 2; ++it) cvf_k1_large_launch(&pp, dx, B, dfeat, nullptr, daux, dslot, nullptr);
  (void)hipEventRecord(e0, nullptr);
  const int reps = 10;
  for (int it = 0; it < reps; ++it) {
    int rc = cvf_k1_large_launch(&pp, dx, B, dfeat, nullptr, daux, dslot, nullptr);
    if (rc) { printf("failed: %s\n", cvf_last_error()); return 1; }
  }
  (void)hipEventRecord(e1, nullptr);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double us = 1e3 * ms / reps, bpf = 12.0 * N + 4 + 4.0 * d_r;
  printf("B=%lld N=%d n_slot=%d d_r=%d: %.1f us/launch, %.0f GB/s algorithmic (%.0f B/frame)\n", (long long)B, N, pp.n_slot, d_r, us,
         bpf * B / us * 1e-3, bpf);
  std::vector<unsigned long long> st(64 * 4096);
  if (B / 8 >= 4 * 256) {   // the pipelined kernel: per-role phase sums (PIPE_T), one launch on zeroed counters
    std::fill(st.begin(), st.end(), 0ull);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), st.data(), st.size() * 8);
    cvf_k1_large_launch(&pp, dx, B, dfeat, nullptr, argc > 2 ? daux : nullptr, argc > 2 ? dslot : nullptr, nullptr);
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8);
    const double groups_per_wg = double((B + 63) / 64 * 8) / 256.0;
    const char* sn[3] = {"", "stream 8 frames", "wait at the barrier"};
    const char* tn[9] = {"", "slot copy + sums", "solve", "features + aux", "wait for the tail waves", "flush", "", "wait at the barrier"};
    for (int wv : {0, 1, 2, 3, 4, 5, 6, 7}) {
      double a1 = 0, a2 = 0;
      for (int b = 0; b < 256; ++b) { a1 += st[(b * 12 + wv) * 64 + 1]; a2 += st[(b * 12 + wv) * 64 + 2]; }
      printf("   streaming wave %d: %s %8.0f | %s %8.0f cycles per group\n", wv, sn[1], a1 / 256 / groups_per_wg, sn[2], a2 / 256 / groups_per_wg);
    }
    {
      const int order0[5] = {1, 2, 3, 4, 8};
      const char* names0[5] = {"barrier exit", "sums", "solve", "aux + position features", "wait at the barrier"};
      printf("   tail wave 0:");
      for (int q = 0; q < 5; ++q) {
        double a = 0;
        for (int b = 0; b < 256; ++b) a += st[(b * 12 + 8) * 64 + order0[q]];
        printf(" %s %.0f |", names0[q], a / 256 / groups_per_wg);
      }
      printf(" cycles per group\n");
    }
    for (int wv : {9, 11}) {
      printf("   tail wave %d:", wv - 8);
      const int order[6] = {1, 2, 3, 5, 6, 8};
      const char* names[6] = {"barrier exit", "slot copy", "other features", "tail waves meet", "gather + flush", "wait at the barrier"};
      for (int q = 0; q < 6; ++q) {
        double a = 0;
        for (int b = 0; b < 256; ++b) a += st[(b * 12 + wv) * 64 + order[q]];
        printf(" %s %.0f |", names[q], a / 256 / groups_per_wg);
      }
      printf(" cycles per group\n");
    }
    return 0;
  }
  (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8);
  const char* nm[9] = {"", "static ref/slot loads", "8-frame stream + reduce", "barrier", "sum over waves", "solve", "aux + slot copy", "features", "flush"};
  double acc[9] = {0}; int n = 0;
  for (int b = 0; b < 2000; ++b) {
    const unsigned long long* s = &st[(b * 2) % 4096 * 64];
    bool ok = s[0] != 0;
    for (int i = 1; i < 9; ++i) ok = ok && s[i] >= s[i - 1] && s[i] - s[i - 1] < 10000000ull;
    if (!ok) continue;
    for (int i = 1; i < 9; ++i) acc[i] += double(s[i] - s[i - 1]);
    ++n;
  }
  double tot = 0;
  for (int i = 1; i < 9; ++i) { printf("   %-26s %8.0f cycles\n", nm[i], acc[i] / n); tot += acc[i] / n; }
  printf("   total %8.0f cycles over %d blocks (wave 0)\n", tot, n);
  {  // residency: how many blocks started before the first one finished?
    const int nb = (int)((B + 7) / 8) < 2048 ? (int)((B + 7) / 8) : 2048;
    std::vector<unsigned long long> starts;
    unsigned long long first_end = ~0ull;
    for (int b = 0; b < nb; ++b) {
      const unsigned long long* q = &st[(b * 2) % 4096 * 64];
      starts.push_back(q[0]);
      if (q[8] > q[0] && q[8] < first_end) first_end = q[8];
    }
    std::sort(starts.begin(), starts.end());
    int early = 0;
    for (auto v : starts) early += v < first_end;
    printf("   %d of %d blocks started before the first block ended (%.2f per CU)\n", early, nb, early / 256.0);
  }
  return 0;
}
