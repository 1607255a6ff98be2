// Developer tool (not part of the library): the streaming alignment + feature kernel (k1_stream_kernel) OUT OF CACHE at the dipeptide
// shape - launch times with HIP events, and with -DCVF_STAMPS the phases of one tile of every wave (its third).
//   hipcc -O3 -std=c++17 -fno-slp-vectorize --offload-arch=gfx950 -DCVF_STAMPS -Iinclude -Icolvars-finder_amd/csrc -Wno-pass-failed \
//       tools/k1_stream_probe.hip -Lcolvars-finder_amd/colvarsfinder -lcvf_hip -Wl,-rpath,$ORIGIN/../../colvars-finder_amd/colvarsfinder -o tools/build/k1_stream_probe
#include "../colvars-finder_amd/csrc/k1_align.hip"
#include <cstdio>
#include <random>
#include <vector>

__global__ void fill_frames(float* x, const float* ref, int nc, int64_t n) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n * nc) return;
  unsigned h = (unsigned)(i * 2654435761u) ^ (unsigned)(i >> 17);
  h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
  x[i] = ref[i % nc] + 0.6f * ((h & 0xffff) / 65536.0f - 0.5f) + 0.01f * (float)((i / nc) % 97);
}

int main() {
  const int N = 22, nc = 3 * N;
  std::mt19937 rng(3);
  std::normal_distribution<float> G(0.0f, 1.0f);
  std::vector<float> ref(nc), refc(nc);
  for (auto& v : ref) v = 2.0f * G(rng);
  float cm[3] = {0, 0, 0};
  for (int a = 0; a < N; ++a) for (int d = 0; d < 3; ++d) cm[d] += ref[3 * a + d] / N;
  for (int a = 0; a < N; ++a) for (int d = 0; d < 3; ++d) refc[3 * a + d] = ref[3 * a + d] - cm[d];
  std::vector<int32_t> align(N), rec(6 * N);
  for (int a = 0; a < N; ++a) { align[a] = a; int32_t r[6] = {CVF_FEAT_POSITION, a, 0, 0, 0, 3 * a}; for (int i = 0; i < 6; ++i) rec[6 * a + i] = r[i]; }
  int32_t *dal, *drec; float *dref, *dref0;
  (void)hipMalloc(&dal, N * 4); (void)hipMalloc(&drec, 6 * N * 4); (void)hipMalloc(&dref, nc * 4); (void)hipMalloc(&dref0, nc * 4);
  (void)hipMemcpy(dal, align.data(), N * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(drec, rec.data(), 6 * N * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dref, refc.data(), nc * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dref0, ref.data(), nc * 4, hipMemcpyHostToDevice);
  cvf_pp_desc pp = {};
  pp.mode = CVF_PP_ALIGN; pp.n_coord = nc; pp.n_align = N; pp.n_rec = N; pp.d_r = nc; pp.has_position = 1;
  pp.flags = CVF_PP_ALIGN_CONTIG | CVF_PP_PURE_POSITION;
  pp.align_idx = dal; pp.ref_c = dref; pp.rec = drec;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int64_t B : {(int64_t)1000000, (int64_t)4000000}) {
    const int64_t T = (B + 63) / 64;
    float *dx, *dfeat, *daux;
    (void)hipMalloc(&dx, B * nc * 4); (void)hipMalloc(&dfeat, T * nc * 64 * 4); (void)hipMalloc(&daux, T * 18 * 64 * 4);
    fill_frames<<<(unsigned)((B * nc + 255) / 256), 256>>>(dx, dref0, nc, B);
    (void)hipDeviceSynchronize();
    for (int mode = 0; mode < 3; ++mode) {   // tiled features only | tiled + aux | row-major features only
      float* ft = mode < 2 ? dfeat : nullptr;
      float* fr = mode == 2 ? dfeat : nullptr;
      float* ax = mode == 1 ? daux : nullptr;
      for (int it = 0; it < 3; ++it) cvf_align_feature_fwd(&pp, dx, B, ft, fr, ax, nullptr, nullptr);
      (void)hipEventRecord(e0, nullptr);
      const int reps = 10;
      for (int it = 0; it < reps; ++it) cvf_align_feature_fwd(&pp, dx, B, ft, fr, ax, nullptr, nullptr);
      (void)hipEventRecord(e1, nullptr);
      (void)hipDeviceSynchronize();
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      const double us = 1e3 * ms / reps;
      printf("B=%lld %s: %.1f us/launch = %.0f GB/s of 532 B/frame = %.3f of 8 TB/s\n", (long long)B,
             mode == 0 ? "tiled features only" : mode == 1 ? "tiled features + aux rows" : "row-major features only", us, 532.0 * B / us * 1e-3,
             532.0 * B / us * 1e-3 / 8000.0);
#ifdef CVF_STAMPS
      std::vector<unsigned long long> st(64 * 4096);
      (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8);
      const char* nm[7] = {"", "LDS write of the prefetched tile", "next tile's loads issued", "centroid + covariance", "rotation solve", "aux rows", "features + stores"};
      double acc[7] = {0}; int n = 0;
      for (int b = 0; b < 2048; ++b) {
        const unsigned long long* s = &st[(size_t)(b * CVF_STAMP_WPB) % 4096 * 64];
        if (s[6] == 0 || s[0] == 0) continue;
        for (int i = 1; i < 7; ++i) acc[i] += double(s[i] - s[i - 1]);
        ++n;
      }
      double tot = 0;
      for (int i = 1; i < 7; ++i) { printf("     %-34s %7.0f cycles\n", nm[i], acc[i] / (n ? n : 1)); tot += acc[i] / (n ? n : 1); }
      printf("     one tile %7.0f cycles (%d waves)\n", tot, n);
      std::fill(st.begin(), st.end(), 0ull);
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), st.data(), st.size() * 8);
#endif
    }
    (void)hipFree(dx); (void)hipFree(dfeat); (void)hipFree(daux);
  }
  return 0;
}
