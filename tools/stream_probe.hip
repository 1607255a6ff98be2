// Developer tool: what read bandwidth does a plain streaming kernel reach on this GPU, by launch shape?
// (sets the achievable ceiling for the align+feature kernel of large molecules).  hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NLOAD, bool NT>
__global__ __launch_bounds__(256) void stream_kernel(const f4* __restrict__ x, size_t n_vec, float* __restrict__ out) {
  // each block streams contiguous chunks of 256*NLOAD float4; grid-stride over chunks
  float acc = 0.0f;
  const size_t chunk = (size_t)256 * NLOAD;
  for (size_t c = blockIdx.x; c * chunk < n_vec; c += gridDim.x) {
    const f4* p = x + c * chunk + threadIdx.x;
    f4 v[NLOAD];
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) v[i] = NT ? __builtin_nontemporal_load(p + 256 * i) : p[256 * i];
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) acc += v[i].x + v[i].y + v[i].z + v[i].w;
  }
  if (acc == 1.2345f) out[0] = acc;
}

// the large-molecule K1's present pattern: lane g reads 48 contiguous bytes as three 16-byte loads (stride 48 B per lane)
template <int UNROLL, bool NT>
__global__ __launch_bounds__(512) void stride48_kernel(const f4* __restrict__ x, size_t n_vec, float* __restrict__ out) {
  float acc = 0.0f;
  const int lane = threadIdx.x & 63;
  const size_t per_wave = 3750;   // one 5000-atom frame per wave
  const size_t wave_id = (size_t)blockIdx.x * 8 + (threadIdx.x >> 6);
  if ((wave_id + 1) * per_wave > n_vec) return;
  const f4* p = x + wave_id * per_wave;
#pragma unroll UNROLL
  for (int g = lane; g < 1250; g += 64) {
    f4 a, b, c;
    if (NT) { a = __builtin_nontemporal_load(p + 3 * g); b = __builtin_nontemporal_load(p + 3 * g + 1); c = __builtin_nontemporal_load(p + 3 * g + 2); }
    else { a = p[3 * g]; b = p[3 * g + 1]; c = p[3 * g + 2]; }
    acc += a.x + b.y + c.z + a.w;
  }
  if (acc == 1.2345f) out[0] = acc;
}
// the same stream with the K1-large loop's other ingredients switched on one by one
template <bool REF, bool SLOT, bool MATH64>
__global__ __launch_bounds__(512) void k1like_kernel(const f4* __restrict__ x, size_t n_vec, const f4* __restrict__ ref,
                                                      const int4* __restrict__ slot, float* __restrict__ out) {
  __shared__ float cap[8][608 * 3];
  const int lane = threadIdx.x & 63, fi = threadIdx.x >> 6;
  const size_t per_wave = 3750;
  const size_t wave_id = (size_t)blockIdx.x * 8 + fi;
  if ((wave_id + 1) * per_wave > n_vec) return;
  const f4* p = x + wave_id * per_wave;
  float accf[12];
  double accd[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) { accf[i] = 0.0f; accd[i] = 0.0; }
#pragma unroll 4
  for (int g = lane; g < 1250; g += 64) {
    const f4 a = p[3 * g], b = p[3 * g + 1], c = p[3 * g + 2];
    f4 pr = {1, 2, 3, 4}, q = {1, 2, 3, 4}, r = {1, 2, 3, 4};
    if (REF) { pr = ref[3 * g]; q = ref[3 * g + 1]; r = ref[3 * g + 2]; }
    const float xs[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
    const float rs[12] = {pr.x, pr.y, pr.z, pr.w, q.x, q.y, q.z, q.w, r.x, r.y, r.z, r.w};
    if (MATH64) {
#pragma unroll
      for (int at = 0; at < 4; ++at)
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
          for (int j = 0; j < 3; ++j) accd[3 * d + j] = fma((double)xs[3 * at + d], (double)rs[3 * at + j], accd[3 * d + j]);
    } else {
#pragma unroll
      for (int at = 0; at < 4; ++at)
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
          for (int j = 0; j < 3; ++j) accf[3 * d + j] = fmaf(xs[3 * at + d], rs[3 * at + j], accf[3 * d + j]);
    }
    if (SLOT) {
      const int4 sl = slot[g];
      if (sl.x >= 0) { cap[fi][3 * sl.x] = a.x; cap[fi][3 * sl.x + 1] = a.y; cap[fi][3 * sl.x + 2] = a.z; }
      if (sl.y >= 0) { cap[fi][3 * sl.y] = a.w; cap[fi][3 * sl.y + 1] = b.x; cap[fi][3 * sl.y + 2] = b.y; }
      if (sl.z >= 0) { cap[fi][3 * sl.z] = b.z; cap[fi][3 * sl.z + 1] = b.w; cap[fi][3 * sl.z + 2] = c.x; }
      if (sl.w >= 0) { cap[fi][3 * sl.w] = c.y; cap[fi][3 * sl.w + 1] = c.z; cap[fi][3 * sl.w + 2] = c.w; }
    }
  }
  float t = 0;
#pragma unroll
  for (int i = 0; i < 12; ++i) t += accf[i] + (float)accd[i];
  if (SLOT) t += cap[fi][lane];
  if (t == 1.2345f) out[0] = t;
}

// coalesced non-temporal loads + transpose through LDS to the 48-bytes-per-lane form
template <bool NT>
__global__ __launch_bounds__(256) void transpose_kernel(const f4* __restrict__ x, size_t n_vec, float* __restrict__ out) {
  __shared__ f4 stage[4][192];
  float acc = 0.0f;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t chunk = 192 * 4;   // float4 per block iteration (4 waves x 3 KB)
  for (size_t c0 = (size_t)blockIdx.x * chunk; c0 + chunk <= n_vec; c0 += (size_t)gridDim.x * chunk) {
    const f4* p = x + c0 + w * 192 + lane;
    f4 a, b, c;
    if (NT) { a = __builtin_nontemporal_load(p); b = __builtin_nontemporal_load(p + 64); c = __builtin_nontemporal_load(p + 128); }
    else { a = p[0]; b = p[64]; c = p[128]; }
    stage[w][lane] = a; stage[w][64 + lane] = b; stage[w][128 + lane] = c;
    // wave-private region: no barrier needed, the LDS ops of one wave are ordered
    const f4 ta = stage[w][3 * lane], tb = stage[w][3 * lane + 1], tc = stage[w][3 * lane + 2];
    acc += ta.x + tb.y + tc.z + ta.w;
  }
  if (acc == 1.2345f) out[0] = acc;
}

// copy: the read:write = 1:1 mix of the small-molecule align+feature kernel (coordinates in, features out)
template <int NLOAD, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_kernel(const f4* __restrict__ x, f4* __restrict__ y, size_t n_vec) {
  const size_t chunk = (size_t)256 * NLOAD;
  for (size_t c = blockIdx.x; (c + 1) * chunk <= n_vec; c += gridDim.x) {
    const f4* p = x + c * chunk + threadIdx.x;
    f4* q = y + c * chunk + threadIdx.x;
    f4 v[NLOAD];
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) v[i] = NTL ? __builtin_nontemporal_load(p + 256 * i) : p[256 * i];
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      if (NTS) __builtin_nontemporal_store(v[i], q + 256 * i);
      else q[256 * i] = v[i];
    }
  }
}
template <int NLOAD, bool NTL, bool NTS>
void run_copy(const f4* x, f4* y, size_t n_vec, int blocks, const char* tag) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((copy_kernel<NLOAD, NTL, NTS>), dim3(blocks), dim3(256), 0, 0, x, y, n_vec);
  (void)hipEventRecord(e0, 0);
  const int reps = 10;
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((copy_kernel<NLOAD, NTL, NTS>), dim3(blocks), dim3(256), 0, 0, x, y, n_vec);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("copy %-22s %5zu MB blocks=%6d loads/thread=%2d  %7.0f GB/s (read + write)\n", tag, n_vec * 16 >> 20, blocks, NLOAD,
         2.0 * n_vec * 16.0 * reps / (ms * 1e-3) * 1e-9);
}

// the plainest copy there is: every thread moves NV float4, one chunk per block (no persistent loop) - the shape behind
// the 6.29 TB/s "float4 copy" of MI355X_MICROARCH.md, to reconcile with the persistent-grid figures above
template <int NV, int THREADS>
__global__ __launch_bounds__(THREADS) void copy_once_kernel(const f4* __restrict__ x, f4* __restrict__ y, size_t n_vec) {
  const size_t base = (size_t)blockIdx.x * THREADS * NV + threadIdx.x;
  f4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = base + (size_t)THREADS * i < n_vec ? x[base + (size_t)THREADS * i] : f4{0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (base + (size_t)THREADS * i < n_vec) y[base + (size_t)THREADS * i] = v[i];
}
template <int NV, int THREADS>
void run_copy_once(const f4* x, f4* y, size_t n_vec) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const unsigned blocks = (unsigned)((n_vec + (size_t)THREADS * NV - 1) / ((size_t)THREADS * NV));
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((copy_once_kernel<NV, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, x, y, n_vec);
  (void)hipEventRecord(e0, 0);
  const int reps = 10;
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((copy_once_kernel<NV, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, x, y, n_vec);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("copy one chunk per block       %5zu MB threads=%4d float4/thread=%2d blocks=%8u  %7.0f GB/s (read + write)\n", n_vec * 16 >> 20,
         THREADS, NV, blocks, 2.0 * n_vec * 16.0 * reps / (ms * 1e-3) * 1e-9);
}

template <int NLOAD, bool NT>
void run(const f4* x, size_t n_vec, float* out, int blocks, const char* tag) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((stream_kernel<NLOAD, NT>), dim3(blocks), dim3(256), 0, 0, x, n_vec, out);
  (void)hipEventRecord(e0, 0);
  const int reps = 5;
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((stream_kernel<NLOAD, NT>), dim3(blocks), dim3(256), 0, 0, x, n_vec, out);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s blocks=%6d loads/thread=%2d  %7.0f GB/s\n", tag, blocks, NLOAD, n_vec * 16.0 * reps / (ms * 1e-3) * 1e-9);
}

int main() {
  const size_t bytes = (size_t)6 << 30;
  const size_t n_vec = bytes / 16;
  f4* x; float* out;
  (void)hipMalloc(&x, bytes); (void)hipMalloc(&out, 4);
  (void)hipMemset(x, 0, bytes);
  {
    f4* y;
    (void)hipMalloc(&y, (size_t)3 << 30);
    for (size_t mb : {264, 3072}) {
      const size_t nv = (mb << 20) / 16;
      run_copy<4, false, false>(x, y, nv, 256 * 8, "plain");
      run_copy<8, false, false>(x, y, nv, 256 * 8, "plain");
      run_copy<8, false, false>(x, y, nv, 256 * 16, "plain");
      run_copy<8, true, false>(x, y, nv, 256 * 8, "nt load");
      run_copy<8, false, true>(x, y, nv, 256 * 8, "nt store");
      run_copy<8, true, true>(x, y, nv, 256 * 8, "nt load+store");
      run_copy<16, true, true>(x, y, nv, 256 * 8, "nt load+store");
      run_copy_once<1, 256>(x, y, nv);
      run_copy_once<2, 256>(x, y, nv);
      run_copy_once<4, 256>(x, y, nv);
      run_copy_once<1, 1024>(x, y, nv);
      run_copy_once<4, 1024>(x, y, nv);
    }
    (void)hipFree(y);
    if (getenv("COPY_ONLY")) return 0;
  }
  const int grids[] = {256 * 8};
  for (int g : grids) {
    run<4, false>(x, n_vec, out, g, "plain");
    run<8, false>(x, n_vec, out, g, "plain");
    run<16, false>(x, n_vec, out, g, "plain");
    run<8, true>(x, n_vec, out, g, "nontemporal");
    run<16, true>(x, n_vec, out, g, "nontemporal");
  }
  for (int pass = 0; pass < 2; ++pass) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int reps = 5;
    float ms;
    const unsigned gb = (unsigned)(n_vec / 3750 / 8);
#define TIME(tag, launch, bytes_)                                                     \
    launch; (void)hipEventRecord(e0, 0); for (int it = 0; it < reps; ++it) launch;    \
    (void)hipEventRecord(e1, 0); (void)hipDeviceSynchronize(); (void)hipEventElapsedTime(&ms, e0, e1); \
    printf("%-44s %7.0f GB/s\n", tag, (double)(bytes_) * reps / (ms * 1e-3) * 1e-9);
    const double b48 = (double)gb * 8 * 3750 * 16;
    TIME("stride-48 (present K1-large), plain, unroll 4", hipLaunchKernelGGL((stride48_kernel<4, false>), dim3(gb), dim3(512), 0, 0, x, n_vec, out), b48)
    TIME("stride-48, non-temporal, unroll 4", hipLaunchKernelGGL((stride48_kernel<4, true>), dim3(gb), dim3(512), 0, 0, x, n_vec, out), b48)
    TIME("stride-48, plain, unroll 8", hipLaunchKernelGGL((stride48_kernel<8, false>), dim3(gb), dim3(512), 0, 0, x, n_vec, out), b48)
    {
      static f4* dref = nullptr; static int4* dslot = nullptr;
      if (!dref) {
        (void)hipMalloc(&dref, 3750 * 16); (void)hipMalloc(&dslot, 1250 * 16);
        (void)hipMemset(dref, 0, 3750 * 16);
        std::vector<int> hs(5000, -1);
        for (int i = 0; i < 608; ++i) hs[(i * 8209) % 5000] = i;
        (void)hipMemcpy(dslot, hs.data(), 5000 * 4, hipMemcpyHostToDevice);
      }
      TIME("k1like: x only, fp32 math", hipLaunchKernelGGL((k1like_kernel<false, false, false>), dim3(gb), dim3(512), 0, 0, x, n_vec, dref, dslot, out), b48)
      TIME("k1like: x + ref, fp32 math", hipLaunchKernelGGL((k1like_kernel<true, false, false>), dim3(gb), dim3(512), 0, 0, x, n_vec, dref, dslot, out), b48)
      TIME("k1like: x + ref, fp64 math", hipLaunchKernelGGL((k1like_kernel<true, false, true>), dim3(gb), dim3(512), 0, 0, x, n_vec, dref, dslot, out), b48)
      TIME("k1like: x + slot capture, fp32 math", hipLaunchKernelGGL((k1like_kernel<false, true, false>), dim3(gb), dim3(512), 0, 0, x, n_vec, dref, dslot, out), b48)
      TIME("k1like: x + ref + slot capture, fp32 math", hipLaunchKernelGGL((k1like_kernel<true, true, false>), dim3(gb), dim3(512), 0, 0, x, n_vec, dref, dslot, out), b48)
    }
    const double bt = (double)(n_vec / 768) * 768 * 16;
    TIME("coalesced + LDS transpose, plain, 4096 blocks", hipLaunchKernelGGL((transpose_kernel<false>), dim3(4096), dim3(256), 0, 0, x, n_vec, out), bt)
    TIME("coalesced + LDS transpose, nt, 4096 blocks", hipLaunchKernelGGL((transpose_kernel<true>), dim3(4096), dim3(256), 0, 0, x, n_vec, out), bt)
    TIME("coalesced + LDS transpose, nt, 16384 blocks", hipLaunchKernelGGL((transpose_kernel<true>), dim3(16384), dim3(256), 0, 0, x, n_vec, out), bt)
  }
  return 0;
}
