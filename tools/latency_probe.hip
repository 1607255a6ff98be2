// Developer tool: calibrates launch floor and dependent-load latency on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty(float* out) { if (threadIdx.x == 999) out[0] = 1; }
template <int N>
__global__ void k_chain(const int* __restrict__ idx, float* out, unsigned long long* st) {
  // N dependent loads (pointer chasing inside a small table), per wave
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int i = (blockIdx.x * 64 + threadIdx.x) & 1023;
#pragma unroll
  for (int n = 0; n < N; ++n) i = idx[i];
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + threadIdx.x] = (float)i;
  if (threadIdx.x == 0) st[blockIdx.x] = t1 - t0;
}
template <class F> float timeit(F f, int n = 200) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 20; ++i) f();
  hipEventRecord(a);
  for (int i = 0; i < n; ++i) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms * 1000.f / n;
}
int main() {
  int* idx; float* out; unsigned long long* st;
  hipMalloc(&idx, 1024 * 4); hipMalloc(&out, 4096 * 64 * 4); hipMalloc(&st, 4096 * 8);
  int h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (i * 37 + 11) & 1023;
  hipMemcpy(idx, h, sizeof(h), hipMemcpyHostToDevice);
  for (int G : {1, 313, 939, 1878}) {
    printf("grid %4d: empty %.2f us", G, timeit([&] { hipLaunchKernelGGL(k_empty, dim3(G), dim3(64), 0, 0, out); }));
    printf("  1 load %.2f", timeit([&] { hipLaunchKernelGGL(k_chain<1>, dim3(G), dim3(64), 0, 0, idx, out, st); }));
    printf("  4 dep loads %.2f", timeit([&] { hipLaunchKernelGGL(k_chain<4>, dim3(G), dim3(64), 0, 0, idx, out, st); }));
    printf("  16 dep loads %.2f us", timeit([&] { hipLaunchKernelGGL(k_chain<16>, dim3(G), dim3(64), 0, 0, idx, out, st); }));
    unsigned long long hs[8]; hipMemcpy(hs, st, sizeof(hs), hipMemcpyDeviceToHost);
    printf("   (16 dep loads = %llu cycles in-kernel)\n", hs[0]);
  }
  return 0;
}
