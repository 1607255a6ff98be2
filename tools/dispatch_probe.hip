// Developer tool: what the DISPATCH of a launch costs by its shape - an (almost) empty kernel launched as many small workgroups or as
// few large ones with the same number of waves and the same LDS per compute unit.  hipcc -O3 --offload-arch=gfx950 tools/dispatch_probe.hip -o tools/build/dispatch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void nop_kernel(float* out, int sleep) {
  extern __shared__ float lds[];
  for (int i = 0; i < sleep; ++i) __builtin_amdgcn_s_sleep(16);
  if (out != nullptr && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) out[0] = lds[0];
}
int main() {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  struct Shape { int blocks, threads, lds; const char* what; };
  const Shape shapes[] = {{1250, 192, 26 * 1024, "front: 1250 x 3 waves, 26 KB"}, {250, 960, 130 * 1024, "front: 250 x 15 waves, 130 KB"},
                          {939, 256, 27 * 1024, "back: 939 x 4 waves, 27 KB"},   {235, 1024, 107 * 1024, "back: 235 x 16 waves, 107 KB"},
                          {313, 768, 80 * 1024, "back: 313 x 12 waves, 80 KB"},  {207, 1024, 8 * 1024, "slab: 207 x 16 waves"},
                          {1, 1024, 8 * 1024, "finish: 1 x 16 waves"},            {3750, 64, 8 * 1024, "3750 x 1 wave"}};
  for (const Shape& s : shapes) {
    hipFuncSetAttribute((const void*)nop_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, s.lds);
    for (int sleep : {0, 20}) {   // 20 x s_sleep(16) ~ 20 k cycles ~ 8 us of "work" per wave
      for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(nop_kernel, dim3(s.blocks), dim3(s.threads), s.lds, 0, nullptr, sleep);
      hipDeviceSynchronize();
      const int reps = 200;
      hipEventRecord(e0);
      for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(nop_kernel, dim3(s.blocks), dim3(s.threads), s.lds, 0, nullptr, sleep);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      printf("%-34s sleep %2d: %7.2f us per launch (back to back)\n", s.what, sleep, ms / reps * 1e3);
    }
  }
  return 0;
}
