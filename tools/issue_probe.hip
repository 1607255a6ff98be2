// Developer tool: issue cost (cycles per instruction, one wave alone on its SIMD, independent instructions) of the
// fp64 / conversion / packed-fp32 instructions the per-frame alignment solve is made of.
//   hipcc -O3 --offload-arch=gfx950 tools/issue_probe.hip -o /tmp/issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define PROBE(NAME, DECL, BODY, SINK)                                                              \
  __global__ void NAME(unsigned long long* out, double seed) {                                    \
    DECL                                                                                           \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                    \
    for (int it = 0; it < 256; ++it) { REP8(BODY) REP8(BODY) }                                     \
    __builtin_amdgcn_s_waitcnt(0);                                                                 \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                    \
    if (threadIdx.x == 0) out[0] = t1 - t0;                                                        \
    SINK                                                                                           \
  }

#define DECL_D double a[8]; for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x; double b = seed * 1.0000001;
#define SINK_D double s = 0; for (int i = 0; i < 8; ++i) s += a[i]; if (s == 1.2345) out[1] = (unsigned long long)s;
#define DECL_F float a[8]; for (int i = 0; i < 8; ++i) a[i] = (float)seed + i + threadIdx.x; float b = (float)seed * 1.0001f;
#define SINK_F float s = 0; for (int i = 0; i < 8; ++i) s += a[i]; if (s == 1.2345f) out[1] = (unsigned long long)s;

#define B_FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
#define B_MUL64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define B_ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define B_RSQ64(i) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[i]));
#define B_RCP64(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
#define B_SQRT64(i) asm volatile("v_sqrt_f64 %0, %0" : "+v"(a[i]));
#define B_FMA32(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
#define B_RSQ32(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
#define B_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));

PROBE(p_fma64, DECL_D, B_FMA64, SINK_D)
PROBE(p_mul64, DECL_D, B_MUL64, SINK_D)
PROBE(p_add64, DECL_D, B_ADD64, SINK_D)
PROBE(p_rsq64, DECL_D, B_RSQ64, SINK_D)
PROBE(p_rcp64, DECL_D, B_RCP64, SINK_D)
PROBE(p_sqrt64, DECL_D, B_SQRT64, SINK_D)
PROBE(p_fma32, DECL_F, B_FMA32, SINK_F)
PROBE(p_rsq32, DECL_F, B_RSQ32, SINK_F)
PROBE(p_cndmask, DECL_F, B_CNDMASK, SINK_F)

// conversions and packed fp32 need mixed register widths
__global__ void p_cvt_f64_f32(unsigned long long* out, double seed) {
  float f[8]; double d[8];
  for (int i = 0; i < 8; ++i) f[i] = (float)seed + i + threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 256; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  double s = 0; for (int i = 0; i < 8; ++i) s += d[i]; if (s == 1.2345) out[1] = 1;
}
__global__ void p_cvt_f32_f64(unsigned long long* out, double seed) {
  float f[8]; double d[8];
  for (int i = 0; i < 8; ++i) d[i] = seed + i + threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 256; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  float s = 0; for (int i = 0; i < 8; ++i) s += f[i]; if (s == 1.2345f) out[1] = 1;
}
__global__ void p_pk_fma32(unsigned long long* out, double seed) {
  double d[8];   // a register pair = two packed floats
  for (int i = 0; i < 8; ++i) d[i] = seed + i + threadIdx.x;
  const double b = seed;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 256; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(d[i]) : "v"(b));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  double s = 0; for (int i = 0; i < 8; ++i) s += d[i]; if (s == 1.2345) out[1] = 1;
}

int main() {
  unsigned long long* d;
  (void)hipMalloc(&d, 16);
  struct { const char* name; void (*k)(unsigned long long*, double); } tests[] = {
      {"v_fma_f32", p_fma32}, {"v_pk_fma_f32", p_pk_fma32}, {"v_rsq_f32", p_rsq32}, {"v_cndmask_b32", p_cndmask},
      {"v_fma_f64", p_fma64}, {"v_mul_f64", p_mul64}, {"v_add_f64", p_add64}, {"v_rsq_f64", p_rsq64}, {"v_rcp_f64", p_rcp64},
      {"v_sqrt_f64", p_sqrt64}, {"v_cvt_f64_f32", p_cvt_f64_f32}, {"v_cvt_f32_f64", p_cvt_f32_f64}};
  for (auto& t : tests) {
    unsigned long long best = ~0ull;
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(t.k, dim3(1), dim3(64), 0, 0, d, 1.5);
      unsigned long long h[2];
      (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      if (h[0] < best) best = h[0];
    }
    printf("%-16s %6.2f s_memtime ticks per instruction (4096 instructions, 8 independent chains)\n", t.name, best / 4096.0);
  }
  return 0;
}
