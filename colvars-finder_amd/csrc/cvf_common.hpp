// Shared device/host helpers for the gfx950 kernels of libcvf_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cvf.h"

#define CVF_WAVE 64

void cvf_set_error(const char* fmt, ...);
int cvf_check_launch(const char* what);

#define CVF_REQUIRE(cond, ...)      \
  do {                              \
    if (!(cond)) {                  \
      cvf_set_error(__VA_ARGS__);   \
      return -1;                    \
    }                               \
  } while (0)

static inline int64_t cvf_ntiles(int64_t B) { return (B + CVF_TILE - 1) / CVF_TILE; }
// compute units of the current device (256 on MI355X), asked once
static inline int cvf_cu_count() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) v = 256;
    return v;
  }();
  return n;
}

// ------------------------------------------------------------------------------------
// wave helpers
// ------------------------------------------------------------------------------------
// Wave-wide sums on the DPP cross-lane path (no LDS crossbar round trips): an inclusive scan inside each row of 16
// lanes (row_shr 1, 2, 4, 8; lanes shifted in from outside the row read 0), then row_bcast:15 / row_bcast:31 carry
// the row totals into lane 63, whose value is broadcast through an SGPR.  The order of the additions is fixed, so the
// result is reproducible run to run; every lane receives the total.  (__shfl_xor compiles to ds_bpermute_b32: six
// dependent ~100-cycle hops per sum, twice that for a double.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_movf(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_movd(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, ROW_MASK, 0xf, false);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_movd<0x111, 0xf>(v);   // row_shr:1
  v += dpp_movd<0x112, 0xf>(v);   // row_shr:2
  v += dpp_movd<0x114, 0xf>(v);   // row_shr:4
  v += dpp_movd<0x118, 0xf>(v);   // row_shr:8   -> lane 15 of each row holds the row's sum
  v += dpp_movd<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
  v += dpp_movd<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, 63);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 63);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float wave_sumf(float v) {
  v += dpp_movf<0x111, 0xf>(v);
  v += dpp_movf<0x112, 0xf>(v);
  v += dpp_movf<0x114, 0xf>(v);
  v += dpp_movf<0x118, 0xf>(v);
  v += dpp_movf<0x142, 0xa>(v);
  v += dpp_movf<0x143, 0xc>(v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// tanh in 7 instructions, absolute error ~6e-8 (fp32 rounding of t):  t = exp(-2|x|) in (0,1],
// tanh|x| = (1 - t) / (1 + t).  (libm's tanhf is two divergent branches of ~35 instructions, and a 20-wide net
// evaluates 60 of them per frame; v_exp_f32 / v_rcp_f32 are ~1 ulp.)  The parity bar on the loss is 1e-5 relative.
__device__ __forceinline__ float cvf_tanh(float x) {
  const float t = __builtin_amdgcn_exp2f(-2.8853900817779268f * fabsf(x));   // exp(-2|x|) = 2^(-2 log2(e) |x|)
  const float r = (1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t);
  return copysignf(r, x);
}

// Activations of the chain kernels (cvf_mlp_desc.act codes, include/cvf.h) and their derivative expressed through the
// OUTPUT h = act(z) - the backward pass keeps activations, not pre-activations.  `kind` is uniform over a launch.
__device__ __forceinline__ float cvf_act(int kind, float z) {
  switch (kind) {
    case CVF_ACT_TANH: return cvf_tanh(z);
    case CVF_ACT_SIGMOID: return 1.0f / (1.0f + expf(-z));
    case CVF_ACT_RELU: return fmaxf(z, 0.0f);
    case CVF_ACT_ELU: return z > 0.0f ? z : expm1f(z);
    case CVF_ACT_LEAKY_RELU: return z > 0.0f ? z : 0.01f * z;
    case CVF_ACT_SOFTPLUS: return z > 20.0f ? z : log1pf(expf(z));
    default: return z;
  }
}
// f''(z) through the output h (the eigenfunction kernels' generator mode differentiates the nets twice)
__device__ __forceinline__ float cvf_act_d2(int kind, float h) {
  switch (kind) {
    case CVF_ACT_TANH: return -2.0f * h * (1.0f - h * h);
    case CVF_ACT_SIGMOID: return h * (1.0f - h) * (1.0f - 2.0f * h);
    case CVF_ACT_ELU: return h > 0.0f ? 0.0f : h + 1.0f;
    case CVF_ACT_SOFTPLUS: {
      const float s = h > 20.0f ? 1.0f : 1.0f - expf(-h);
      return s * (1.0f - s);
    }
    default: return 0.0f;   // piecewise linear (ReLU, LeakyReLU) or none
  }
}
__device__ __forceinline__ float cvf_act_d1(int kind, float h) {
  switch (kind) {
    case CVF_ACT_TANH: return 1.0f - h * h;
    case CVF_ACT_SIGMOID: return h * (1.0f - h);
    case CVF_ACT_RELU: return h > 0.0f ? 1.0f : 0.0f;
    case CVF_ACT_ELU: return h > 0.0f ? 1.0f : h + 1.0f;
    case CVF_ACT_LEAKY_RELU: return h > 0.0f ? 1.0f : 0.01f;
    case CVF_ACT_SOFTPLUS: return h > 20.0f ? 1.0f : 1.0f - expf(-h);   // 1 - e^{-softplus(z)} = sigmoid(z)
    default: return 1.0f;
  }
}

// Developer aid: the tools/*.hip probes compile a kernel file with -DCVF_STAMPS to read s_memtime at phase
// boundaries of one wave per block; in the shipped library the macro is empty.
#ifdef CVF_STAMPS
#ifndef CVF_STAMP_WPB
#define CVF_STAMP_WPB 2   // waves per block that get a stamp row of their own
#endif
static __device__ unsigned long long g_stamps[64 * 4096];
#define CVF_STAMP(i)                                                                              \
  do {                                                                                            \
    if ((threadIdx.x & 63) == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 4096 && (threadIdx.x >> 6) < CVF_STAMP_WPB)   \
      g_stamps[(blockIdx.x * CVF_STAMP_WPB + (threadIdx.x >> 6)) % 4096 * 64 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
// the same with the chip-wide 100 MHz counter (s_memtime is per compute unit: not comparable between workgroups)
#define CVF_STAMP_RT(i)                                                                           \
  do {                                                                                            \
    if ((threadIdx.x & 63) == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 4096 && (threadIdx.x >> 6) < CVF_STAMP_WPB)   \
      g_stamps[(blockIdx.x * CVF_STAMP_WPB + (threadIdx.x >> 6)) % 4096 * 64 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define CVF_STAMP(i) do {} while (0)
#define CVF_STAMP_RT(i) do {} while (0)
#endif

// ------------------------------------------------------------------------------------
// Stage one tile (64 frames) of a row-major [B][nc] fp32 array into LDS as
// lds[frame * stride + j], stride odd so that per-lane reads (lane = frame) are
// conflict-free.  Global reads are coalesced (the tile is one contiguous 64*nc*4 B run).
// Frames past B replicate frame B-1 (their weight is forced to 0 by the callers).
// ------------------------------------------------------------------------------------
// nc = 2 (mod 4): stride nc itself - lane l's row starts at bank 2 l (mod 64), distinct over each 32-lane half (the unit
// in which ds_read_b32 / ds_read_b64 are serviced), and the tile is then a plain copy of memory (16-byte LDS writes, no
// index arithmetic); otherwise the next odd number.
__host__ __device__ __forceinline__ int x_tile_stride(int nc) { return (nc & 3) == 2 ? nc : (nc | 1); }

// `tid` of `nthreads` (a multiple of 64) threads share the work; callers follow with a barrier.
// `frames` (a multiple of 4) frames per tile: 64, or 16 for the four-lanes-per-frame kernels.
template <int kBatch = 18>
__device__ __forceinline__ void load_x_tile(const float* __restrict__ x, int64_t B, int nc, int64_t tile, float* lds,
                                            int tid, int nthreads = CVF_WAVE, int frames = CVF_TILE) {
  const int stride = x_tile_stride(nc);
  const int64_t f0 = tile * frames;
  const int total = frames * nc;
  if (f0 + frames <= B) {
    const float4* src = reinterpret_cast<const float4*>(x + f0 * nc);  // frames*nc*4 B tiles are 16-B aligned
    const int nvec = total >> 2;                                       // total is a multiple of 4
    // (frame, j) of element 4*v advance incrementally: +4*nthreads elements per iteration
    int e = tid * 4;
    int fr = e / nc;
    int j = e - fr * nc;
    const int step = 4 * nthreads;
    const int dfr = step / nc, dj = step - dfr * nc;
    // Batches of kBatch 16-byte loads are issued back to back before the first LDS write: this wave is usually
    // alone on its SIMD, and a load -> wait -> write loop paid one full memory round trip per 1 KiB of the tile
    // (17 of them for 22 atoms).  Indices are clamped instead of
    // predicated so the loads stay unconditional (counted waits).
    if (stride == nc) {   // the tile as it lies in memory
      float4* dst = reinterpret_cast<float4*>(lds);
      for (int v0 = tid; v0 < nvec; v0 += nthreads * kBatch) {
        float4 val[kBatch];
#pragma unroll
        for (int i = 0; i < kBatch; ++i) {
          const int v = v0 + nthreads * i;
          val[i] = src[v < nvec ? v : nvec - 1];
        }
#pragma unroll
        for (int i = 0; i < kBatch; ++i) {
          const int v = v0 + nthreads * i;
          if (v < nvec) dst[v] = val[i];
        }
      }
      return;
    }
    for (int v0 = tid; v0 < nvec; v0 += nthreads * kBatch) {
      float4 val[kBatch];
#pragma unroll
      for (int i = 0; i < kBatch; ++i) {
        const int v = v0 + nthreads * i;
        val[i] = src[v < nvec ? v : nvec - 1];
      }
#pragma unroll
      for (int i = 0; i < kBatch; ++i) {
        if (v0 + nthreads * i < nvec) {
          int f = fr, jj = j;
          lds[f * stride + jj] = val[i].x;
          if (++jj == nc) { jj = 0; ++f; }
          lds[f * stride + jj] = val[i].y;
          if (++jj == nc) { jj = 0; ++f; }
          lds[f * stride + jj] = val[i].z;
          if (++jj == nc) { jj = 0; ++f; }
          lds[f * stride + jj] = val[i].w;
        }
        fr += dfr;
        j += dj;
        if (j >= nc) { j -= nc; ++fr; }
      }
    }
  } else {
    for (int e = tid; e < total; e += nthreads) {
      int fr = e / nc;
      int j = e - fr * nc;
      int64_t src = f0 + fr;
      if (src > B - 1) src = B - 1;
      lds[fr * stride + j] = x[src * nc + j];
    }
  }
}

// ------------------------------------------------------------------------------------
// 3x3 helpers (per lane, registers)
// ------------------------------------------------------------------------------------
// Workgroup barrier for data exchanged through LDS only: waits for this wave's LDS traffic, not for its global
// stores (__syncthreads() also drains vmcnt - with stores of a few KB per wave in flight that is 2-3 k cycles of a
// kernel whose time is one wave's chain).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// two floats in a register pair: v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 do two fp32 operations per instruction
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 splat2(float v) { return f2{v, v}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

struct V3 {
  float x, y, z;
};
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// row-major 3x3 in 9 floats:  (v R)_j = sum_i v_i R[3i+j]   (row vector times matrix)
__device__ __forceinline__ V3 row_times(V3 v, const float* R) {
  return V3{v.x * R[0] + v.y * R[3] + v.z * R[6], v.x * R[1] + v.y * R[4] + v.z * R[7],
            v.x * R[2] + v.y * R[5] + v.z * R[8]};
}
// (R g)_i = sum_j R[3i+j] g_j
__device__ __forceinline__ V3 mat_times(const float* R, V3 g) {
  return V3{R[0] * g.x + R[1] * g.y + R[2] * g.z, R[3] * g.x + R[4] * g.y + R[5] * g.z,
            R[6] * g.x + R[7] * g.y + R[8] * g.z};
}
// symmetric 3x3 from 6 floats (00 01 02 11 12 22) times vector
__device__ __forceinline__ V3 sym_times(const float* K, V3 t) {
  return V3{K[0] * t.x + K[1] * t.y + K[2] * t.z, K[1] * t.x + K[3] * t.y + K[4] * t.z,
            K[2] * t.x + K[4] * t.y + K[5] * t.z};
}
