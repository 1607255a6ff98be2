// AutoEncoderTask step (core.py:652-666,699-712) and generic net inference, for arbitrary layer
// widths.  One lane = one frame; every activation vector of the chain lives in wave-private LDS
// as a [row][frame] image (66-dword pitch) so that it serves both the per-lane matrix-vector
// products (weights are wave-uniform -> scalar loads) and, unchanged, as an MFMA operand of the
// weight-gradient contraction  W_l += sum_frames zbar_l (x) [a_{l-1}; 1]  (K = frames).
// Each block accumulates its share of the gradient in an LDS image of the flat parameter
// buffer and writes it to its slab row once; slab rows are then summed in fixed order.
#include "cvf_adam.hpp"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int P = 66;  // LDS row pitch in dwords
constexpr int kAeBlocks = 256;

__host__ __device__ inline int up16(int n) { return (n + 15) & ~15; }

struct AeLayout {
  int act_off[CVF_MAX_LAYERS + 1];  // dword offsets of the activation images a_0..a_{L-1}
  int zb_off, ab_off, gi_off, total;
};

__host__ __device__ inline AeLayout ae_layout(const cvf_mlp_desc& m, bool with_grad) {
  AeLayout lay;
  int pos = 0, dmax = 0;
  for (int l = 0; l < m.n_layers; ++l) {
    lay.act_off[l] = pos;
    pos += up16(m.dims[l] + 1) * P;
  }
  for (int l = 0; l <= m.n_layers; ++l) dmax = m.dims[l] > dmax ? m.dims[l] : dmax;
  lay.zb_off = pos;
  pos += up16(dmax) * P;
  lay.ab_off = pos;
  pos += up16(dmax) * P;
  lay.gi_off = pos;
  if (with_grad) pos += m.n_params;
  lay.total = pos;
  return lay;
}

// out[o] = act(b[o] + sum_i W[o][i] in[i]) for one lane, 8 outputs at a time
__device__ __forceinline__ void dense_fwd(const float* __restrict__ W, const float* __restrict__ b, int din, int dout,
                                          const float* in, float* out, bool act, int lane) {
  for (int o0 = 0; o0 < dout; o0 += 8) {
    float acc[8];
    int ro[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int oj = o0 + j < dout ? o0 + j : dout - 1;
      ro[j] = oj * din;
      acc[j] = b[oj];
    }
    for (int i = 0; i < din; ++i) {
      const float a = in[i * P + lane];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = fmaf(W[ro[j] + i], a, acc[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (o0 + j < dout) out[(o0 + j) * P + lane] = act ? cvf_tanh(acc[j]) : acc[j];
  }
}

// out[i] = sum_o W[o][i] z[o]
__device__ __forceinline__ void dense_bwd_data(const float* __restrict__ W, int din, int dout, const float* z, float* out,
                                               int lane) {
  for (int i0 = 0; i0 < din; i0 += 8) {
    float acc[8];
    int ci[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      ci[j] = i0 + j < din ? i0 + j : din - 1;
      acc[j] = 0.0f;
    }
    for (int o = 0; o < dout; ++o) {
      const float zo = z[o * P + lane];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = fmaf(W[o * din + ci[j]], zo, acc[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (i0 + j < din) out[(i0 + j) * P + lane] = acc[j];
  }
}

__global__ __launch_bounds__(64) void ae_step_kernel(cvf_mlp_desc mlp, const float* __restrict__ theta,
                                                      const float* __restrict__ feat_rows,
                                                      const int64_t* __restrict__ idx, int64_t B,
                                                      const float* __restrict__ w, double inv_wsum, int with_grad,
                                                      float* __restrict__ slab, double* __restrict__ partial,
                                                      int32_t* __restrict__ step) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x;
  const int L = mlp.n_layers;
  const int d0 = mlp.dims[0], dL = mlp.dims[L];
  const AeLayout lay = ae_layout(mlp, with_grad != 0);
  float* ZB = lds + lay.zb_off;
  float* AB = lds + lay.ab_off;
  float* GI = lds + lay.gi_off;
  const int row16 = lane & 15, kq = lane >> 4;
  // ones rows (bias columns) and a clean gradient image
  for (int l = 0; l < L; ++l) {
    float* a = lds + lay.act_off[l];
    const int rows = up16(mlp.dims[l] + 1);
    for (int r = mlp.dims[l]; r < rows; ++r) a[r * P + lane] = r == mlp.dims[l] ? 1.0f : 0.0f;
  }
  if (with_grad)
    for (int p = lane; p < mlp.n_params; p += 64) GI[p] = 0.0f;
  __syncthreads();
  const int64_t T = (B + CVF_TILE - 1) / CVF_TILE;
  double loss_acc = 0.0, w_acc = 0.0;
  for (int64_t tile = blockIdx.x; tile < T; tile += gridDim.x) {
    const int64_t b = tile * CVF_TILE + lane;
    const bool valid = b < B;
    const int64_t bb = valid ? b : B - 1;
    const int64_t frame = idx ? idx[bb] : bb;
    const float wb = valid ? w[bb] : 0.0f;
    const float* __restrict__ frow = feat_rows + frame * d0;
    float* a0 = lds + lay.act_off[0];
    for (int j = 0; j < d0; ++j) a0[j * P + lane] = frow[j];
    // forward
    for (int l = 0; l < L - 1; ++l)
      dense_fwd(theta + mlp.w_off[0][l], theta + mlp.b_off[0][l], mlp.dims[l], mlp.dims[l + 1], lds + lay.act_off[l],
                lds + lay.act_off[l + 1], mlp.act[l] != 0, lane);
    dense_fwd(theta + mlp.w_off[0][L - 1], theta + mlp.b_off[0][L - 1], mlp.dims[L - 1], dL, lds + lay.act_off[L - 1], ZB,
              mlp.act[L - 1] != 0, lane);
    // weighted squared error and zbar_L = 2 w (out - f) / sum(w)     (core.py:666)
    float err2 = 0.0f;
    const float scale = (float)(2.0 * (double)wb * inv_wsum);
    for (int j = 0; j < dL; ++j) {
      const float out = ZB[j * P + lane];
      const float df = out - a0[j * P + lane];
      err2 = fmaf(df, df, err2);
      float zb = scale * df;
      if (mlp.act[L - 1]) zb *= 1.0f - out * out;
      ZB[j * P + lane] = zb;
    }
    loss_acc += (double)wb * (double)err2;
    w_acc += (double)wb;
    if (!with_grad) continue;
    // backward
    for (int l = L - 1; l >= 0; --l) {
      const int din = mlp.dims[l], dout = mlp.dims[l + 1];
      const float* Ain = lds + lay.act_off[l];
      const int wo = mlp.w_off[0][l], bo = mlp.b_off[0][l];
      __syncthreads();
      for (int rt = 0; rt * 16 < dout; ++rt)
        for (int ct = 0; ct * 16 < din + 1; ++ct) {
          f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 4
          for (int s = 0; s < 16; ++s)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ZB[(16 * rt + row16) * P + 4 * s + kq],
                                                       Ain[(16 * ct + row16) * P + 4 * s + kq], acc, 0, 0, 0);
          const int i = 16 * ct + row16;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int o = 16 * rt + 4 * kq + r;
            if (o < dout) {
              if (i < din) GI[wo + o * din + i] += acc[r];
              else if (i == din) GI[bo + o] += acc[r];
            }
          }
        }
      __syncthreads();
      if (l > 0) {
        dense_bwd_data(theta + wo, din, dout, ZB, AB, lane);
        for (int i = 0; i < din; ++i) {
          float v = AB[i * P + lane];
          if (mlp.act[l - 1]) {
            const float a = Ain[i * P + lane];
            v *= 1.0f - a * a;
          }
          ZB[i * P + lane] = v;
        }
      }
    }
  }
  __syncthreads();
  const double ls = wave_sum(loss_acc), wsum = wave_sum(w_acc);
  if (lane == 0) {
    partial[2 * blockIdx.x] = ls;
    partial[2 * blockIdx.x + 1] = wsum;
  }
  if (with_grad) {
    float* out = slab + (int64_t)blockIdx.x * mlp.n_params;
    for (int p = lane; p < mlp.n_params; p += 64) out[p] = GI[p];
    if (step != nullptr && blockIdx.x == 0 && lane == 0) *step += 1;  // one gradient per optimiser step
  }
}

__global__ void ae_reduce_kernel(const float* __restrict__ slab, const double* __restrict__ partial, int nblocks, int Pn,
                                 float* __restrict__ grad, double* __restrict__ out2, int use_adam, AdamDev adam) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (grad && p < Pn) {
    float acc = 0.0f;
    for (int g = 0; g < nblocks; ++g) acc += slab[(int64_t)g * Pn + p];
    grad[p] = acc;
    if (use_adam) adam_apply(adam, adam_scalars(adam), cvf_mlp_desc{}, p, acc);
  }
  if (p < 2) {
    double acc = 0.0;
    for (int g = 0; g < nblocks; ++g) acc += partial[2 * g + p];
    out2[p] = acc;
  }
}

// inference on row-major features: every net of the model, optionally stopping after `upto` layers
__global__ __launch_bounds__(64) void mlp_eval_rows_kernel(cvf_mlp_desc mlp, const float* __restrict__ theta,
                                                            const float* __restrict__ feat_rows, int64_t B, int upto,
                                                            float* __restrict__ out) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x;
  const int64_t b = (int64_t)blockIdx.x * CVF_TILE + lane;
  const bool valid = b < B;
  const int64_t bb = valid ? b : B - 1;
  const AeLayout lay = ae_layout(mlp, false);
  const int d0 = mlp.dims[0], dU = mlp.dims[upto];
  float* a0 = lds + lay.act_off[0];
  for (int j = 0; j < d0; ++j) a0[j * P + lane] = feat_rows[bb * d0 + j];
  float* ZB = lds + lay.zb_off;
  for (int net = 0; net < mlp.n_nets; ++net) {
    for (int l = 0; l < upto; ++l)
      dense_fwd(theta + mlp.w_off[net][l], theta + mlp.b_off[net][l], mlp.dims[l], mlp.dims[l + 1], lds + lay.act_off[l],
                l + 1 < upto ? lds + lay.act_off[l + 1] : ZB, mlp.act[l] != 0, lane);
    if (valid)
      for (int j = 0; j < dU; ++j) out[bb * (int64_t)(mlp.n_nets * dU) + net * dU + j] = ZB[j * P + lane];
  }
}

int ae_grid(int64_t B) {
  const int64_t T = cvf_ntiles(B);
  return (int)(T < kAeBlocks ? T : kAeBlocks);
}

}  // namespace

extern "C" int64_t cvf_ae_scratch_floats(const cvf_mlp_desc* mlp, int64_t B) {
  // slab rows + (2 doubles per block) expressed in floats
  return (int64_t)kAeBlocks * mlp->n_params + 4 * (int64_t)kAeBlocks + 4;
}

extern "C" int cvf_ae_step(const cvf_mlp_desc* mlp, const float* theta, const float* feat_rows, const int64_t* idx,
                           int64_t B, const float* w, double inv_wsum, float* scratch, double* out2, float* grad,
                           int32_t* step_count, const cvf_adam_args* adam, void* stream) {
  CVF_REQUIRE(mlp && theta && feat_rows && w && scratch && out2 && B > 0, "cvf_ae_step: bad argument");
  CVF_REQUIRE(mlp->n_nets == 1 && mlp->n_layers >= 1 && mlp->n_layers <= CVF_MAX_LAYERS, "cvf_ae_step: one chain expected");
  CVF_REQUIRE(mlp->dims[0] == mlp->dims[mlp->n_layers], "cvf_ae_step: output width %d != input width %d",
              mlp->dims[mlp->n_layers], mlp->dims[0]);
  const AeLayout lay = ae_layout(*mlp, grad != nullptr);
  const size_t lds = (size_t)lay.total * sizeof(float);
  CVF_REQUIRE(lds <= 160 * 1024, "cvf_ae_step: the chain needs %zu B of LDS per wave (> 160 KiB)", lds);
  const int G = ae_grid(B);
  // scratch: [slab floats][partials as doubles, 8-byte aligned]
  float* slab = scratch;
  double* partial = reinterpret_cast<double*>(scratch + (((int64_t)kAeBlocks * mlp->n_params + 1) & ~(int64_t)1));
  hipStream_t s = (hipStream_t)stream;
  (void)hipFuncSetAttribute((const void*)ae_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  AdamDev ad{};
  if (adam != nullptr) {
    CVF_REQUIRE(grad && adam->theta && adam->m && adam->v && adam->step_count, "cvf_ae_step: incomplete adam arguments");
    ad = AdamDev{adam->theta, adam->m, adam->v, (float)adam->lr, (float)adam->beta1, (float)adam->beta2, (float)adam->eps,
                 adam->step_count, nullptr};
  }
  hipLaunchKernelGGL(ae_step_kernel, dim3(G), dim3(64), lds, s, *mlp, theta, feat_rows, idx, B, w, inv_wsum,
                     grad ? 1 : 0, slab, partial, grad ? step_count : nullptr);
  int rc = cvf_check_launch("ae_step_kernel");
  if (rc) return rc;
  const int Pn = mlp->n_params;
  hipLaunchKernelGGL(ae_reduce_kernel, dim3((Pn + 255) / 256), dim3(256), 0, s, slab, partial, G, Pn, grad, out2, adam != nullptr ? 1 : 0, ad);
  return cvf_check_launch("ae_reduce_kernel");
}

extern "C" int cvf_mlp_eval_rows(const cvf_mlp_desc* mlp, const float* theta, const float* feat_rows, int64_t B,
                                 int upto_layer, float* out, void* stream) {
  CVF_REQUIRE(mlp && theta && feat_rows && out && B > 0, "cvf_mlp_eval_rows: bad argument");
  CVF_REQUIRE(upto_layer >= 1 && upto_layer <= mlp->n_layers, "cvf_mlp_eval_rows: upto_layer=%d out of range", upto_layer);
  const AeLayout lay = ae_layout(*mlp, false);
  const size_t lds = (size_t)lay.total * sizeof(float);
  CVF_REQUIRE(lds <= 160 * 1024, "cvf_mlp_eval_rows: the chain needs %zu B of LDS per wave (> 160 KiB)", lds);
  (void)hipFuncSetAttribute((const void*)mlp_eval_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(mlp_eval_rows_kernel, dim3((unsigned)cvf_ntiles(B)), dim3(64), lds, (hipStream_t)stream, *mlp, theta,
                     feat_rows, B, upto_layer, out);
  return cvf_check_launch("mlp_eval_rows_kernel");
}
