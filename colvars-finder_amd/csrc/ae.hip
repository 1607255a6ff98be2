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

// ------------------------------------------------------------------------------------------------------------------
// The training step on the matrix cores.  Block = one 64-frame tile, four waves; wave v owns the frames 16 v .. 16 v + 15
// (the N = 16 columns of v_mfma_f32_16x16x4_f32).  The activations a_1 .. a_{L-1} stay in LDS as [row][frame] images;
//   forward    z = W a  :  A = W (LDS copy of theta), B = the previous image's 16 frame columns  -> wave-local, no barrier
//   error      zbar_L = 2 w (out - f) / sum w  (f re-read from global in the accumulator layout)
//   per layer, backwards, one barrier each:
//     W_l gradient = zbar_{l+1} (x) [a_l ; 1] over the 64 frames: 16x16 tiles dealt round-robin to the waves, both operands
//                    read as 16-byte LDS vectors (k-slot order 16 j + 4 kq + c on both), written straight to the block's
//                    slab row (each tile has exactly one owner: no atomics, fixed summation order);
//                    the first layer's B operand (the features) comes from global memory in the same k order
//     zbar_l     = (W^T zbar_{l+1}) .* act'(a_l):  A = W^T from LDS, B = zbar_{l+1}'s 16 frame columns -> wave-local
// The first, lane-per-frame version of this step (one wave per tile, scalar weight loads, 256 blocks) took 290 us at
// B = 20 000 with [66,20,20,20,2]/[2,10,10,66]; see DESIGN.md for this one.
// ------------------------------------------------------------------------------------------------------------------
constexpr int AP = 68;          // image pitch: rows are 16-byte aligned
constexpr int kAeMaxBlocks = 2048;

struct AeMLayout {
  int img_off[CVF_MAX_LAYERS + 1];   // dword offset of image a_l, l = 1..L-1 (rows d_l + 1, the last one all ones)
  int zb_off, ab_off, w_off, total;
  int zb_rows, ab_rows;
};
__host__ __device__ inline AeMLayout ae_mlayout(const cvf_mlp_desc& m) {
  AeMLayout lay;
  int rows = 0, dh = 1, dall = 1;
  for (int l = 1; l < m.n_layers; ++l) {
    lay.img_off[l] = rows * AP;
    rows += m.dims[l] + 1;
    dh = m.dims[l] > dh ? m.dims[l] : dh;
  }
  rows += 16;   // an operand tile may read up to 15 rows past the last image: keep them inside the zeroed area
  for (int l = 1; l <= m.n_layers; ++l) dall = m.dims[l] > dall ? m.dims[l] : dall;
  lay.zb_rows = up16(dall) + 16;
  lay.ab_rows = up16(dh) + 16;
  lay.zb_off = rows * AP;
  lay.ab_off = lay.zb_off + lay.zb_rows * AP;
  lay.w_off = lay.ab_off + lay.ab_rows * AP;
  lay.total = lay.w_off + ((m.n_params + 3) & ~3);
  return lay;
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__global__ __launch_bounds__(256) void ae_mfma_kernel(cvf_mlp_desc mlp, const float* __restrict__ theta,
                                                       const float* __restrict__ feat_rows, const int64_t* __restrict__ idx,
                                                       int64_t B, const float* __restrict__ w, double inv_wsum, int with_grad,
                                                       float* __restrict__ slab, double* __restrict__ partial,
                                                       int32_t* __restrict__ step) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  CVF_STAMP(18);
  const int row16 = lane & 15, kq = lane >> 4;
  const int fcol = 16 * wv + row16;          // this lane's frame column in forward / backward-data MFMAs
  // The layer table is indexed with a run-time l: read from the by-value kernel argument that turns into a private
  // (scratch-memory) copy and a ~700-cycle round trip per access; a copy in LDS costs an LDS read.
  __shared__ int s_dims[CVF_MAX_LAYERS + 1], s_woff[CVF_MAX_LAYERS], s_boff[CVF_MAX_LAYERS], s_act[CVF_MAX_LAYERS];
  __shared__ int s_img[CVF_MAX_LAYERS + 1];
  const AeMLayout lay = ae_mlayout(mlp);
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i <= CVF_MAX_LAYERS; ++i) {
      s_dims[i] = mlp.dims[i];
      s_img[i] = lay.img_off[i];
    }
#pragma unroll
    for (int i = 0; i < CVF_MAX_LAYERS; ++i) {
      s_woff[i] = mlp.w_off[0][i];
      s_boff[i] = mlp.b_off[0][i];
      s_act[i] = mlp.act[i];
    }
  }
  __syncthreads();
  const int L = mlp.n_layers;
  const int d0 = s_dims[0], dL = s_dims[L];
  float* ZB = lds + lay.zb_off;
  float* AB = lds + lay.ab_off;
  float* WL = lds + lay.w_off;
  // zero everything (operand tiles read rows past an image: they must be finite), ones rows, weights
  {
    float4* l4 = reinterpret_cast<float4*>(lds);
    const float4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int i = tid; i < lay.w_off / 4; i += 256) l4[i] = z4;   // (every region is a multiple of AP = 68 dwords)
#pragma unroll 4
    for (int i = tid; i < mlp.n_params; i += 256) WL[i] = theta[i];
  }
  __syncthreads();
  for (int l = 1; l < L; ++l)
    if (tid < 64) lds[s_img[l] + s_dims[l] * AP + tid] = 1.0f;
  __syncthreads();
  CVF_STAMP(19);
  const int64_t T = (B + CVF_TILE - 1) / CVF_TILE;
  double loss_acc = 0.0, w_acc = 0.0;
  float* out_row = slab + (int64_t)blockIdx.x * mlp.n_params;
  for (int64_t tile = blockIdx.x; tile < T; tile += gridDim.x) {
    const bool first = tile == (int64_t)blockIdx.x;
    // ---- this lane's frame (forward layout) and the tile's 64 frames (outer-product layout)
    const int64_t b = tile * CVF_TILE + fcol;
    const bool valid = b < B;
    const int64_t bb = valid ? b : B - 1;
    const int64_t frame = idx ? idx[bb] : bb;
    const float wb = valid ? w[bb] : 0.0f;
    const float* __restrict__ frow = feat_rows + frame * d0;
    CVF_STAMP(20);
    // ---- forward
    for (int l = 0; l < L; ++l) {
      const int din = s_dims[l], dout = s_dims[l + 1];
      const float* Wl = WL + s_woff[l];
      const float* bl = WL + s_boff[l];
      const float* in = l > 0 ? lds + s_img[l] : nullptr;
      float* dst = l + 1 < L ? lds + s_img[l + 1] : ZB;
      const bool act = s_act[l] != 0;
      auto finish = [&](int rt, const f32x4& acc) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = 16 * rt + 4 * kq + r;
          if (o < dout) {
            const float v = acc[r] + bl[o];
            dst[o * AP + fcol] = act ? cvf_tanh(v) : v;
          }
        }
      };
      // B operand = this lane's frame column of the input (global feature row for the first layer, LDS image after
      // that).  CH k-steps' worth of B values are requested together and shared by up to kRT row tiles, whose A values
      // (weights, LDS) are requested together as well: one wait per batch instead of one per matrix instruction.
      constexpr int kRT = 8;
      auto layer = [&](auto ch_) {
        constexpr int CH = decltype(ch_)::value;
        for (int rt0 = 0; 16 * rt0 < dout; rt0 += kRT) {
          f32x4 acc[kRT];
#pragma unroll
          for (int t = 0; t < kRT; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
          for (int s0 = 0; 4 * s0 < din; s0 += CH) {
            float bv[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
              const int i = 4 * (s0 + u) + kq;
              bv[u] = l == 0 ? frow[i < din ? i : din - 1] : in[i * AP + fcol];   // (rows past the image: finite, times a = 0)
            }
#pragma unroll
            for (int t = 0; t < kRT; ++t) {
              const int rt = rt0 + t;
              if (16 * rt < dout) {   // wave-uniform
                const int oa = 16 * rt + row16;
                const float* wr = Wl + (oa < dout ? oa : dout - 1) * din;
                float av[CH];
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                  const int i = 4 * (s0 + u) + kq;
                  av[u] = wr[i < din ? i : din - 1];
                }
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                  const int i = 4 * (s0 + u) + kq;
                  acc[t] = mfma4((oa < dout && i < din) ? av[u] : 0.0f, bv[u], acc[t]);
                }
              }
            }
          }
#pragma unroll
          for (int t = 0; t < kRT; ++t)
            if (16 * (rt0 + t) < dout) finish(rt0 + t, acc[t]);
        }
      };
      if (l == 0) layer(std::integral_constant<int, 20>{});   // 80 inputs per global round trip
      else layer(std::integral_constant<int, 8>{});
      CVF_STAMP(41 + l);
    }
    CVF_STAMP(21);
    // ---- weighted squared error and zbar_L = 2 w (out - f) / sum(w)     (core.py:666); rows o = 16 rt + 4 kq + r
    {
      const float scale = (float)(2.0 * (double)wb * inv_wsum);
      const bool act_last = s_act[L - 1] != 0;
      float err2 = 0.0f;
      for (int rt0 = 0; 16 * rt0 < dL; rt0 += 4) {
        float fv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int o = 16 * (rt0 + (u >> 2)) + 4 * kq + (u & 3);
          fv[u] = frow[o < dL ? o : dL - 1];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int o = 16 * (rt0 + (u >> 2)) + 4 * kq + (u & 3);
          if (o < dL) {
            const float out = ZB[o * AP + fcol];
            const float df = out - fv[u];
            err2 = fmaf(df, df, err2);
            float zb = scale * df;
            if (act_last) zb *= 1.0f - out * out;
            ZB[o * AP + fcol] = zb;
          }
        }
      }
      loss_acc += (double)wb * (double)err2;
      if (kq == 0) w_acc += (double)wb;
    }
    CVF_STAMP(22);
    if (!with_grad) continue;
    // ---- backward
    float* Zc = ZB;
    float* Zn = AB;
    for (int l = L - 1; l >= 0; --l) {
      const int din = s_dims[l], dout = s_dims[l + 1];
      const int wo = s_woff[l], bo = s_boff[l];
      __syncthreads();   // zbar_{l+1}: all 64 frame columns are in place
      CVF_STAMP(23 + 2 * (L - 1 - l));
      // (a) weight gradient tiles, round-robin over the waves
      const int nct = (din + 1 + 15) / 16, nrt = (dout + 15) / 16;
      int64_t foff[16];   // first layer: row offsets of this lane's sixteen k-slot frames (see below)
      if (l == 0) {
#pragma unroll
        for (int jc = 0; jc < 16; ++jc) {
          const int64_t fb = tile * CVF_TILE + 16 * (jc >> 2) + 4 * kq + (jc & 3);
          const int64_t fbc = fb < B ? fb : B - 1;
          foff[jc] = (idx ? idx[fbc] : fbc) * d0;
        }
      }
      for (int pr = wv; pr < nrt * nct; pr += 4) {
        const int rt = pr / nct, ct = pr - rt * nct;
        const int i = 16 * ct + row16;
        const float4* za = reinterpret_cast<const float4*>(Zc + (16 * rt + row16) * AP + 4 * kq);
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        if (l > 0) {
          const float4* ib = reinterpret_cast<const float4*>(lds + s_img[l] + i * AP + 4 * kq);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float4 a = za[4 * j], bq = ib[4 * j];
            acc = mfma4(a.x, bq.x, acc);
            acc = mfma4(a.y, bq.y, acc);
            acc = mfma4(a.z, bq.z, acc);
            acc = mfma4(a.w, bq.w, acc);
          }
        } else {
          // B = [f ; 1] from global memory: k-slot kq of k-step (j, c) is frame 16 j + 4 kq + c of the tile
          const int ic = i < d0 ? i : d0 - 1;
          const float pad = i == d0 ? 1.0f : 0.0f;
          float bvals[16];
#pragma unroll
          for (int jc = 0; jc < 16; ++jc) bvals[jc] = feat_rows[foff[jc] + ic];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float4 a = za[4 * j];
            acc = mfma4(a.x, i < d0 ? bvals[4 * j + 0] : pad, acc);
            acc = mfma4(a.y, i < d0 ? bvals[4 * j + 1] : pad, acc);
            acc = mfma4(a.z, i < d0 ? bvals[4 * j + 2] : pad, acc);
            acc = mfma4(a.w, i < d0 ? bvals[4 * j + 3] : pad, acc);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = 16 * rt + 4 * kq + r;
          if (o < dout && i <= din) {
            float* dstp = out_row + (i < din ? wo + o * din + i : bo + o);
            *dstp = first ? acc[r] : *dstp + acc[r];
          }
        }
      }
      CVF_STAMP(24 + 2 * (L - 1 - l));
      // (b) zbar_l for this wave's frames
      if (l > 0) {
        const float* Wl = WL + wo;
        const float* al = lds + s_img[l];
        const bool act = s_act[l - 1] != 0;
        constexpr int kBT = 4, kBC = 8;   // row tiles per pass, k-steps per batch (as in the forward layers)
        for (int rt0 = 0; 16 * rt0 < din; rt0 += kBT) {
          f32x4 acc[kBT];
#pragma unroll
          for (int t = 0; t < kBT; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
          for (int s0 = 0; 4 * s0 < dout; s0 += kBC) {
            float bv[kBC];
#pragma unroll
            for (int u = 0; u < kBC; ++u) bv[u] = Zc[(4 * (s0 + u) + kq) * AP + fcol];
#pragma unroll
            for (int t = 0; t < kBT; ++t) {
              const int rt = rt0 + t;
              if (16 * rt < din) {   // wave-uniform
                const int ia = 16 * rt + row16;
                const int iac = ia < din ? ia : din - 1;
                float av[kBC];
#pragma unroll
                for (int u = 0; u < kBC; ++u) {
                  const int o = 4 * (s0 + u) + kq;
                  av[u] = Wl[(o < dout ? o : dout - 1) * din + iac];
                }
#pragma unroll
                for (int u = 0; u < kBC; ++u) {
                  const int o = 4 * (s0 + u) + kq;
                  acc[t] = mfma4((ia < din && o < dout) ? av[u] : 0.0f, bv[u], acc[t]);
                }
              }
            }
          }
#pragma unroll
          for (int t = 0; t < kBT; ++t) {
            const int rt = rt0 + t;
            if (16 * rt < din) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int i = 16 * rt + 4 * kq + r;
                if (i < din) {
                  float v = acc[t][r];
                  if (act) {
                    const float a = al[i * AP + fcol];
                    v *= 1.0f - a * a;
                  }
                  Zn[i * AP + fcol] = v;
                }
              }
            }
          }
        }
        float* t = Zc;
        Zc = Zn;
        Zn = t;
      }
    }
    CVF_STAMP(40);
    __syncthreads();   // the next tile's forward overwrites the images / ZB
  }
  // ---- per-block loss partials: fixed-order reduction over the block's four waves
  __shared__ double red[4][2];
  const double ls = wave_sum(loss_acc), wsum = wave_sum(w_acc);
  if (lane == 0) {
    red[wv][0] = ls;
    red[wv][1] = wsum;
  }
  __syncthreads();
  if (tid == 0) {
    partial[2 * blockIdx.x] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
    partial[2 * blockIdx.x + 1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    if (with_grad && step != nullptr && blockIdx.x == 0) *step += 1;  // one gradient per optimiser step
  }
}

// out2 = fixed-order sums of the per-block {sum w*err, sum w}: lane l adds rows l, l+64, ..., then the DPP reduction
__global__ __launch_bounds__(64) void ae_loss_sum_kernel(const double* __restrict__ partial, int nblocks, double* __restrict__ out2) {
  const int lane = threadIdx.x;
  double a0 = 0.0, a1 = 0.0;
  for (int g = lane; g < nblocks; g += 64) {
    a0 += partial[2 * g];
    a1 += partial[2 * g + 1];
  }
  a0 = wave_sum(a0);
  a1 = wave_sum(a1);
  if (lane == 0) {
    out2[0] = a0;
    out2[1] = a1;
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
  return (int)(T < kAeMaxBlocks ? T : kAeMaxBlocks);
}

}  // namespace

extern "C" int64_t cvf_ae_scratch_floats(const cvf_mlp_desc* mlp, int64_t B) {
  // slab rows + (2 doubles per block) expressed in floats
  const int64_t G = ae_grid(B);
  return G * mlp->n_params + 4 * G + 4;
}

extern "C" int cvf_ae_step(const cvf_mlp_desc* mlp, const float* theta, const float* feat_rows, const int64_t* idx,
                           int64_t B, const float* w, double inv_wsum, float* scratch, double* out2, float* grad,
                           int32_t* step_count, const cvf_adam_args* adam, void* stream) {
  CVF_REQUIRE(mlp && theta && feat_rows && w && scratch && out2 && B > 0, "cvf_ae_step: bad argument");
  CVF_REQUIRE(mlp->n_nets == 1 && mlp->n_layers >= 1 && mlp->n_layers <= CVF_MAX_LAYERS, "cvf_ae_step: one chain expected");
  CVF_REQUIRE(mlp->dims[0] == mlp->dims[mlp->n_layers], "cvf_ae_step: output width %d != input width %d",
              mlp->dims[mlp->n_layers], mlp->dims[0]);
  CVF_REQUIRE(adam == nullptr || (grad && adam->theta && adam->m && adam->v && adam->step_count),
              "cvf_ae_step: incomplete adam arguments");
  const AeMLayout lay = ae_mlayout(*mlp);
  const size_t lds = (size_t)lay.total * sizeof(float);
  CVF_REQUIRE(lds <= 160 * 1024, "cvf_ae_step: the chain needs %zu B of LDS per workgroup (> 160 KiB)", lds);
  const int G = ae_grid(B);
  const int Pn = mlp->n_params;
  // scratch: [slab floats][partials as doubles, 8-byte aligned]
  float* slab = scratch;
  double* partial = reinterpret_cast<double*>(scratch + (((int64_t)G * Pn + 1) & ~(int64_t)1));
  hipStream_t s = (hipStream_t)stream;
  if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)ae_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(ae_mfma_kernel, dim3(G), dim3(256), lds, s, *mlp, theta, feat_rows, idx, B, w, inv_wsum, grad ? 1 : 0, slab,
                     partial, grad ? step_count : nullptr);
  int rc = cvf_check_launch("ae_mfma_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(ae_loss_sum_kernel, dim3(1), dim3(64), 0, s, partial, G, out2);
  rc = cvf_check_launch("ae_loss_sum_kernel");
  if (rc || grad == nullptr) return rc;
  // fixed-order sum of the slab rows (+ the Adam update when asked for): the kernel the eigenfunction path uses
  cvf_adam_args ad;
  if (adam != nullptr) {
    ad = *adam;
    ad.mlp = nullptr;      // an AutoEncoder has no MFMA fragment copy to refresh
    ad.packed = nullptr;
  }
  return cvf_slab_reduce(slab, G, Pn, grad, adam != nullptr ? &ad : nullptr, stream);
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
