// AutoEncoderTask step (core.py:652-666,699-712) and generic net inference, for arbitrary layer
// widths.  One lane = one frame; every activation vector of the chain lives in wave-private LDS
// as a [row][frame] image (66-dword pitch) so that it serves both the per-lane matrix-vector
// products (weights are wave-uniform -> scalar loads) and, unchanged, as an MFMA operand of the
// weight-gradient contraction  W_l += sum_frames zbar_l (x) [a_{l-1}; 1]  (K = frames).
// Each block accumulates its share of the gradient in an LDS image of the flat parameter
// buffer and writes it to its slab row once; slab rows are then summed in fixed order.
#include "cvf_adam.hpp"
#include <stddef.h>
#include <stdlib.h>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// (csrc/ef_mfma.hip) fixed-order sum of slab rows [+ mask] [+ Adam] [+ one extra block adding n_pair [a, b] rows -> [a, b, a / b]]
int cvf_slab_reduce_impl(const float* slab, int64_t n_rows, int64_t n_params, float* grad, const float* mask,
                         const cvf_adam_args* adam, void* stream, const double* pair_partial = nullptr, int n_pair = 0,
                         double* pair_out = nullptr);

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

// TANH: every activation of the chain is Tanh or none (the reference's notebooks): the training kernel is instantiated once
// with tanh inlined and once with the run-time switch of cvf_common.hpp (the switch in the inner loops cost the tanh chains
// 8 % of the autoencoder step and 18 % of the regularised one)
template <bool TANH>
__device__ __forceinline__ float act_f(int kind, float z) {
  if constexpr (TANH) return kind ? cvf_tanh(z) : z;
  else return cvf_act(kind, z);
}
template <bool TANH>
__device__ __forceinline__ float act_d(int kind, float h) {
  if constexpr (TANH) return 1.0f - h * h;   // (callers test kind != 0 first)
  else return cvf_act_d1(kind, h);
}
bool chain_is_tanh(const cvf_mlp_desc* m) {
  for (int l = 0; l < m->n_layers; ++l)
    if (m->act[l] != CVF_ACT_NONE && m->act[l] != CVF_ACT_TANH) return false;
  return true;
}

// out[o] = act(b[o] + sum_i W[o][i] in[i]) for one lane, 8 outputs at a time
__device__ __forceinline__ void dense_fwd(const float* __restrict__ W, const float* __restrict__ b, int din, int dout,
                                          const float* in, float* out, int act, int lane) {
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
      if (o0 + j < dout) out[(o0 + j) * P + lane] = cvf_act(act, acc[j]);
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
  int zb_off, ab_off, w_off, tail_off, total;
  int zb_rows, ab_rows;
  int skip0;   // > 0: the first layer's weights (theta[0 .. skip0), read once per tile by the forward pass) stay in global memory
};
// Operand tiles read up to 15 rows past an image / past zbar: those rows belong to the next region (next image, zbar, the
// second zbar buffer, the weights), which always holds finite numbers (everything is zeroed once, then activations and
// weights), and they meet A values forced to 0 or land in output rows that are discarded - no padding rows needed.
// The forward-only pass (test loop, RegAutoEncoderTask's statistics pass) has no second zbar buffer.
__host__ __device__ inline AeMLayout ae_mlayout_of(const cvf_mlp_desc& m, bool with_grad, bool tight) {
  AeMLayout lay;
  int rows = 0, dh = 1, dall = 1;
  for (int l = 1; l < m.n_layers; ++l) {
    lay.img_off[l] = rows * AP;
    rows += m.dims[l] + 1;
    dh = m.dims[l] > dh ? m.dims[l] : dh;
  }
  for (int l = 1; l <= m.n_layers; ++l) dall = m.dims[l] > dall ? m.dims[l] : dall;
  // tight: zbar buffers of exactly the rows written (what is read past them lies in the next region, see above), and the
  // first layer's weights left in global memory
  lay.zb_rows = tight ? dall : up16(dall);
  lay.ab_rows = with_grad ? (tight ? dh : up16(dh)) : 0;
  lay.skip0 = tight && m.w_off[0][0] == 0 ? m.dims[0] * m.dims[1] : 0;
  lay.zb_off = rows * AP;
  lay.ab_off = lay.zb_off + lay.zb_rows * AP;
  lay.w_off = lay.ab_off + lay.ab_rows * AP;
  // batches of k-steps read up to 31 rows past zbar: when theta is shorter than that, a zeroed tail keeps them finite
  const int np4 = (m.n_params - lay.skip0 + 3) & ~3;
  lay.tail_off = lay.w_off + np4;
  lay.total = lay.tail_off + (np4 < 32 * AP ? 32 * AP - np4 : 0);
  return lay;
}
// The roomy layout unless only the tight one lets TWO workgroups share a CU's 160 KB (RegAutoEncoderTask's chain with gradient:
// 86.7 KB roomy - one workgroup per CU, 626 tiles in three rounds, 147 us - against 77.6 KB tight).
__host__ __device__ inline AeMLayout ae_mlayout(const cvf_mlp_desc& m, bool with_grad) {
  constexpr int kHalf = 80 * 1024 - 1024;   // (less the kernel's static tables)
  const AeMLayout roomy = ae_mlayout_of(m, with_grad, false);
  if (roomy.total * (int)sizeof(float) <= kHalf) return roomy;
  const AeMLayout tight = ae_mlayout_of(m, with_grad, true);
  return tight.total * (int)sizeof(float) <= kHalf ? tight : roomy;
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// RegAutoEncoderTask (core.py:746-1217) runs the same chain with K extra outputs: the encoder, then the decoder and the
// K regulariser nets side by side (block-structured layers built by the host), so the last layer yields
// [reconstruction (n_mse = d_0 rows) | y_1..y_K].  Tiles 0..T-1 take the rows idx (+ reconstruction error against the rows
// idx + lag_t: the time-lagged autoencoder), tiles T..2T-1 the rows idx + lag_in (the lagged arguments of the
// transfer-operator loss, no reconstruction error).  The heads' output gradients are those of EigenFunctionTask's transfer
// loss: coefficients from cvf_ef_loss, partner values from the forward pass's y_tiled.  K = 0: plain autoencoder.
struct AeReg {
  int K;                 // regulariser heads (last K outputs)
  int write_y;           // forward pass: store the heads' outputs in y_tiled
  int64_t T;             // tiles of the direct pass
  int64_t n_tiles;       // T or 2 T
  int64_t lag_t;         // reconstruction target row = idx + lag_t
  int64_t lag_in;        // input rows of the second pass = idx + lag_in
  double head_scale;     // gamma_0
  const double* coef;    // [gS1(K), gS2(K*K), gT(K), gS1'(K), gS2'_ii(K)]  (CVF_COEF_LEN)
  const float* w_lag;    // [B] weights of the lagged frames
  float* y_tiled;        // [n_tiles][K][64]
  // encoder regularisers (core.py:912-971: variance and covariance penalties on the latent vector, rows idx only)
  int enc_layer;         // image index of the latent vector = number of encoder layers
  int k_enc;             // its width
  float* enc_tiled;      // forward pass: [T][k_enc][64] latent values (NULL: not wanted)
  const double* enc_coef;   // backward pass: [gS1(k), gS2(k*k)] of eta_1 norm + eta_2 orth penalties (NULL: off)
  // activation hand-off between the statistics pass and the gradient pass of one step (cvf_regae_forward_keep /
  // cvf_regae_backward_reuse): every tile's images and last-layer outputs, [n_tiles][image rows + d_L][64]
  float* hand;           // NULL: none
  int hand_mode;         // 1: the forward pass stores them, 2: the gradient pass loads them instead of running the chain forward
};

template <bool TANH>
__global__ __launch_bounds__(256, 2) void ae_mfma_kernel(cvf_mlp_desc mlp, const float* __restrict__ theta,
                                                       const float* __restrict__ feat_rows, const int64_t* __restrict__ idx,
                                                       int64_t B, const float* __restrict__ w, double inv_wsum, int with_grad,
                                                       float* __restrict__ slab, double* __restrict__ partial,
                                                       int32_t* __restrict__ step, AeReg reg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  CVF_STAMP(18);
  const int row16 = lane & 15, kq = lane >> 4;
  const int fcol = 16 * wv + row16;          // this lane's frame column in forward / backward-data MFMAs
  // The layer table is indexed with a run-time l: read from the by-value kernel argument that turns into a private
  // (scratch-memory) copy and a ~700-cycle round trip per access; a copy in LDS costs an LDS read.
  __shared__ int s_dims[CVF_MAX_LAYERS + 1], s_woff[CVF_MAX_LAYERS], s_boff[CVF_MAX_LAYERS], s_act[CVF_MAX_LAYERS];
  __shared__ int s_img[CVF_MAX_LAYERS + 1];
  const AeMLayout lay = ae_mlayout(mlp, with_grad != 0);
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
  float* WL = lds + lay.w_off - lay.skip0;   // indexed like theta (entries below skip0 are not in LDS)
  // zero everything (operand tiles read rows past an image: they must be finite), ones rows, weights
  {
    float4* l4 = reinterpret_cast<float4*>(lds);
    const float4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int i = tid; i < lay.w_off / 4; i += 256) l4[i] = z4;   // (every region is a multiple of AP = 68 dwords)
    for (int i = lay.tail_off / 4 + tid; i < lay.total / 4; i += 256) l4[i] = z4;
#pragma unroll 4
    for (int i = lay.skip0 + tid; i < mlp.n_params; i += 256) WL[i] = theta[i];
  }
  __syncthreads();
  for (int l = 1; l < L; ++l)
    if (tid < 64) lds[s_img[l] + s_dims[l] * AP + tid] = 1.0f;
  __syncthreads();
  CVF_STAMP(19);
  const int n_mse = dL - reg.K;   // reconstruction rows of the last layer
  double loss_acc = 0.0, w_acc = 0.0;
  float* out_row = slab + (int64_t)blockIdx.x * mlp.n_params;
  for (int64_t tile = blockIdx.x; tile < reg.n_tiles; tile += gridDim.x) {
    const bool first = tile == (int64_t)blockIdx.x;
    const bool lagged = tile >= reg.T;            // second pass: rows idx + lag_in, heads only
    const int64_t t0 = lagged ? tile - reg.T : tile;
    const int64_t in_shift = lagged ? reg.lag_in : 0;
    // ---- this lane's frame (forward layout) and the tile's 64 frames (outer-product layout)
    const int64_t b = t0 * CVF_TILE + fcol;
    const bool valid = b < B;
    const int64_t bb = valid ? b : B - 1;
    const int64_t frame = idx ? idx[bb] : bb;
    const float wraw = valid ? w[bb] : 0.0f;
    const float wb = lagged ? 0.0f : wraw;        // weight of the reconstruction error
    const float* __restrict__ frow = feat_rows + (frame + in_shift) * d0;
    const float* __restrict__ frow_t = feat_rows + (frame + reg.lag_t) * d0;
    CVF_STAMP(20);
    const int n_img = lay.zb_off / AP;   // rows of the images (activations of layers 1..L-1, each with its row of ones)
    float* hand = reg.hand != nullptr ? reg.hand + tile * (int64_t)(n_img + dL) * CVF_TILE + fcol : nullptr;
    if (hand != nullptr && reg.hand_mode == 2) {
      // the statistics pass of this step left the activations of this tile: this wave's 16 frame columns of every row
      for (int r = kq; r < n_img + dL; r += 4) (r < n_img ? lds + r * AP : ZB + (r - n_img) * AP)[fcol] = hand[(int64_t)r * CVF_TILE];
    }
    // ---- forward
    for (int l = 0; l < (hand != nullptr && reg.hand_mode == 2 ? 0 : L); ++l) {
      const int din = s_dims[l], dout = s_dims[l + 1];
      const float* Wl = (l == 0 && lay.skip0 > 0 ? theta : WL) + s_woff[l];
      const float* bl = WL + s_boff[l];
      const float* in = l > 0 ? lds + s_img[l] : nullptr;
      float* dst = l + 1 < L ? lds + s_img[l + 1] : ZB;
      const int act = s_act[l];
      auto finish = [&](int rt, const f32x4& acc) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = 16 * rt + 4 * kq + r;
          if (o < dout) {
            const float v = acc[r] + bl[o];
            dst[o * AP + fcol] = act_f<TANH>(act, v);
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
    if (hand != nullptr && reg.hand_mode == 1) {
      for (int r = kq; r < n_img + dL; r += 4) hand[(int64_t)r * CVF_TILE] = (r < n_img ? lds + r * AP : ZB + (r - n_img) * AP)[fcol];
    }
    CVF_STAMP(21);
    if (reg.enc_tiled != nullptr && reg.write_y && !lagged) {   // the latent vector of this wave's frames
      const float* ei = lds + s_img[reg.enc_layer];
      for (int j = kq; j < reg.k_enc; j += 4) reg.enc_tiled[(t0 * reg.k_enc + j) * CVF_TILE + fcol] = ei[j * AP + fcol];
    }
    // ---- weighted squared error and zbar_L = 2 w (out - f) / sum(w)     (core.py:666); rows o = 16 rt + 4 kq + r
    {
      const float scale = (float)(2.0 * (double)wb * inv_wsum);
      const int act_last = s_act[L - 1];
      float err2 = 0.0f;
      for (int rt0 = 0; 16 * rt0 < dL; rt0 += 4) {
        float fv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int o = 16 * (rt0 + (u >> 2)) + 4 * kq + (u & 3);
          fv[u] = frow_t[o < n_mse ? o : n_mse - 1];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int o = 16 * (rt0 + (u >> 2)) + 4 * kq + (u & 3);
          if (o < n_mse) {
            const float out = ZB[o * AP + fcol];
            const float df = out - fv[u];
            err2 = fmaf(df, df, err2);
            float zb = scale * df;
            if (act_last) zb *= act_d<TANH>(act_last, out);
            ZB[o * AP + fcol] = zb;
          } else if (o < dL) {
            // regulariser head i: output gradient of the transfer-operator loss (as ef_bwd_mfma_kernel, lag_idx > 0)
            const int i = o - n_mse, K = reg.K;
            const float out = ZB[o * AP + fcol];
            if (reg.write_y) reg.y_tiled[(tile * K + i) * CVF_TILE + fcol] = out;
            float zb = 0.0f;
            if (with_grad && reg.coef != nullptr) {   // (no coefficients: the regulariser is switched off, gamma = 0)
              const float* yb = reg.y_tiled + t0 * K * CVF_TILE + fcol;
              const float* yl = reg.y_tiled + (reg.T + t0) * K * CVF_TILE + fcol;
              const double* gS1 = reg.coef;
              const double* gS2 = reg.coef + K;
              const double* gT = reg.coef + K + K * K;
              const double* gS1l = reg.coef + 2 * K + K * K;
              const double* gS2l = reg.coef + 3 * K + K * K;
              const double diff = (double)yl[i * CVF_TILE] - (double)yb[i * CVF_TILE];
              const double tterm = 2.0 * (double)wraw * gT[i] * diff;
              double g;
              if (!lagged) {
                double a = gS1[i];
                for (int j = 0; j < K; ++j) a += (j == i ? 2.0 : 1.0) * gS2[i * K + j] * (double)yb[j * CVF_TILE];
                g = (double)wraw * a - tterm;
              } else {
                const float wl = valid ? reg.w_lag[bb] : 0.0f;
                g = (double)wl * (gS1l[i] + 2.0 * gS2l[i] * (double)yl[i * CVF_TILE]) + tterm;
              }
              zb = (float)(reg.head_scale * g);
              if (act_last) zb *= act_d<TANH>(act_last, out);
            }
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
          const int64_t fb = t0 * CVF_TILE + 16 * (jc >> 2) + 4 * kq + (jc & 3);
          const int64_t fbc = fb < B ? fb : B - 1;
          foff[jc] = ((idx ? idx[fbc] : fbc) + in_shift) * d0;
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
        const int act = s_act[l - 1];
        const bool add_enc = reg.enc_coef != nullptr && l == reg.enc_layer && !lagged;
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
                  if (act) v *= act_d<TANH>(act, al[i * AP + fcol]);
                  Zn[i * AP + fcol] = v;
                }
              }
            }
          }
        }
        if (add_enc) {
          // + d (eta_1 norm + eta_2 orth) / d latent_i of this frame: w (gS1_i + sum_j c_ij gS2_ij e_j); the latent layer has
          // no activation (host check), so the term adds to zbar as it stands.  Kept out of the loops above: inside them it
          // cost the plain autoencoder 28 registers and its second workgroup per CU.
          for (int i = kq; i < din; i += 4) {
            double a = reg.enc_coef[i];
            for (int j = 0; j < din; ++j) a += (j == i ? 2.0 : 1.0) * reg.enc_coef[din + i * din + j] * (double)al[j * AP + fcol];
            Zn[i * AP + fcol] += (float)((double)wraw * a);
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

// ------------------------------------------------------------------------------------------------------------------
// The plain AutoEncoderTask step with the chain in REGISTERS (config 2: 55 us at B = 20 000 on the
// kernel above, whose every layer reads its operands back from the LDS images - 3-6 k cycles of dependent LDS round trips
// per layer for ~300 cycles of matrix instructions).  Same block = 64-frame tile, wave = 16 frames, same slab / partial
// outputs; what changes is the data flow of the two dependent chains:
//   * a vector of up to 80 units lives in the accumulator layout of v_mfma_f32_16x16x4_f32, PERMUTED so that register
//     (rt, r) of lane group q holds unit 4 (4 rt + r) + q: the next layer's k-step g then takes units 4 g .. 4 g + 3 straight
//     from register g of the four lane groups - a layer's output IS the next layer's B operand, no LDS, no lane movement;
//   * the weights (A operands) come from the block's LDS copy of theta with the matching index arithmetic, requested a
//     whole row tile ahead of its matrix instructions - they do not depend on the previous layer;
//   * the [unit][frame] images are still written (the weight gradients contract over the tile's 64 frames and need both
//     operands transposed), but nothing in the forward or backward-data chain waits for them;
//   * the reconstruction target is the input vector, already in registers in the same layout.
// One barrier per layer in the backward pass (the zbar image of all four waves), none in the forward pass.
// Covers chains whose hidden widths are <= 32 and whose input is <= 80 wide, Tanh / no activation (the reference's own
// autoencoders); everything else - and RegAutoEncoderTask's heads, lags and latent penalties - stays on the kernel above.
// ------------------------------------------------------------------------------------------------------------------
template <int RT>
struct PVec {
  f32x4 v[RT];
};
constexpr int kAe16Pad = 576;   // > 15 rows x 32 columns + 32: the farthest an unclamped read of the last layer's weights reaches
struct Ae16Lay {
  int img[CVF_MAX_LAYERS + 1];    // dword offset of image a_l, l = 1..L-1 (d_l rows + a row of ones)
  int zimg[CVF_MAX_LAYERS + 1];   // dword offset of image zbar_l, l = 1..L (d_l rows)
  int w, total;
};
__host__ __device__ inline Ae16Lay ae16_layout(const cvf_mlp_desc& m) {
  Ae16Lay lay = {};
  int rows = 0;
  for (int l = 1; l < m.n_layers; ++l) {
    lay.img[l] = rows * AP;
    rows += m.dims[l] + 1;
  }
  for (int l = 1; l <= m.n_layers; ++l) {
    lay.zimg[l] = rows * AP;
    rows += m.dims[l];
  }
  // (operand tiles read up to 15 rows past an image: into the next image or the weights below - finite values whose
  //  products land in output rows / columns that are discarded; the allocation ends 16 rows past the weights' start)
  lay.w = rows * AP;
  // behind theta: a zeroed pad that the unclamped weight reads of the last layer run into (at most 15 rows of <= 80 floats)
  const int wfl = ((m.n_params + 3) & ~3) + kAe16Pad;
  lay.total = lay.w + (wfl > 16 * AP ? wfl : 16 * AP);
  return lay;
}
__host__ inline bool ae16_shape(const cvf_mlp_desc* m) {
  if (m->n_nets != 1 || m->n_layers < 2 || m->n_layers > CVF_MAX_LAYERS) return false;
  if (m->dims[0] > 80 || m->dims[0] != m->dims[m->n_layers]) return false;
  for (int l = 1; l < m->n_layers; ++l)
    if (m->dims[l] > 32 || m->dims[l] < 1) return false;
  return chain_is_tanh(m) && (size_t)ae16_layout(*m).total * sizeof(float) <= 80 * 1024;   // two workgroups per CU
}

// unit of register r of row tile rt in lane group q / unit computed by A-operand row rho of row tile rt
__device__ __forceinline__ int pv_unit(int rt, int r, int q) { return 4 * (4 * rt + r) + q; }
__device__ __forceinline__ int pv_row_unit(int rt, int rho) { return 4 * (4 * rt + (rho & 3)) + (rho >> 2); }

template <int RT>
__device__ __forceinline__ void ae16_store_image(float* __restrict__ img, const PVec<RT>& x, int d, int kq, int fo) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int u = pv_unit(rt, r, kq);
      if (u < d) img[u * AP + fo] = x.v[rt][r];
    }
}
// one 16x16 tile of  A B^T  over the tile's 64 frames, operands = [row][frame] images (k-slot kq of k-step (j, c) = frame
// 16 j + 4 kq + c on both: eight 16-byte reads)
__device__ __forceinline__ f32x4 ae16_outer(const float* __restrict__ A, const float* __restrict__ Bm, int rt, int ct, int lane) {
  const int row = lane & 15, kq = lane >> 4;
  const float4* a = reinterpret_cast<const float4*>(A + (16 * rt + row) * AP + 4 * kq);
  const float4* b = reinterpret_cast<const float4*>(Bm + (16 * ct + row) * AP + 4 * kq);
  float4 av[4], bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    av[j] = a[4 * j];
    bv[j] = b[4 * j];
  }
  f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    acc = mfma4(av[j].x, bv[j].x, acc);
    acc = mfma4(av[j].y, bv[j].y, acc);
    acc = mfma4(av[j].z, bv[j].z, acc);
    acc = mfma4(av[j].w, bv[j].w, acc);
  }
  return acc;
}

// One layer product in the permuted accumulator layout, in two halves so that the NEXT layer's operands can be requested
// before the current layer's matrix instructions:
//   ae16_request: the A values (weights from the block's LDS copy of theta; the bias goes into the accumulators) of every row
//                 tile - unconditional, UNMASKED and unclamped reads: one base address per row tile plus a constant per
//                 k-step.  A value past the layer's rows or columns is a neighbouring weight or the zeroed pad behind theta -
//                 finite - and meets either a B value that is exactly 0 (padded input units are kept at 0) or lands in a
//                 padded output unit, which the callers set to 0 after the activation.  (A select on the load made the LOAD
//                 conditional - an exec-masked branch with its own wait per weight, forty serialized LDS round trips in the
//                 first layer; a 0 / 1 mask still cost three vector instructions per weight.)
//   ae16_apply:   out = b + W in (TRANSPOSED: out = W^T in), the row tiles' accumulation chains interleaved (a lone chain of
//                 v_mfma_f32_16x16x4_f32 waits 40 cycles per step for its own accumulator); only the k-steps and row tiles the
//                 layer has are executed (wave-uniform scalar branches).
template <int RTO, int RTI, bool TRANSPOSED>
struct Ae16Frag {
  float a[RTO][4 * RTI];
  f32x4 b[RTO];
};
template <int RTO, int RTI, bool TRANSPOSED>
__device__ __forceinline__ void ae16_request(Ae16Frag<RTO, RTI, TRANSPOSED>& f, const float* __restrict__ Wl, const float* __restrict__ bl,
                                             int din, int lane) {
  const int rho = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int rt = 0; rt < RTO; ++rt) {
    const int uo = pv_row_unit(rt, rho);
    const float* base = TRANSPOSED ? Wl + kq * din + uo : Wl + uo * din + kq;
#pragma unroll
    for (int g = 0; g < 4 * RTI; ++g) f.a[rt][g] = TRANSPOSED ? base[4 * g * din] : base[4 * g];
#pragma unroll
    for (int r = 0; r < 4; ++r) f.b[rt][r] = TRANSPOSED ? 0.0f : bl[pv_unit(rt, r, kq)];
  }
}
template <int RTO, int RTI, bool TRANSPOSED>
__device__ __forceinline__ void ae16_apply(PVec<RTO>& out, const Ae16Frag<RTO, RTI, TRANSPOSED>& f, int n_in, int n_out,
                                           const PVec<RTI>& in) {
  const int ngi = (n_in + 3) >> 2;
#pragma unroll
  for (int rt = 0; rt < RTO; ++rt) out.v[rt] = f.b[rt];
#pragma unroll
  for (int g = 0; g < 4 * RTI; ++g) {
    if (g < ngi) {   // wave-uniform
#pragma unroll
      for (int rt = 0; rt < RTO; ++rt)
        if (16 * rt < n_out) out.v[rt] = mfma4(f.a[rt][g], in.v[g >> 2][g & 3], out.v[rt]);   // wave-uniform
    }
  }
}
template <int RTO, int RTI>
__device__ __forceinline__ void ae16_mul(PVec<RTO>& out, const float* __restrict__ Wl, const float* __restrict__ bl, int din,
                                         int dout, const PVec<RTI>& in, int lane) {
  Ae16Frag<RTO, RTI, false> f;
  ae16_request<RTO, RTI, false>(f, Wl, bl, din, lane);
  ae16_apply<RTO, RTI, false>(out, f, din, dout, in);
}
// zin = W^T zout
template <int RTI, int RTO>
__device__ __forceinline__ void ae16_mul_t(PVec<RTI>& zin, const float* __restrict__ Wl, int din, int dout, const PVec<RTO>& zout,
                                           int lane) {
  Ae16Frag<RTI, RTO, true> f;
  ae16_request<RTI, RTO, true>(f, Wl, nullptr, din, lane);
  ae16_apply<RTI, RTO, true>(zin, f, dout, din, zout);
}

template <int RTD, int RTH>
__global__ __launch_bounds__(256, 2) void ae16_kernel(cvf_mlp_desc mlp, const float* __restrict__ theta,
                                                       const float* __restrict__ feat_rows, const int64_t* __restrict__ idx,
                                                       int64_t B, const float* __restrict__ w, double inv_wsum, int with_grad,
                                                       float* __restrict__ slab, double* __restrict__ partial,
                                                       int32_t* __restrict__ step, Ae16Lay lay) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  CVF_STAMP(18);
  const int col = lane & 15, kq = lane >> 4;
  const int fo = 16 * wv + col;              // this lane's frame column of the tile
  __shared__ int s_dims[CVF_MAX_LAYERS + 1], s_woff[CVF_MAX_LAYERS], s_boff[CVF_MAX_LAYERS], s_act[CVF_MAX_LAYERS];
  __shared__ int s_img[CVF_MAX_LAYERS + 1], s_zimg[CVF_MAX_LAYERS + 1];
  const int L = mlp.n_layers, d0 = mlp.dims[0];
  const int64_t T = (B + CVF_TILE - 1) / CVF_TILE;
  // ---- this lane's frame of a tile: index, weight, and the input vector (= the reconstruction target) in the permuted
  // accumulator layout.  Requested for the FIRST tile before anything else: the two dependent round trips (index, then the
  // row) run beside the copy of the weights into LDS.
  struct Frame {
    PVec<RTD> x;
    float wb;
  };
  auto load_frame = [&](int64_t tile, int fo, int kq) {
    Frame f;
    const int64_t b = tile * CVF_TILE + fo;
    const bool valid = b < B;
    const int64_t bb = valid ? b : B - 1;
    const int64_t frame = idx ? idx[bb] : bb;
    const float wr = w[bb];
    f.wb = valid ? wr : 0.0f;
    const float* __restrict__ frow = feat_rows + frame * d0;
#pragma unroll
    for (int rt = 0; rt < RTD; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int u = pv_unit(rt, r, kq);
        const float x = frow[u < d0 ? u : d0 - 1];
        f.x.v[rt][r] = x * (u < d0 ? 1.0f : 0.0f);   // (mask, not select: keeps the loads unconditional and batched)
      }
    return f;
  };
  // Order of the prologue's requests (vector memory returns in issue order): the first tile's frame index and weight, the
  // weights (theta), then - once the index is there - the frame's row; the LDS copy of theta waits for theta only, and the
  // barrier behind it for LDS traffic only, so the row's round trip runs into the first layer instead of in front of the barrier.
  Frame fr;
  const float* __restrict__ frow0;
  {
    const int64_t tile0 = blockIdx.x < T ? (int64_t)blockIdx.x : T - 1;
    const int64_t b = tile0 * CVF_TILE + fo;
    const bool valid = b < B;
    const int64_t bb = valid ? b : B - 1;
    const int64_t frame = idx ? idx[bb] : bb;
    const float wr = w[bb];
    fr.wb = valid ? wr : 0.0f;
    frow0 = feat_rows + frame * d0;
  }
  {
    // the layer table, one entry per thread, straight from the kernel-argument segment (`mlp` is the first argument): indexed
    // with a run-time index the by-value struct would be copied to scratch memory first (~6 k cycles of one thread's work)
    typedef const int __attribute__((address_space(4))) kernarg_int;
    kernarg_int* ka = (kernarg_int*)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int oDims = offsetof(cvf_mlp_desc, dims) / 4, oAct = offsetof(cvf_mlp_desc, act) / 4;
    constexpr int oW = offsetof(cvf_mlp_desc, w_off) / 4, oB = offsetof(cvf_mlp_desc, b_off) / 4;
    if (tid <= CVF_MAX_LAYERS) s_dims[tid] = ka[oDims + tid];
    if (tid < CVF_MAX_LAYERS) {
      s_woff[tid] = ka[oW + tid];
      s_boff[tid] = ka[oB + tid];
      s_act[tid] = ka[oAct + tid];
    }
    if (tid == 64) {
#pragma unroll
      for (int i = 0; i <= CVF_MAX_LAYERS; ++i) {   // (compile-time indices: scalar moves)
        s_img[i] = lay.img[i];
        s_zimg[i] = lay.zimg[i];
      }
    }
  }
  float* WL = lds + lay.w;
  {
    // theta -> LDS: 16-byte pieces, eight requests in flight per thread before the first LDS write (a load -> store loop is one
    // full memory round trip per 256 floats)
    const int n4 = mlp.n_params >> 2;
    const float4* t4 = reinterpret_cast<const float4*>(theta);   // (the flat parameter buffer starts 16-byte aligned: host check)
    float4* w4 = reinterpret_cast<float4*>(WL);
    float4 val[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int v = tid + 256 * u;
      val[u] = t4[v < n4 ? v : n4 - 1];
    }
    // the first tile's row (its address needs the index requested above)
#pragma unroll
    for (int rt = 0; rt < RTD; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int u = pv_unit(rt, r, kq);
        const float x = frow0[u < d0 ? u : d0 - 1];
        fr.x.v[rt][r] = x * (u < d0 ? 1.0f : 0.0f);
      }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int v = tid + 256 * u;
      if (v < n4) w4[v] = val[u];
    }
    for (int v0 = tid + 256 * 8; v0 < n4; v0 += 256) w4[v0] = t4[v0];   // (chains of more than 8192 parameters)
    for (int i = 4 * n4 + tid; i < mlp.n_params; i += 256) WL[i] = theta[i];
    for (int i = mlp.n_params + tid; i < lay.total - lay.w; i += 256) WL[i] = 0.0f;
  }
  lds_barrier();   // (LDS traffic only: the row requested above is still on its way)
  for (int l = 1; l < L; ++l)
    if (tid < 64) lds[__builtin_amdgcn_readfirstlane(s_img[l]) + __builtin_amdgcn_readfirstlane(s_dims[l]) * AP + tid] = 1.0f;   // the bias column's row of ones
  double loss_acc = 0.0, w_acc = 0.0;
  float* out_row = slab + (int64_t)blockIdx.x * mlp.n_params;
  CVF_STAMP(19);
  const int lane_outer = lane;
  for (int64_t tile = blockIdx.x; tile < T; tile += gridDim.x) {
    // (the lane number made opaque INSIDE the loop: everything below is loop-invariant address arithmetic to the compiler, which
    //  hoists all of it in front of a loop that normally runs once - ~200 live registers, spills, and the scalar ones parked in
    //  vector lanes)
    int lane = lane_outer;
    asm volatile("" : "+v"(lane));
    const int col = lane & 15, kq = lane >> 4, fo = 16 * wv + col;
    const bool first = tile == (int64_t)blockIdx.x;
    if (!first) fr = load_frame(tile, fo, kq);   // (uniform; launches of more than 2048 tiles only)
    const PVec<RTD> xin = fr.x;
    const float wb = fr.wb;
    CVF_STAMP(20);
    // ---- forward: first layer, hidden layers (the next layer's weights requested ahead of the current one's matrix
    // instructions), last layer
    PVec<RTH> cur;
    {
      const int d1 = __builtin_amdgcn_readfirstlane(s_dims[1]), act = __builtin_amdgcn_readfirstlane(s_act[0]);
      Ae16Frag<RTH, RTD, false> f0;
      ae16_request<RTH, RTD, false>(f0, WL + __builtin_amdgcn_readfirstlane(s_woff[0]), WL + __builtin_amdgcn_readfirstlane(s_boff[0]), d0, lane);
      ae16_apply<RTH, RTD, false>(cur, f0, d0, d1, xin);
#pragma unroll
      for (int rt = 0; rt < RTH; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) cur.v[rt][r] = (pv_unit(rt, r, kq) < d1) ? act_f<true>(act, cur.v[rt][r]) : 0.0f;
      ae16_store_image<RTH>(lds + __builtin_amdgcn_readfirstlane(s_img[1]), cur, d1, kq, fo);
    }
    CVF_STAMP(21);
    {
      Ae16Frag<RTH, RTH, false> fh;
      if (L > 2) ae16_request<RTH, RTH, false>(fh, WL + __builtin_amdgcn_readfirstlane(s_woff[1]), WL + __builtin_amdgcn_readfirstlane(s_boff[1]),
                                               __builtin_amdgcn_readfirstlane(s_dims[1]), lane);
      for (int l = 1; l + 1 < L; ++l) {
        const int din = __builtin_amdgcn_readfirstlane(s_dims[l]), dout = __builtin_amdgcn_readfirstlane(s_dims[l + 1]);
        const int act = __builtin_amdgcn_readfirstlane(s_act[l]);
        Ae16Frag<RTH, RTH, false> fn = fh;
        if (l + 2 < L)   // (uniform) the next hidden layer's weights, while this layer multiplies
          ae16_request<RTH, RTH, false>(fn, WL + __builtin_amdgcn_readfirstlane(s_woff[l + 1]), WL + __builtin_amdgcn_readfirstlane(s_boff[l + 1]), dout, lane);
        PVec<RTH> nxt;
        ae16_apply<RTH, RTH, false>(nxt, fh, din, dout, cur);
#pragma unroll
        for (int rt = 0; rt < RTH; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) cur.v[rt][r] = (pv_unit(rt, r, kq) < dout) ? act_f<true>(act, nxt.v[rt][r]) : 0.0f;
        ae16_store_image<RTH>(lds + __builtin_amdgcn_readfirstlane(s_img[l + 1]), cur, dout, kq, fo);
        fh = fn;
      }
    }
    CVF_STAMP(22);
    // (the target = the input vector is requested AGAIN here, behind the hidden layers, instead of holding twenty registers
    //  through them - with the fragments of two layers in flight that was the difference between 256 registers + spills and none;
    //  the rows are in L2 / the Infinity Cache and the round trip runs beside the last layer's matrix instructions)
    asm volatile("" ::: "memory");   // (not earlier: the compiler would hoist these loads to the top of the forward pass)
    const PVec<RTD> xt = load_frame(tile, fo, kq).x;
    PVec<RTD> zl;   // the output, then zbar_L
    const int dl1 = __builtin_amdgcn_readfirstlane(s_dims[L - 1]);
    ae16_mul<RTD, RTH>(zl, WL + __builtin_amdgcn_readfirstlane(s_woff[L - 1]), WL + __builtin_amdgcn_readfirstlane(s_boff[L - 1]), dl1, d0, cur, lane);
    // ---- weighted squared error and zbar_L = 2 w (out - f) / sum(w)     (core.py:666)
    {
      const float scale = (float)(2.0 * (double)wb * inv_wsum);
      const int act_last = __builtin_amdgcn_readfirstlane(s_act[L - 1]);
      float err2 = 0.0f;
#pragma unroll
      for (int rt = 0; rt < RTD; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool live = pv_unit(rt, r, kq) < d0;
          float out = zl.v[rt][r];
          if (act_last) out = act_f<true>(act_last, out);
          const float df = live ? out - xt.v[rt][r] : 0.0f;
          err2 = fmaf(df, df, err2);
          float zb = scale * df;
          if (act_last) zb *= act_d<true>(act_last, out);
          zl.v[rt][r] = zb;
        }
      loss_acc += (double)wb * (double)err2;
      if (kq == 0) w_acc += (double)wb;
    }
    CVF_STAMP(23);
    if (!with_grad) continue;   // (uniform) test pass: no images are read by other waves, no barrier needed
    // ---- backward data chain, all of it, in registers: zbar_l = (W_{l+1}^T zbar_{l+1}) .* act'(a_l), l = L-1 .. 1, every zbar
    // into an image of its own - ONE barrier for the whole backward pass instead of one per layer
    ae16_store_image<RTD>(lds + __builtin_amdgcn_readfirstlane(s_zimg[L]), zl, d0, kq, fo);
    {
      PVec<RTH> zc;
      {
        const int act = __builtin_amdgcn_readfirstlane(s_act[L - 2]);
        ae16_mul_t<RTH, RTD>(zc, WL + __builtin_amdgcn_readfirstlane(s_woff[L - 1]), dl1, d0, zl, lane);
        const float* al = lds + __builtin_amdgcn_readfirstlane(s_img[L - 1]);
#pragma unroll
        for (int rt = 0; rt < RTH; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int u = pv_unit(rt, r, kq);
            const float hv = al[(u < dl1 ? u : 0) * AP + fo];
            float v = zc.v[rt][r];
            if (act) v *= act_d<true>(act, hv);
            zc.v[rt][r] = u < dl1 ? v : 0.0f;
          }
        ae16_store_image<RTH>(lds + __builtin_amdgcn_readfirstlane(s_zimg[L - 1]), zc, dl1, kq, fo);
      }
      CVF_STAMP(24);
      Ae16Frag<RTH, RTH, true> th;
      if (L > 2) ae16_request<RTH, RTH, true>(th, WL + __builtin_amdgcn_readfirstlane(s_woff[L - 2]), nullptr,
                                              __builtin_amdgcn_readfirstlane(s_dims[L - 2]), lane);
      for (int l = L - 2; l >= 1; --l) {   // zbar_l from zbar_{l+1}
        const int din = __builtin_amdgcn_readfirstlane(s_dims[l]), dout = __builtin_amdgcn_readfirstlane(s_dims[l + 1]);
        const int act = __builtin_amdgcn_readfirstlane(s_act[l - 1]);
        Ae16Frag<RTH, RTH, true> tn = th;
        if (l >= 2)
          ae16_request<RTH, RTH, true>(tn, WL + __builtin_amdgcn_readfirstlane(s_woff[l - 1]), nullptr, __builtin_amdgcn_readfirstlane(s_dims[l - 1]), lane);
        const float* al = lds + __builtin_amdgcn_readfirstlane(s_img[l]);
        float hv[RTH][4];
#pragma unroll
        for (int rt = 0; rt < RTH; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int u = pv_unit(rt, r, kq);
            hv[rt][r] = al[(u < din ? u : 0) * AP + fo];
          }
        PVec<RTH> zp;
        ae16_apply<RTH, RTH, true>(zp, th, dout, din, zc);
#pragma unroll
        for (int rt = 0; rt < RTH; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int u = pv_unit(rt, r, kq);
            float v = zp.v[rt][r];
            if (act) v *= act_d<true>(act, hv[rt][r]);
            zc.v[rt][r] = u < din ? v : 0.0f;
          }
        ae16_store_image<RTH>(lds + __builtin_amdgcn_readfirstlane(s_zimg[l]), zc, din, kq, fo);
        th = tn;
      }
    }
    CVF_STAMP(25);
    // ---- the first layer's gradient tiles need B = [f ; 1] from GLOBAL memory (k-slot kq of k-step (j, c) is frame 16 j + 4 kq + c
    // of the tile): this wave's (at most kL0) tiles of that layer are known now, so their sixteen values per lane are requested
    // here, in front of the barrier and of the other layers' tiles - in the layer-0 loop itself each tile waited a full round trip
    constexpr int kL0 = 3;
    int base0 = 0;   // tiles of the layers L-1 .. 1 (they come first in the list the waves share round-robin)
    for (int l = L - 1; l >= 1; --l)
      base0 += ((__builtin_amdgcn_readfirstlane(s_dims[l]) + 1 + 15) / 16) * ((__builtin_amdgcn_readfirstlane(s_dims[l + 1]) + 15) / 16);
    const int nct0 = (d0 + 1 + 15) / 16, n0 = nct0 * ((__builtin_amdgcn_readfirstlane(s_dims[1]) + 15) / 16);
    const int pr0 = (wv - (base0 & 3) + 4) & 3;   // this wave's first tile of layer 0; then every fourth
    float bpre[kL0][16];
    {
      int64_t foff[16];
#pragma unroll
      for (int jc = 0; jc < 16; ++jc) {
        const int64_t fb = tile * CVF_TILE + 16 * (jc >> 2) + 4 * kq + (jc & 3);
        const int64_t fbc = fb < B ? fb : B - 1;
        foff[jc] = (idx ? idx[fbc] : fbc) * d0;
      }
#pragma unroll
      for (int t = 0; t < kL0; ++t) {
        const int pr = pr0 + 4 * t < n0 ? pr0 + 4 * t : 0;
        const int i = 16 * (pr % nct0) + col;
        const int ic = i < d0 ? i : d0 - 1;
#pragma unroll
        for (int jc = 0; jc < 16; ++jc) bpre[t][jc] = feat_rows[foff[jc] + ic];
      }
    }
    __syncthreads();   // every zbar image and every activation image of all four waves is in place
    CVF_STAMP(26);
    // ---- weight gradients: layer l's = zbar_{l+1} (x) [a_l ; 1] over the 64 frames, 16x16 tiles of ALL layers dealt round-robin
    // to the four waves (one list, no barrier in between); each tile has exactly one owner and goes straight to the slab row
    auto emit = [&](int l, int rt, int ct, const f32x4& acc) {
      const int din = __builtin_amdgcn_readfirstlane(s_dims[l]), dout = __builtin_amdgcn_readfirstlane(s_dims[l + 1]);
      const int wo = __builtin_amdgcn_readfirstlane(s_woff[l]), bo = __builtin_amdgcn_readfirstlane(s_boff[l]);
      const int i = 16 * ct + col;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = 16 * rt + 4 * kq + r;
        if (o < dout && i <= din) {
          float* dstp = out_row + (i < din ? wo + o * din + i : bo + o);
          *dstp = first ? acc[r] : *dstp + acc[r];
        }
      }
    };
    int next = wv;   // index of this wave's next tile in the list of all layers' tiles (layers L-1 .. 1, then layer 0)
    int base = 0;
    for (int l = L - 1; l >= 1; --l) {
      const int din = __builtin_amdgcn_readfirstlane(s_dims[l]), dout = __builtin_amdgcn_readfirstlane(s_dims[l + 1]);
      const int nct = (din + 1 + 15) / 16, nrt = (dout + 15) / 16;
      const float* Zimg = lds + __builtin_amdgcn_readfirstlane(s_zimg[l + 1]);
      const float* Aimg = lds + __builtin_amdgcn_readfirstlane(s_img[l]);
      for (; next < base + nrt * nct; next += 4) {
        const int pr = next - base, rt = pr / nct, ct = pr - rt * nct;
        emit(l, rt, ct, ae16_outer(Zimg, Aimg, rt, ct, lane));
      }
      base += nrt * nct;
    }
    CVF_STAMP(27);
    {   // l = 0 (B operands requested above; tiles past kL0 per wave - first layers wider than 12 column tiles - load here)
      const float* Zimg = lds + __builtin_amdgcn_readfirstlane(s_zimg[1]);
      int t = 0;
      for (int pr = pr0; pr < n0; pr += 4, ++t) {
        const int rt = pr / nct0, ct = pr - rt * nct0;
        const int i = 16 * ct + col;
        const float pad = i == d0 ? 1.0f : 0.0f;
        float bvals[16];
        if (t < kL0) {   // (uniform)
#pragma unroll
          for (int jc = 0; jc < 16; ++jc) {
            float v = bpre[0][jc];
#pragma unroll
            for (int u = 1; u < kL0; ++u) v = (t == u) ? bpre[u][jc] : v;
            bvals[jc] = v;
          }
        } else {
          const int ic = i < d0 ? i : d0 - 1;
#pragma unroll
          for (int jc = 0; jc < 16; ++jc) {
            const int64_t fb = tile * CVF_TILE + 16 * (jc >> 2) + 4 * kq + (jc & 3);
            const int64_t fbc = fb < B ? fb : B - 1;
            bvals[jc] = feat_rows[(idx ? idx[fbc] : fbc) * d0 + ic];
          }
        }
        const float4* za = reinterpret_cast<const float4*>(Zimg + (16 * rt + col) * AP + 4 * kq);
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 a = za[4 * j];
          acc = mfma4(a.x, i < d0 ? bvals[4 * j + 0] : pad, acc);
          acc = mfma4(a.y, i < d0 ? bvals[4 * j + 1] : pad, acc);
          acc = mfma4(a.z, i < d0 ? bvals[4 * j + 2] : pad, acc);
          acc = mfma4(a.w, i < d0 ? bvals[4 * j + 3] : pad, acc);
        }
        emit(0, rt, ct, acc);
      }
    }
    CVF_STAMP(28);
    __syncthreads();   // the next tile's forward overwrites the images
    CVF_STAMP(29);
  }
  // ---- per-block loss partials: fixed-order reduction over the block's four waves (as the kernel above)
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

template <class F>
bool ae16_dispatch(int d0, int hmax, F&& f) {
  const int rtd = (d0 + 15) / 16, rth = (hmax + 15) / 16;
#define AE16_CASE(D_, H_)                                                      \
  if (rtd == D_ && rth == H_) {                                                \
    f(std::integral_constant<int, D_>{}, std::integral_constant<int, H_>{});   \
    return true;                                                               \
  }
  AE16_CASE(1, 1) AE16_CASE(2, 1) AE16_CASE(3, 1) AE16_CASE(4, 1) AE16_CASE(5, 1)
  AE16_CASE(1, 2) AE16_CASE(2, 2) AE16_CASE(3, 2) AE16_CASE(4, 2) AE16_CASE(5, 2)
#undef AE16_CASE
  return false;
}

// out2 = fixed-order sums of the per-block {sum w*err, sum w} (+ their ratio): lane l adds rows l, l+64, ..., then the DPP reduction
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
    out2[2] = a0 / a1;   // the loss of a single-process run (core.py:666)
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
                l + 1 < upto ? lds + lay.act_off[l + 1] : ZB, mlp.act[l], lane);
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
  const int G = ae_grid(B);
  const int Pn = mlp->n_params;
  // scratch: [slab floats][partials as doubles, 8-byte aligned]
  float* slab = scratch;
  double* partial = reinterpret_cast<double*>(scratch + (((int64_t)G * Pn + 1) & ~(int64_t)1));
  hipStream_t s = (hipStream_t)stream;
  static const bool no_fast = getenv("CVF_NO_AE16") != nullptr;
  int hmax = 1;
  for (int l = 1; l < mlp->n_layers; ++l) hmax = mlp->dims[l] > hmax ? mlp->dims[l] : hmax;
  bool fast = !no_fast && ae16_shape(mlp) && (reinterpret_cast<uintptr_t>(theta) & 15) == 0;
  if (fast) {   // the chain in registers (ae16_kernel)
    const size_t lds16 = (size_t)ae16_layout(*mlp).total * sizeof(float);
    fast = ae16_dispatch(mlp->dims[0], hmax, [&](auto d_, auto h_) {
      auto kernel = ae16_kernel<decltype(d_)::value, decltype(h_)::value>;
      if (lds16 > 48 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);
      hipLaunchKernelGGL(kernel, dim3(G), dim3(256), lds16, s, *mlp, theta, feat_rows, idx, B, w, inv_wsum, grad ? 1 : 0, slab, partial,
                         grad ? step_count : nullptr, ae16_layout(*mlp));
    });
  }
  if (!fast) {
    const AeMLayout lay = ae_mlayout(*mlp, grad != nullptr);
    const size_t lds = (size_t)lay.total * sizeof(float);
    CVF_REQUIRE(lds <= 160 * 1024, "cvf_ae_step: the chain needs %zu B of LDS per workgroup (> 160 KiB)", lds);
    auto kernel = chain_is_tanh(mlp) ? ae_mfma_kernel<true> : ae_mfma_kernel<false>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    AeReg none = {};
    none.T = none.n_tiles = cvf_ntiles(B);
    hipLaunchKernelGGL(kernel, dim3(G), dim3(256), lds, s, *mlp, theta, feat_rows, idx, B, w, inv_wsum, grad ? 1 : 0, slab,
                       partial, grad ? step_count : nullptr, none);
  }
  int rc = cvf_check_launch(fast ? "ae16_kernel" : "ae_mfma_kernel");
  if (rc) return rc;
  if (grad == nullptr) {   // loss only: the blocks' [sum w e^2, sum w] pairs, fixed order
    hipLaunchKernelGGL(ae_loss_sum_kernel, dim3(1), dim3(64), 0, s, partial, G, out2);
    return cvf_check_launch("ae_loss_sum_kernel");
  }
  // fixed-order sum of the slab rows (+ the Adam update when asked for): the kernel the eigenfunction path uses; the loss
  // pairs ride along in one extra block of the same launch
  cvf_adam_args ad;
  if (adam != nullptr) {
    ad = *adam;
    ad.mlp = nullptr;      // an AutoEncoder has no MFMA fragment copy to refresh
    ad.packed = nullptr;
  }
  return cvf_slab_reduce_impl(slab, G, Pn, grad, nullptr, adam != nullptr ? &ad : nullptr, stream, partial, G, out2);
}

// ---- RegAutoEncoderTask (time-lagged autoencoder + transfer-operator regulariser heads)
static int regae_grid(int64_t n_tiles) { return (int)(n_tiles < kAeMaxBlocks ? n_tiles : kAeMaxBlocks); }

static int64_t regae_hand_rows(const cvf_mlp_desc* mlp) {   // image rows + last-layer outputs (see AeReg.hand)
  int64_t rows = mlp->dims[mlp->n_layers];
  for (int l = 1; l < mlp->n_layers; ++l) rows += mlp->dims[l] + 1;
  return rows;
}
static int64_t regae_hand_offset(const cvf_mlp_desc* mlp, int64_t B) {   // floats of the slab and the loss partials in front of it
  const int64_t G = regae_grid(2 * cvf_ntiles(B));
  return (G * mlp->n_params + 4 * G + 4 + 3) & ~(int64_t)3;
}
extern "C" int64_t cvf_regae_scratch_floats(const cvf_mlp_desc* mlp, int64_t B) {
  return regae_hand_offset(mlp, B) + 2 * cvf_ntiles(B) * regae_hand_rows(mlp) * CVF_TILE;
}

static int regae_launch(const cvf_mlp_desc* mlp, const float* theta, const float* feat_rows, const int64_t* idx, int64_t B,
                        const float* w, double mse_scale, bool with_grad, float* scratch, int32_t* step_count, AeReg reg,
                        int* grid_out, hipStream_t s, bool handoff = false) {
  CVF_REQUIRE(mlp->n_nets == 1 && mlp->n_layers >= 2 && mlp->n_layers <= CVF_MAX_LAYERS, "cvf_regae: one chain expected");
  CVF_REQUIRE(reg.K >= 0 && reg.K <= CVF_MAX_NETS && mlp->dims[mlp->n_layers] == mlp->dims[0] + reg.K,
              "cvf_regae: the chain must end in d_0 = %d reconstruction rows + K = %d heads (it has %d outputs)", mlp->dims[0],
              reg.K, mlp->dims[mlp->n_layers]);
  CVF_REQUIRE(reg.lag_t >= 0 && reg.lag_in >= 0 && (reg.coef == nullptr || reg.lag_in > 0),
              "cvf_regae: the regulariser needs lag_input > 0 (transfer-operator loss; the generator loss is not built for this task)");
  if (reg.enc_tiled != nullptr || reg.enc_coef != nullptr) {
    CVF_REQUIRE(reg.enc_layer >= 1 && reg.enc_layer < mlp->n_layers, "cvf_regae: n_enc_layers=%d out of range", reg.enc_layer);
    reg.k_enc = mlp->dims[reg.enc_layer];
    CVF_REQUIRE(mlp->act[reg.enc_layer - 1] == 0, "cvf_regae: the encoder's last layer must have no activation");
    CVF_REQUIRE(reg.k_enc <= CVF_MAX_NETS, "cvf_regae: latent width %d > %d", reg.k_enc, CVF_MAX_NETS);
  }
  const AeMLayout lay = ae_mlayout(*mlp, with_grad);
  const size_t lds = (size_t)lay.total * sizeof(float);
  CVF_REQUIRE(lds <= 160 * 1024, "cvf_regae: the chain needs %zu B of LDS per workgroup (> 160 KiB)", lds);
  reg.T = cvf_ntiles(B);
  reg.n_tiles = (reg.K > 0 && reg.lag_in > 0) ? 2 * reg.T : reg.T;
  const int G = regae_grid(reg.n_tiles);
  *grid_out = G;
  if (handoff) {
    reg.hand = scratch + regae_hand_offset(mlp, B);
    reg.hand_mode = with_grad ? 2 : 1;
  }
  float* slab = scratch;
  double* partial = reinterpret_cast<double*>(scratch + (((int64_t)G * mlp->n_params + 1) & ~(int64_t)1));
  auto kernel = chain_is_tanh(mlp) ? ae_mfma_kernel<true> : ae_mfma_kernel<false>;
  if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kernel, dim3(G), dim3(256), lds, s, *mlp, theta, feat_rows, idx, B, w, mse_scale, with_grad ? 1 : 0, slab,
                     partial, with_grad ? step_count : nullptr, reg);
  return cvf_check_launch("ae_mfma_kernel");
}

static int regae_forward_impl(const cvf_mlp_desc* mlp, const float* theta, const float* feat_rows, const int64_t* idx,
                              int64_t B, int64_t lag_target, int64_t lag_input, int K, const float* w, float* scratch,
                              float* y_tiled, int n_enc_layers, float* enc_tiled, double* out2, void* stream, bool keep) {
  CVF_REQUIRE(mlp && theta && feat_rows && w && scratch && out2 && B > 0 && (K == 0 || y_tiled), "cvf_regae_forward: bad argument");
  AeReg reg = {};
  reg.K = K;
  reg.write_y = K > 0 || enc_tiled != nullptr;
  reg.lag_t = lag_target;
  reg.lag_in = lag_input;
  reg.y_tiled = y_tiled;
  reg.enc_layer = n_enc_layers;
  reg.enc_tiled = enc_tiled;
  int G = 0;
  hipStream_t s = (hipStream_t)stream;
  int rc = regae_launch(mlp, theta, feat_rows, idx, B, w, 0.0, false, scratch, nullptr, reg, &G, s, keep);
  if (rc) return rc;
  double* partial = reinterpret_cast<double*>(scratch + (((int64_t)G * mlp->n_params + 1) & ~(int64_t)1));
  hipLaunchKernelGGL(ae_loss_sum_kernel, dim3(1), dim3(64), 0, s, partial, G, out2);
  return cvf_check_launch("ae_loss_sum_kernel");
}
extern "C" int cvf_regae_forward(const cvf_mlp_desc* mlp, const float* theta, const float* feat_rows, const int64_t* idx,
                                 int64_t B, int64_t lag_target, int64_t lag_input, int K, const float* w, float* scratch,
                                 float* y_tiled, int n_enc_layers, float* enc_tiled, double* out2, void* stream) {
  return regae_forward_impl(mlp, theta, feat_rows, idx, B, lag_target, lag_input, K, w, scratch, y_tiled, n_enc_layers, enc_tiled,
                            out2, stream, false);
}
extern "C" int cvf_regae_forward_keep(const cvf_mlp_desc* mlp, const float* theta, const float* feat_rows, const int64_t* idx,
                                      int64_t B, int64_t lag_target, int64_t lag_input, int K, const float* w, float* scratch,
                                      float* y_tiled, int n_enc_layers, float* enc_tiled, double* out2, void* stream) {
  return regae_forward_impl(mlp, theta, feat_rows, idx, B, lag_target, lag_input, K, w, scratch, y_tiled, n_enc_layers, enc_tiled,
                            out2, stream, true);
}

static int regae_backward_impl(const cvf_mlp_desc* mlp, const float* theta, const float* feat_rows, const int64_t* idx,
                               int64_t B, int64_t lag_target, int64_t lag_input, int K, const float* w, const float* w_lag,
                               double mse_scale, double head_scale, const float* y_tiled, const double* coef,
                               int n_enc_layers, const double* enc_coef, float* scratch, float* grad, const float* mask,
                               int32_t* step_count, const cvf_adam_args* adam, void* stream, bool reuse) {
  CVF_REQUIRE(mlp && theta && feat_rows && w && scratch && grad && B > 0, "cvf_regae_backward: bad argument");
  CVF_REQUIRE(coef == nullptr || (K > 0 && w_lag && y_tiled), "cvf_regae_backward: the regulariser needs heads, w_lag and y_tiled");
  CVF_REQUIRE(adam == nullptr || (adam->theta && adam->m && adam->v && adam->step_count), "cvf_regae_backward: incomplete adam arguments");
  AeReg reg = {};
  reg.K = K;
  reg.lag_t = lag_target;
  reg.lag_in = lag_input;
  reg.head_scale = head_scale;
  reg.coef = coef;
  reg.w_lag = w_lag;
  reg.y_tiled = const_cast<float*>(y_tiled);
  reg.enc_layer = n_enc_layers;
  reg.enc_coef = enc_coef;
  int G = 0;
  int rc = regae_launch(mlp, theta, feat_rows, idx, B, w, mse_scale, true, scratch, step_count, reg, &G, (hipStream_t)stream, reuse);
  if (rc) return rc;
  cvf_adam_args ad;
  if (adam != nullptr) {
    ad = *adam;
    ad.mlp = nullptr;
    ad.packed = nullptr;
  }
  return cvf_slab_reduce_impl(scratch, G, mlp->n_params, grad, mask, adam != nullptr ? &ad : nullptr, stream);
}
extern "C" int cvf_regae_backward(const cvf_mlp_desc* mlp, const float* theta, const float* feat_rows, const int64_t* idx,
                                  int64_t B, int64_t lag_target, int64_t lag_input, int K, const float* w, const float* w_lag,
                                  double mse_scale, double head_scale, const float* y_tiled, const double* coef,
                                  int n_enc_layers, const double* enc_coef, float* scratch, float* grad, const float* mask,
                                  int32_t* step_count, const cvf_adam_args* adam, void* stream) {
  return regae_backward_impl(mlp, theta, feat_rows, idx, B, lag_target, lag_input, K, w, w_lag, mse_scale, head_scale, y_tiled, coef,
                             n_enc_layers, enc_coef, scratch, grad, mask, step_count, adam, stream, false);
}
extern "C" int cvf_regae_backward_reuse(const cvf_mlp_desc* mlp, const float* theta, const float* feat_rows, const int64_t* idx,
                                        int64_t B, int64_t lag_target, int64_t lag_input, int K, const float* w, const float* w_lag,
                                        double mse_scale, double head_scale, const float* y_tiled, const double* coef,
                                        int n_enc_layers, const double* enc_coef, float* scratch, float* grad, const float* mask,
                                        int32_t* step_count, const cvf_adam_args* adam, void* stream) {
  return regae_backward_impl(mlp, theta, feat_rows, idx, B, lag_target, lag_input, K, w, w_lag, mse_scale, head_scale, y_tiled, coef,
                             n_enc_layers, enc_coef, scratch, grad, mask, step_count, adam, stream, true);
}

// row of RegAutoEncoderTask's loss list (core.py:1112-1124): [loss, ae, npl, pen, eig_1..K, enc_grad (0), enc_norm, enc_orth]
__global__ void regae_loss_row_kernel(const double* __restrict__ out2, const double* __restrict__ loss_vec, double alpha, double g0,
                                      double g1, int K, const double* __restrict__ enc_terms, double eta1, double eta2,
                                      double* __restrict__ row) {
  if (threadIdx.x != 0) return;
  const double ae = alpha != 0.0 ? out2[0] / out2[1] : 0.0;
  const double npl = loss_vec ? loss_vec[1] : 0.0, pen = loss_vec ? loss_vec[2] : 0.0;
  const double en = (enc_terms && eta1 != 0.0) ? enc_terms[0] : 0.0, eo = (enc_terms && eta2 != 0.0) ? enc_terms[1] : 0.0;
  row[0] = alpha * ae + g0 * npl + g1 * pen + eta1 * en + eta2 * eo;
  row[1] = ae;
  row[2] = npl;
  row[3] = pen;
  for (int i = 0; i < K; ++i) row[4 + i] = loss_vec ? loss_vec[3 + i] : 0.0;
  row[4 + K] = 0.0;
  row[5 + K] = en;
  row[6 + K] = eo;
}

extern "C" int cvf_regae_loss_row(const double* out2, const double* loss_vec, double alpha, double gamma0, double gamma1, int K,
                                  const double* enc_terms, double eta1, double eta2, double* row, void* stream) {
  CVF_REQUIRE(out2 && row && K >= 0 && K <= CVF_MAX_NETS, "cvf_regae_loss_row: bad argument");
  hipLaunchKernelGGL(regae_loss_row_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out2, loss_vec, alpha, gamma0, gamma1, K,
                     enc_terms, eta1, eta2, row);
  return cvf_check_launch("regae_loss_row_kernel");
}

// Encoder regularisers (core.py:912-971) from the latent vector's batch sums [W, S1(k), S2(i<=j), ...] (cvf_ef_stats layout):
// terms = {sum_j (var_j - 1)^2, sum_{i<j} cov_ij^2}; coef = d(eta1 terms[0] + eta2 terms[1]) / d{S1_j, S2_ij} in the layout the
// backward pass reads ([gS1(k), gS2(k*k) symmetric]; a frame's latent gradient is w (gS1_i + sum_j c_ij gS2_ij e_j), c_ii = 2)
__global__ void regae_enc_loss_kernel(const double* __restrict__ stats, int k, double eta1, double eta2, double* __restrict__ terms,
                                      double* __restrict__ coef) {
  if (threadIdx.x != 0) return;
  const double W = stats[0];
  const double* S1 = stats + 1;
  const double* S2 = stats + 1 + k;
  auto s2 = [&](int i, int j) { return S2[i * k - (i * (i - 1)) / 2 + (j - i)]; };   // i <= j
  double mean[CVF_MAX_NETS], var[CVF_MAX_NETS];
  for (int j = 0; j < k; ++j) {
    mean[j] = S1[j] / W;
    var[j] = s2(j, j) / W - mean[j] * mean[j];
  }
  double tn = 0.0, to = 0.0;
  for (int j = 0; j < k; ++j) {
    tn += (var[j] - 1.0) * (var[j] - 1.0);
    coef[j] = eta1 * 2.0 * (var[j] - 1.0) * (-2.0 * mean[j] / W);
    for (int i = 0; i < k; ++i) coef[k + j * k + i] = 0.0;
  }
  for (int j = 0; j < k; ++j) coef[k + j * k + j] = eta1 * 2.0 * (var[j] - 1.0) / W;
  for (int i = 0; i < k; ++i)
    for (int j = i + 1; j < k; ++j) {
      const double cov = s2(i, j) / W - mean[i] * mean[j];
      to += cov * cov;
      const double g = eta2 * 2.0 * cov;
      coef[k + i * k + j] = coef[k + j * k + i] = g / W;
      coef[i] += g * (-mean[j] / W);
      coef[j] += g * (-mean[i] / W);
    }
  terms[0] = tn;
  terms[1] = to;
}

extern "C" int cvf_regae_enc_loss(const double* stats, int k, double eta1, double eta2, double* terms, double* coef, void* stream) {
  CVF_REQUIRE(stats && terms && coef && k >= 1 && k <= CVF_MAX_NETS, "cvf_regae_enc_loss: bad argument (k=%d)", k);
  hipLaunchKernelGGL(regae_enc_loss_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, stats, k, eta1, eta2, terms, coef);
  return cvf_check_launch("regae_enc_loss_kernel");
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
