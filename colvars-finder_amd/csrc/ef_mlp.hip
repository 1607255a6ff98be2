// K4a / K4b: the k eigenfunction nets (colvarsfinder.nn.EigenFunctions, nn.py:242-293).
//
// Shape handled by these kernels: every net is  d0 -> H -> ... -> H -> 1  with NH hidden
// layers of equal width H and tanh after every Linear but the last (nn.py:52-59).
// One lane = one frame: the matrix-vector chains of a 20-wide net have no reuse across
// frames that a tile could exploit beyond the weights themselves, and those are
// wave-uniform, so they come through the scalar cache as SGPR operands of v_fma_f32 -
// the VALU runs at its full 64 lanes with no padding of 20 to 32.  The only genuine
// dense contractions are the weight gradients  W_l += sum_frames zbar_l (x) h_{l-1}:
// those run on the matrix cores (v_mfma_f32_16x16x4_f32, K = frames), operands staged
// through wave-private LDS in [row][frame] order with a 66-dword row pitch (conflict-free
// for both the lane=frame writes and the MFMA operand reads).
#include "cvf_common.hpp"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kPitch = 66;  // LDS row pitch (dwords) of the [row][frame] operand images
constexpr int kMaxCT1 = 8;  // column tiles (16 wide) of the first layer handled per block: d0+1 <= 128

struct EfBwdArgs {
  int k;
  int lag_idx;          // 0: generator (tangent terms on), >0: transfer (feat/y hold 2T tiles)
  int64_t B;
  int64_t T;            // tiles per pass
  int64_t n_tiles;      // T or 2T
};

// ---- forward chain for one lane; h[l][o] = tanh(z_l[o]) ---------------------------------
template <int H, int NH>
__device__ __forceinline__ float ef_forward(const cvf_mlp_desc& mlp, const float* __restrict__ theta, int net,
                                            const float* __restrict__ f /* + lane, stride 64 */, float (&h)[NH][H]) {
  const int D = mlp.dims[0];
  {
    const float* __restrict__ W = theta + mlp.w_off[net][0];
    const float* __restrict__ b = theta + mlp.b_off[net][0];
#pragma unroll
    for (int o = 0; o < H; ++o) h[0][o] = b[o];
    for (int i = 0; i < D; ++i) {
      const float fi = f[i * CVF_TILE];
#pragma unroll
      for (int o = 0; o < H; ++o) h[0][o] = fmaf(W[o * D + i], fi, h[0][o]);
    }
#pragma unroll
    for (int o = 0; o < H; ++o) h[0][o] = cvf_tanh(h[0][o]);
  }
#pragma unroll
  for (int l = 1; l < NH; ++l) {
    const float* __restrict__ W = theta + mlp.w_off[net][l];
    const float* __restrict__ b = theta + mlp.b_off[net][l];
#pragma unroll
    for (int o = 0; o < H; ++o) {
      float acc = b[o];
#pragma unroll
      for (int i = 0; i < H; ++i) acc = fmaf(W[o * H + i], h[l - 1][i], acc);
      h[l][o] = cvf_tanh(acc);
    }
  }
  const float* __restrict__ W = theta + mlp.w_off[net][NH];
  float y = theta[mlp.b_off[net][NH]];
#pragma unroll
  for (int i = 0; i < H; ++i) y = fmaf(W[i], h[NH - 1][i], y);
  return y;
}

// ---- input-gradient chain: d[l] = dy/dz_l ---------------------------------------------------
template <int H, int NH>
__device__ __forceinline__ void ef_dchain(const cvf_mlp_desc& mlp, const float* __restrict__ theta, int net,
                                          const float (&h)[NH][H], float (&d)[NH][H]) {
  const float* __restrict__ WL = theta + mlp.w_off[net][NH];
#pragma unroll
  for (int i = 0; i < H; ++i) d[NH - 1][i] = WL[i] * (1.0f - h[NH - 1][i] * h[NH - 1][i]);
#pragma unroll
  for (int l = NH - 1; l >= 1; --l) {
    const float* __restrict__ W = theta + mlp.w_off[net][l];
    float e[H];
#pragma unroll
    for (int i = 0; i < H; ++i) e[i] = 0.0f;
#pragma unroll
    for (int o = 0; o < H; ++o)
#pragma unroll
      for (int i = 0; i < H; ++i) e[i] = fmaf(W[o * H + i], d[l][o], e[i]);
#pragma unroll
    for (int i = 0; i < H; ++i) d[l - 1][i] = e[i] * (1.0f - h[l - 1][i] * h[l - 1][i]);
  }
}

template <int H, int NH>
__global__ __launch_bounds__(64) void ef_fwd_kernel(cvf_mlp_desc mlp, const float* __restrict__ theta,
                                                     const float* __restrict__ feat, float* __restrict__ y_tiled,
                                                     float* __restrict__ g_tiled) {
  const int lane = threadIdx.x;
  const int64_t tile = blockIdx.x;
  const int net = blockIdx.y;
  const int k = mlp.n_nets;
  const int D = mlp.dims[0];
  float h[NH][H];
  const float y = ef_forward<H, NH>(mlp, theta, net, feat + tile * D * CVF_TILE + lane, h);
  y_tiled[(tile * k + net) * CVF_TILE + lane] = y;
  if (g_tiled == nullptr) return;
  float d[NH][H];
  ef_dchain<H, NH>(mlp, theta, net, h, d);
  const float* __restrict__ W = theta + mlp.w_off[net][0];
  float* g = g_tiled + (tile * k + net) * (int64_t)D * CVF_TILE + lane;
  for (int i = 0; i < D; ++i) {
    float acc = 0.0f;
#pragma unroll
    for (int o = 0; o < H; ++o) acc = fmaf(W[o * D + i], d[0][o], acc);
    g[i * CVF_TILE] = acc;
  }
}

// ---- MFMA accumulation of  acc[rt][ct] += A(rows 16rt..) x B(rows 16ct..)  over 64 frames -------------
template <int RT, int CT>
__device__ __forceinline__ void mfma_outer(const float* __restrict__ A, const float* __restrict__ Bm, int lane,
                                           f32x4 (&acc)[RT][CT]) {
  const int row = lane & 15, kq = lane >> 4;
#pragma unroll 4
  for (int s = 0; s < 16; ++s) {
    float a[RT], b[CT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) a[rt] = A[(16 * rt + row) * kPitch + 4 * s + kq];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) b[ct] = Bm[(16 * ct + row) * kPitch + 4 * s + kq];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt], b[ct], acc[rt][ct], 0, 0, 0);
  }
}

template <int H, int NH>
__global__ __launch_bounds__(64) void ef_bwd_kernel(EfBwdArgs args, cvf_mlp_desc mlp, const float* __restrict__ theta,
                                                     const float* __restrict__ w, const float* __restrict__ w_lag,
                                                     const float* __restrict__ feat, const float* __restrict__ y_tiled,
                                                     const float* __restrict__ q_tiled, const double* __restrict__ coef,
                                                     float* __restrict__ slab) {
  constexpr int RT = (H + 15) / 16;      // row tiles of an H-row operand
  constexpr int CTH = (H + 1 + 15) / 16; // column tiles of [h ; 1]
  constexpr int RA = RT * 16, RB = CTH * 16;
  __shared__ float SA1[RA * kPitch], SA2[RA * kPitch], SB1[RB * kPitch], SB2[RB * kPitch];
  const int lane = threadIdx.x;
  const int net = blockIdx.y;
  const int k = args.k;
  const int D = mlp.dims[0];
  const int CT1 = (D + 1 + 15) / 16;
  const bool tangent = args.lag_idx == 0;
  const int row16 = lane & 15, kq = lane >> 4;

  // zero the operand images once: rows past H (and past H+1 in B) stay zero for the whole kernel
  for (int i = lane; i < RA * kPitch; i += 64) { SA1[i] = 0.0f; SA2[i] = 0.0f; }
  for (int i = lane; i < RB * kPitch; i += 64) { SB1[i] = 0.0f; SB2[i] = 0.0f; }
  __syncthreads();

  f32x4 acc1[RT][kMaxCT1];
  f32x4 accH[NH > 1 ? NH - 1 : 1][RT][CTH];
  f32x4 accL[1][CTH];
  const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < kMaxCT1; ++ct) acc1[rt][ct] = zero4;
#pragma unroll
  for (int l = 0; l < (NH > 1 ? NH - 1 : 1); ++l)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ct = 0; ct < CTH; ++ct) accH[l][rt][ct] = zero4;
#pragma unroll
  for (int ct = 0; ct < CTH; ++ct) accL[0][ct] = zero4;

  // coefficients of this net (wave-uniform)
  const double* gS1 = coef;
  const double* gS2 = coef + k;
  const double* gEt = coef + k + k * k;
  const double* gS1l = coef + 2 * k + k * k;
  const double* gS2l = coef + 3 * k + k * k;

  for (int64_t tile = blockIdx.x; tile < args.n_tiles; tile += gridDim.x) {
    const int pass = tile >= args.T ? 1 : 0;
    const int64_t t0 = pass ? tile - args.T : tile;  // tile index within its pass
    const int64_t frame = t0 * CVF_TILE + lane;
    const bool valid = frame < args.B;
    const float wb = valid ? w[frame] : 0.0f;
    // ---- per-frame coefficients alpha = dL/dy_net, gamma = 2 w gE (generator)
    float alpha, gamma = 0.0f;
    {
      const float* yb = y_tiled + t0 * k * CVF_TILE + lane;  // pass-0 outputs
      if (args.lag_idx == 0) {
        double acc = gS1[net];
        for (int j = 0; j < k; ++j) {
          const double yj = (double)yb[j * CVF_TILE];
          acc += (j == net ? 2.0 : 1.0) * gS2[net * k + j] * yj;
        }
        alpha = (float)((double)wb * acc);
        gamma = (float)(2.0 * (double)wb * gEt[net]);
      } else {
        const float* yl = y_tiled + (args.T + t0) * k * CVF_TILE + lane;  // lagged outputs
        const double diff = (double)yl[net * CVF_TILE] - (double)yb[net * CVF_TILE];
        const double tterm = 2.0 * (double)wb * gEt[net] * diff;
        if (pass == 0) {
          double acc = gS1[net];
          for (int j = 0; j < k; ++j) {
            const double yj = (double)yb[j * CVF_TILE];
            acc += (j == net ? 2.0 : 1.0) * gS2[net * k + j] * yj;
          }
          alpha = (float)((double)wb * acc - tterm);
        } else {
          const float wl = valid ? w_lag[frame] : 0.0f;
          const double acc = gS1l[net] + 2.0 * gS2l[net] * (double)yl[net * CVF_TILE];
          alpha = (float)((double)wl * acc + tterm);
        }
      }
    }
    const float* f = feat + tile * (int64_t)D * CVF_TILE + lane;
    const float* q = tangent ? q_tiled + (tile * k + net) * (int64_t)D * CVF_TILE + lane : nullptr;

    float h[NH][H];
    ef_forward<H, NH>(mlp, theta, net, f, h);
    float d[NH][H];
    float t[NH][H];  // t_l = W_l tdot_{l-1}
    if (tangent) {
      ef_dchain<H, NH>(mlp, theta, net, h, d);
      const float* __restrict__ W = theta + mlp.w_off[net][0];
#pragma unroll
      for (int o = 0; o < H; ++o) t[0][o] = 0.0f;
      for (int i = 0; i < D; ++i) {
        const float qi = q[i * CVF_TILE];
#pragma unroll
        for (int o = 0; o < H; ++o) t[0][o] = fmaf(W[o * D + i], qi, t[0][o]);
      }
#pragma unroll
      for (int o = 0; o < H; ++o) t[0][o] *= gamma;
#pragma unroll
      for (int l = 1; l < NH; ++l) {
        const float* __restrict__ Wl = theta + mlp.w_off[net][l];
#pragma unroll
        for (int o = 0; o < H; ++o) {
          float acc = 0.0f;
#pragma unroll
          for (int i = 0; i < H; ++i)
            acc = fmaf(Wl[o * H + i], (1.0f - h[l - 1][i] * h[l - 1][i]) * t[l - 1][i], acc);
          t[l][o] = acc;
        }
      }
    }

    // ---- last layer (1 x H): W += alpha h_{NH-1} + tdot_{NH-1} ; b += alpha
    {
      SA1[lane] = alpha;                       // row 0 of A1
      if (tangent) SA2[lane] = 1.0f;           // row 0 of A2 (d_L = 1)
#pragma unroll
      for (int i = 0; i < H; ++i) {
        SB1[i * kPitch + lane] = h[NH - 1][i];
        if (tangent) SB2[i * kPitch + lane] = (1.0f - h[NH - 1][i] * h[NH - 1][i]) * t[NH - 1][i];
      }
      SB1[H * kPitch + lane] = 1.0f;           // bias column
      __syncthreads();
      {
        f32x4(&accl)[1][CTH] = accL;
        // only row tile 0 of A is meaningful here (1 output row)
        mfma_outer<1, CTH>(SA1, SB1, lane, accl);
        if (tangent) mfma_outer<1, CTH>(SA2, SB2, lane, accl);
      }
      __syncthreads();
    }
    // ---- reverse sweep over hidden layers
    float hbar[H];
    {
      const float* __restrict__ WL = theta + mlp.w_off[net][NH];
#pragma unroll
      for (int i = 0; i < H; ++i) hbar[i] = alpha * WL[i];
    }
#pragma unroll
    for (int l = NH - 1; l >= 0; --l) {
      // e_l = d_l / (1-h_l^2) would lose accuracy near saturation; recompute e_l from the chain instead:
      // e_{NH-1} = W_L, e_{l} = W_{l+1}^T d_{l+1}
      float zbar[H];
      if (tangent) {
        float e[H];
        if (l == NH - 1) {
          const float* __restrict__ WL = theta + mlp.w_off[net][NH];
#pragma unroll
          for (int i = 0; i < H; ++i) e[i] = WL[i];
        } else {
          const float* __restrict__ Wn = theta + mlp.w_off[net][l + 1];
#pragma unroll
          for (int i = 0; i < H; ++i) e[i] = 0.0f;
#pragma unroll
          for (int o = 0; o < H; ++o)
#pragma unroll
            for (int i = 0; i < H; ++i) e[i] = fmaf(Wn[o * H + i], d[l + 1][o], e[i]);
        }
#pragma unroll
        for (int i = 0; i < H; ++i) hbar[i] = fmaf(-2.0f * h[l][i] * t[l][i], e[i], hbar[i]);
      }
#pragma unroll
      for (int i = 0; i < H; ++i) zbar[i] = (1.0f - h[l][i] * h[l][i]) * hbar[i];
      // stage A operands
#pragma unroll
      for (int i = 0; i < H; ++i) {
        SA1[i * kPitch + lane] = zbar[i];
        if (tangent) SA2[i * kPitch + lane] = (l == 0 ? gamma : 1.0f) * d[l][i];
      }
      if (l > 0) {
#pragma unroll
        for (int i = 0; i < H; ++i) {
          SB1[i * kPitch + lane] = h[l - 1][i];
          if (tangent) SB2[i * kPitch + lane] = (1.0f - h[l - 1][i] * h[l - 1][i]) * t[l - 1][i];
        }
        // SB1 row H is still the ones row, SB2 row H still zero
        __syncthreads();
        mfma_outer<RT, CTH>(SA1, SB1, lane, accH[l - 1]);
        if (tangent) mfma_outer<RT, CTH>(SA2, SB2, lane, accH[l - 1]);
        __syncthreads();
        // hbar_{l-1} = W_l^T zbar_l
        const float* __restrict__ Wl = theta + mlp.w_off[net][l];
        float nb[H];
#pragma unroll
        for (int i = 0; i < H; ++i) nb[i] = 0.0f;
#pragma unroll
        for (int o = 0; o < H; ++o)
#pragma unroll
          for (int i = 0; i < H; ++i) nb[i] = fmaf(Wl[o * H + i], zbar[o], nb[i]);
#pragma unroll
        for (int i = 0; i < H; ++i) hbar[i] = nb[i];
      } else {
        // first layer: B operands are the features (+ ones row) and q, read straight from
        // global memory in MFMA operand order (the tile was just streamed by the chains: L2 hits)
        __syncthreads();
#pragma unroll 2
        for (int s = 0; s < 16; ++s) {
          float a1[RT], a2[RT];
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            a1[rt] = SA1[(16 * rt + row16) * kPitch + 4 * s + kq];
            a2[rt] = tangent ? SA2[(16 * rt + row16) * kPitch + 4 * s + kq] : 0.0f;
          }
          const int fr = 4 * s + kq;
#pragma unroll
          for (int ct = 0; ct < kMaxCT1; ++ct) {
            if (ct < CT1) {
              const int i = 16 * ct + row16;
              const float b1 = i < D ? f[(int64_t)i * CVF_TILE - lane + fr] : (i == D ? 1.0f : 0.0f);
#pragma unroll
              for (int rt = 0; rt < RT; ++rt) acc1[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[rt], b1, acc1[rt][ct], 0, 0, 0);
              if (tangent) {
                const float b2 = i < D ? q[(int64_t)i * CVF_TILE - lane + fr] : 0.0f;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                  acc1[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[rt], b2, acc1[rt][ct], 0, 0, 0);
              }
            }
          }
        }
        __syncthreads();
      }
    }
  }

  // ---- flush this block's partial gradient of `net` into its slab row
  float* out = slab + (int64_t)blockIdx.x * mlp.n_params;
  const int r0 = 4 * kq;  // accumulator register r holds row 16*rt + 4*(lane>>4) + r, column 16*ct + (lane&15)
  {
    const int wo = mlp.w_off[net][0], bo = mlp.b_off[net][0];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ct = 0; ct < kMaxCT1; ++ct) {
        if (ct < CT1) {
          const int i = 16 * ct + row16;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int o = 16 * rt + r0 + r;
            if (o < H) {
              if (i < D) out[wo + o * D + i] = acc1[rt][ct][r];
              else if (i == D) out[bo + o] = acc1[rt][ct][r];
            }
          }
        }
      }
  }
#pragma unroll
  for (int l = 1; l < NH; ++l) {
    const int wo = mlp.w_off[net][l], bo = mlp.b_off[net][l];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ct = 0; ct < CTH; ++ct) {
        const int i = 16 * ct + row16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = 16 * rt + r0 + r;
          if (o < H) {
            if (i < H) out[wo + o * H + i] = accH[l - 1][rt][ct][r];
            else if (i == H) out[bo + o] = accH[l - 1][rt][ct][r];
          }
        }
      }
  }
  {
    const int wo = mlp.w_off[net][NH], bo = mlp.b_off[net][NH];
#pragma unroll
    for (int ct = 0; ct < CTH; ++ct) {
      const int i = 16 * ct + row16;
      if (kq == 0) {  // row 0 lives in register 0 of lanes 0..15
        if (i < H) out[wo + i] = accL[0][ct][0];
        else if (i == H) out[bo] = accL[0][ct][0];
      }
    }
  }
}

__global__ void slab_reduce_kernel(const float* __restrict__ slab, int64_t nblocks, int P, float* __restrict__ grad) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  float acc = 0.0f;
  for (int64_t g = 0; g < nblocks; ++g) acc += slab[g * P + p];
  grad[p] = acc;
}

bool ef_shape(const cvf_mlp_desc* m, int* H, int* NH) {
  if (m->n_layers < 2 || m->n_layers > CVF_MAX_LAYERS || m->dims[m->n_layers] != 1) return false;
  *H = m->dims[1];
  *NH = m->n_layers - 1;
  for (int l = 1; l < m->n_layers; ++l)
    if (m->dims[l] != *H) return false;
  for (int l = 0; l < m->n_layers; ++l)
    if (m->act[l] != (l + 1 < m->n_layers ? 1 : 0)) return false;
  return true;
}

int64_t bwd_grid(int64_t n_tiles) { return n_tiles < 1024 ? n_tiles : 1024; }

}  // namespace

template <class F>
static bool ef_dispatch(int H, int NH, F&& f) {
#define EF_CASE(H_, NH_)                                                        \
  if (H == H_ && NH == NH_) {                                                   \
    f(std::integral_constant<int, H_>{}, std::integral_constant<int, NH_>{});   \
    return true;                                                                \
  }
  EF_CASE(8, 1) EF_CASE(8, 2) EF_CASE(8, 3)
  EF_CASE(12, 1) EF_CASE(12, 2) EF_CASE(12, 3)
  EF_CASE(16, 1) EF_CASE(16, 2) EF_CASE(16, 3)
  EF_CASE(20, 1) EF_CASE(20, 2) EF_CASE(20, 3)
  EF_CASE(24, 2) EF_CASE(24, 3)
  EF_CASE(30, 2) EF_CASE(30, 3)
#undef EF_CASE
  return false;
}

extern "C" int cvf_ef_mlp_fwd(const cvf_mlp_desc* mlp, const float* theta, const float* feat_tiled, int64_t n_tiles,
                              float* y_tiled, float* g_tiled, void* stream) {
  CVF_REQUIRE(mlp && theta && feat_tiled && y_tiled && n_tiles > 0, "cvf_ef_mlp_fwd: bad argument");
  int H, NH;
  CVF_REQUIRE(ef_shape(mlp, &H, &NH),
              "cvf_ef_mlp_fwd: nets must be d0->H->..->H->1 with tanh between layers (got %d layers)", mlp->n_layers);
  CVF_REQUIRE(mlp->n_nets >= 1 && mlp->n_nets <= CVF_MAX_NETS, "cvf_ef_mlp_fwd: k=%d out of range", mlp->n_nets);
  dim3 grid((unsigned)n_tiles, mlp->n_nets);
  const bool launched = ef_dispatch(H, NH, [&](auto h_, auto nh_) {
    constexpr int kH = decltype(h_)::value, kNH = decltype(nh_)::value;
    hipLaunchKernelGGL((ef_fwd_kernel<kH, kNH>), grid, dim3(64), 0, (hipStream_t)stream, *mlp, theta, feat_tiled, y_tiled,
                       g_tiled);
  });
  CVF_REQUIRE(launched, "cvf_ef_mlp_fwd: no kernel instance for hidden width %d x %d layers", H, NH);
  return cvf_check_launch("ef_fwd_kernel");
}

extern "C" int64_t cvf_ef_backward_slab_floats(const cvf_mlp_desc* mlp, int64_t n_tiles) {
  return bwd_grid(n_tiles) * (int64_t)mlp->n_params;
}

extern "C" int cvf_ef_backward(const cvf_ef_cfg* cfg, const cvf_mlp_desc* mlp, const float* theta, int64_t B,
                               const float* w, const float* w_lag, const float* feat_tiled, const float* y_tiled,
                               const float* q_tiled, const double* coef, float* slab, float* grad, void* stream) {
  CVF_REQUIRE(cfg && mlp && theta && w && feat_tiled && y_tiled && coef && slab && grad && B > 0,
              "cvf_ef_backward: bad argument");
  CVF_REQUIRE(cfg->lag_idx > 0 || q_tiled, "cvf_ef_backward: generator mode needs q");
  CVF_REQUIRE(cfg->lag_idx == 0 || w_lag, "cvf_ef_backward: transfer mode needs w_lag");
  int H, NH;
  CVF_REQUIRE(ef_shape(mlp, &H, &NH), "cvf_ef_backward: unsupported net shape");
  CVF_REQUIRE(mlp->dims[0] + 1 <= 16 * kMaxCT1, "cvf_ef_backward: feature dimension %d > %d not supported yet",
              mlp->dims[0], 16 * kMaxCT1 - 1);
  EfBwdArgs a;
  a.k = cfg->k;
  a.lag_idx = cfg->lag_idx;
  a.B = B;
  a.T = cvf_ntiles(B);
  a.n_tiles = cfg->lag_idx > 0 ? 2 * a.T : a.T;
  const int64_t G = bwd_grid(a.n_tiles);
  dim3 grid((unsigned)G, cfg->k);
  const bool launched = ef_dispatch(H, NH, [&](auto h_, auto nh_) {
    constexpr int kH = decltype(h_)::value, kNH = decltype(nh_)::value;
    hipLaunchKernelGGL((ef_bwd_kernel<kH, kNH>), grid, dim3(64), 0, (hipStream_t)stream, a, *mlp, theta, w, w_lag,
                       feat_tiled, y_tiled, q_tiled, coef, slab);
  });
  CVF_REQUIRE(launched, "cvf_ef_backward: no kernel instance for hidden width %d x %d layers", H, NH);
  int rc = cvf_check_launch("ef_bwd_kernel");
  if (rc) return rc;
  const int P = mlp->n_params;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, slab, G, P, grad);
  return cvf_check_launch("slab_reduce_kernel");
}
