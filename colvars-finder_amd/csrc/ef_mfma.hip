// K4a / K4b on the matrix cores: the k eigenfunction nets (colvarsfinder.nn.EigenFunctions,
// nn.py:242-293; shape d0 -> H -> .. -> H -> 1, NH hidden layers, tanh) for one 64-frame tile.
//
// Every product of the nets is a small dense contraction [H x K] x [K x frames]; it runs on
// v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate: bitwise an fmaf chain, so the 1e-5 parity
// bar holds) with the WEIGHTS as the A operand - fetched once per wave into VGPR fragments and
// reused over the wave's 64 frames - and the activations as the B operand.
//
// Register layout ("acc layout") of a hidden vector X of width H over FT sub-tiles of 16 frames:
//   f32x4 X[RT][FT];  X[rt][ft][r] (lane: col = lane&15, q = lane>>4) = feature 4*g + q of frame
//   16*ft + col, with g = 4*rt + r the "k-group".  This is exactly the C/D layout of the MFMA
//   when row rho = 4*qq + rr of row-tile rt is assigned feature 4*(4*rt + rr) + qq, and it makes
//   register r of row-tile rt directly the B operand of k-step g = 4*rt + r of the next layer
//   (k-slot q <-> feature 4*g + q): activations never leave registers between layers, and the
//   K dimension of an H = 20 layer costs 5 k-steps, not 8.
//
// The weight gradients  W_l += sum_frames zbar_l (x) [h_{l-1}; 1] + d_l (x) [tdot_{l-1}; 0]  have
// K = frames: their operands are transposed through wave-private LDS images [feature][frame]
// (66-dword pitch, conflict-free) and accumulated across the block's tiles in an LDS image of the
// net's parameters; each block writes one slab row, rows are summed in fixed order by cvf_slab_reduce.
#include "cvf_common.hpp"
#include "cvf_adam.hpp"
#include "cvf_metric.hpp"
#include "cvf_p2p.hpp"
#include <stdlib.h>
#include <type_traits>

#include "ef_frag.hpp"

namespace {

// ------------------------------------------------------------------------------------------------
// K4a: y and g = dy/dfeat for one (sub-tile group, net)
// ------------------------------------------------------------------------------------------------
template <int H, int NH, int FT>
__global__ __launch_bounds__(64) void ef_fwd_mfma_kernel(cvf_mlp_desc mlp, const float* __restrict__ theta,
                                                          const float* __restrict__ packed,
                                                          const float* __restrict__ feat, float* __restrict__ y_tiled,
                                                          float* __restrict__ g_tiled, float* __restrict__ saved) {
  constexpr int RT = Hid<H>::RT, NG = Hid<H>::NG;
  constexpr int SUB = 4 / FT;  // blocks per 64-frame tile
  const int lane = threadIdx.x, col = lane & 15, q = lane >> 4;
  const int64_t tile = blockIdx.x / SUB;
  const int ft0 = (blockIdx.x % SUB) * FT;
  const int net = blockIdx.y;
  const int k = mlp.n_nets, D = mlp.dims[0];
  const PackLayout L = pack_layout(H, NH, D);
  const float* pk = packed + (int64_t)net * L.per_net;
  const int fo = 4 * col + ft0;  // this lane's frames are fo .. fo+FT-1 of the tile
  const float* in_lane = feat + tile * (int64_t)D * CVF_TILE + fo;

  CVF_STAMP(0);
  Vec<H, FT> h[NH];
  chain_forward<H, NH, FT>(mlp, theta, pk, L, net, in_lane, lane, h);
  CVF_STAMP(4);
  if (saved != nullptr) {   // hidden activations for the backward kernel (the layout of ef_fwd_wg_kernel's hand-off)
    float* sv = saved + (tile * k + net) * (int64_t)(NH * saved_per_vec<H>());
#pragma unroll
    for (int l = 0; l < NH; ++l) {
      if constexpr (FT == 4) save_vec<H>(sv + l * saved_per_vec<H>(), h[l], lane);
      else save_vec_half<H>(sv + l * saved_per_vec<H>(), h[l], ft0 / 2, lane);
    }
  }

  float wl[RT][4];
  load_hid_const<H>(theta + mlp.w_off[net][NH], q, wl);
  const float bL = theta[mlp.b_off[net][NH]];
  // fragments of the d-chain (W_l^T) and the first chunk of W0^T are requested now: the output reduction,
  // its cross-lane sums and the stores cover their latency
  HFrag<H> tf[NH > 1 ? NH - 1 : 1];
  if (g_tiled != nullptr) {
#pragma unroll
    for (int l = 1; l < NH; ++l) load_hfrag<H>(tf[l - 1], pk + L.th(l), lane);
  }
  {
    float yv[FT];
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) {
      float part = 0.0f;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) part = fmaf(wl[rt][r], h[NH - 1].v[rt][ft][r], part);
      yv[ft] = sum_over_q(part) + bL;
    }
    if (q == 0) store_frames<FT>(y_tiled + (tile * k + net) * CVF_TILE + fo, yv);
  }
  CVF_STAMP(5);
  if (g_tiled == nullptr) return;

  // d_{NH-1} = W_L .* (1 - h^2);  d_{l-1} = (W_l^T d_l) .* (1 - h_{l-1}^2)
  Vec<H, FT> d;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float hv = h[NH - 1].v[rt][ft][r];
        d.v[rt][ft][r] = wl[rt][r] * act_d1(mlp.act[0], hv);
      }
#pragma unroll
  for (int l = NH - 1; l >= 1; --l) {
    Vec<H, FT> e;
    init_bias<H, FT>(e, nullptr, q);
    hidden_mul<H, FT>(e, tf[l - 1], d);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ft = 0; ft < FT; ++ft)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float hv = h[l - 1].v[rt][ft][r];
          d.v[rt][ft][r] = e.v[rt][ft][r] * act_d1(mlp.act[0], hv);
        }
  }
  CVF_STAMP(6);
  // g = W0^T d_0 : rows = input features in natural order, K = H; the fragments of row tile rt+1 load
  // while row tile rt multiplies
  const float* pT0 = pk + L.t0();
  float* gout = g_tiled + (tile * k + net) * (int64_t)D * CVF_TILE + fo;
  const int CT = (D + 15) >> 4;
  float an[NG];
#pragma unroll
  for (int s = 0; s < NG; ++s) an[s] = pT0[s * 64 + lane];
  for (int rt = 0; rt < CT; ++rt) {
    float ac[NG];
#pragma unroll
    for (int s = 0; s < NG; ++s) ac[s] = an[s];
    if (rt + 1 < CT) {
#pragma unroll
      for (int s = 0; s < NG; ++s) an[s] = pT0[((rt + 1) * NG + s) * 64 + lane];
    }
    f32x4 acc[FT];
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) acc[ft] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int s = 0; s < NG; ++s) {
#pragma unroll
      for (int ft = 0; ft < FT; ++ft) acc[ft] = mfma4(ac[s], d.v[s >> 2][ft][s & 3], acc[ft]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * rt + 4 * q + r;
      if (i < D) {
        float gv[FT];
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) gv[ft] = acc[ft][r];
        store_frames<FT>(gout + (int64_t)i * CVF_TILE, gv);
      }
    }
  }
  CVF_STAMP(7);
}

// K4a for small feature dimensions (d0 <= 72: 18 k-steps, 5 output row tiles - the dipeptide-sized layers): EVERY
// global load of the kernel (all weight fragments, biases, the wave's slice of the feature tile) is issued before
// the first MFMA.  At small batch sizes the whole launch is one wave per SIMD slot, nothing else is resident to hide a
// dependent round trip (2-5k cycles under the start-up burst), and the phased version paid five of them in sequence.
template <int H, int NH, int FT>
__global__ __launch_bounds__(64) void ef_fwd_pre_kernel(cvf_mlp_desc mlp, const float* __restrict__ theta,
                                                         const float* __restrict__ packed,
                                                         const float* __restrict__ feat, float* __restrict__ y_tiled,
                                                         float* __restrict__ g_tiled) {
  constexpr int RT = Hid<H>::RT, NG = Hid<H>::NG;
  constexpr int SUB = 4 / FT, CH = 6, CTMAX = 5;
  const int lane = threadIdx.x, col = lane & 15, q = lane >> 4;
  const int64_t tile = blockIdx.x / SUB;
  const int ft0 = (blockIdx.x % SUB) * FT;
  const int net = blockIdx.y;
  const int k = mlp.n_nets, D = mlp.dims[0];
  const int S = (D + 3) >> 2, CT = (D + 15) >> 4;
  const PackLayout L = pack_layout(H, NH, D);
  const float* pk = packed + (int64_t)net * L.per_net;
  const int fo = 4 * col + ft0;
  const float* in_lane = feat + tile * (int64_t)D * CVF_TILE + fo;
  const bool want_g = g_tiled != nullptr;

  CVF_STAMP(0);
  // ---- all loads
  L0Chunk<H, FT, CH> c0, c1, c2;
  load_l0chunk<H, FT, CH>(c0, pk + L.f0(), D, S, in_lane, 0, lane);
  load_l0chunk<H, FT, CH>(c1, pk + L.f0(), D, S, in_lane, CH, lane);
  load_l0chunk<H, FT, CH>(c2, pk + L.f0(), D, S, in_lane, 2 * CH, lane);
  HConst<H> bias[NH];
#pragma unroll
  for (int l = 0; l < NH; ++l) load_hconst<H>(bias[l], theta + mlp.b_off[net][l], q);
  HFrag<H> hf[NH > 1 ? NH - 1 : 1];
#pragma unroll
  for (int l = 1; l < NH; ++l) load_hfrag<H>(hf[l - 1], pk + L.fh(l), lane);
  float wl[RT][4];
  load_hid_const<H>(theta + mlp.w_off[net][NH], q, wl);
  const float bL = theta[mlp.b_off[net][NH]];
  HFrag<H> tf[NH > 1 ? NH - 1 : 1];
  float t0[CTMAX][NG];
  if (want_g) {
#pragma unroll
    for (int l = 1; l < NH; ++l) load_hfrag<H>(tf[l - 1], pk + L.th(l), lane);
    const float* pT0 = pk + L.t0();
#pragma unroll
    for (int rt = 0; rt < CTMAX; ++rt) {
      const int rte = rt < CT ? rt : CT - 1;
#pragma unroll
      for (int s = 0; s < NG; ++s) {
        const float v = pT0[(rte * NG + s) * 64 + lane];
        t0[rt][s] = rt < CT ? v : 0.0f;
      }
    }
  }

  CVF_STAMP(1);
  // ---- compute
  Vec<H, FT> h[NH];
  set_const<H, FT>(h[0], bias[0]);
  mul_l0chunk<H, FT, CH>(h[0], c0);
  CVF_STAMP(2);
  mul_l0chunk<H, FT, CH>(h[0], c1);
  mul_l0chunk<H, FT, CH>(h[0], c2);
  CVF_STAMP(3);
  tanh_inplace<H, FT>(h[0], mlp.act[0]);
#pragma unroll
  for (int l = 1; l < NH; ++l) {
    set_const<H, FT>(h[l], bias[l]);
    hidden_mul<H, FT>(h[l], hf[l - 1], h[l - 1]);
    tanh_inplace<H, FT>(h[l], mlp.act[0]);
  }
  {
    float yv[FT];
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) {
      float part = 0.0f;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) part = fmaf(wl[rt][r], h[NH - 1].v[rt][ft][r], part);
      yv[ft] = sum_over_q(part) + bL;
    }
    if (q == 0) store_frames<FT>(y_tiled + (tile * k + net) * CVF_TILE + fo, yv);
  }
  CVF_STAMP(4);
  if (!want_g) return;
  Vec<H, FT> d;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float hv = h[NH - 1].v[rt][ft][r];
        d.v[rt][ft][r] = wl[rt][r] * act_d1(mlp.act[0], hv);
      }
#pragma unroll
  for (int l = NH - 1; l >= 1; --l) {
    Vec<H, FT> e;
    init_bias<H, FT>(e, nullptr, q);
    hidden_mul<H, FT>(e, tf[l - 1], d);
    tangent_of<H, FT>(d, h[l - 1], e, mlp.act[0]);
  }
  CVF_STAMP(5);
  float* gout = g_tiled + (tile * k + net) * (int64_t)D * CVF_TILE + fo;
#pragma unroll
  for (int rt = 0; rt < CTMAX; ++rt) {
    if (rt < CT) {  // wave-uniform
      f32x4 acc[FT];
#pragma unroll
      for (int ft = 0; ft < FT; ++ft) acc[ft] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int s = 0; s < NG; ++s)
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) acc[ft] = mfma4(t0[rt][s], d.v[s >> 2][ft][s & 3], acc[ft]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * rt + 4 * q + r;
        if (i < D) {
          float gv[FT];
#pragma unroll
          for (int ft = 0; ft < FT; ++ft) gv[ft] = acc[ft][r];
          store_frames<FT>(gout + (int64_t)i * CVF_TILE, gv);
        }
      }
    }
  }
  CVF_STAMP(6);
  CVF_STAMP(7);
}

// ------------------------------------------------------------------------------------------------------------------
// K4a + K2/K3 (+K5a) in one launch, fast layout (pure positions, contiguous align set, D <= 72): block = one 64-frame
// tile, one wave per net.  Each wave runs the forward chain of its net on the matrix cores (as ef_fwd_pre_kernel:
// every operand load up front), keeps g = dy/dfeat in ITS LDS image instead of writing it to HBM, and goes straight
// on, lane = frame, with the three passes of q = J A J^T g (cvf_metric.hpp), which read that image in place.
// Against the two separate launches this removes the 16 MB g store, the 16 MB g load (at 20 000 frames both kernels
// spend about half their time in those phases, which all waves enter together) and one launch boundary.
// ------------------------------------------------------------------------------------------------------------------
template <int H, int NH, bool K1>
__global__ __launch_bounds__(512) void ef_fwd_metric_kernel(cvf_mlp_desc mlp, const float* __restrict__ theta,
                                                             const float* __restrict__ packed,
                                                             float* __restrict__ feat, cvf_pp_desc pp,
                                                             const float* __restrict__ x, int64_t B,
                                                             float* __restrict__ aux_tiled,
                                                             const float* __restrict__ a, float* __restrict__ y_tiled,
                                                             float* __restrict__ saved, float* __restrict__ q_tiled,
                                                             float* __restrict__ e_tiled, MetricFuse fuse) {
  constexpr int RT = Hid<H>::RT, NG = Hid<H>::NG, FT = 4, CH = 6, CTMAX = 5;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x;
  const int col = lane & 15, q = lane >> 4;
  CVF_STAMP(50);
  const int64_t tile = blockIdx.x;
  const int net = wave;                         // the host launches one wave per net
  const int k = mlp.n_nets, D = mlp.dims[0];
  const int S = (D + 3) >> 2, CT = (D + 15) >> 4;
  const int nc = pp.n_coord, nal = pp.n_align;
  const int stride = x_tile_stride(nc);
  float* refL = lds + CVF_TILE * stride;                              // [3*nal]
  float* aL = refL + 3 * nal;                                         // [nc]
  float* Ub = lds + ((CVF_TILE * stride + 3 * nal + nc + 3) & ~3);   // images start 16-byte aligned
  float* Uw = Ub + (size_t)wave * nc * CVF_TILE;                      // this wave's image [nc][64]
  float* yL = Ub + (size_t)k * nc * CVF_TILE;                         // [k][64]
  const PackLayout L = pack_layout(H, NH, D);
  const float* pk = packed + (int64_t)net * L.per_net;
  const int fo = 4 * col;
  const float* in_lane = feat + tile * (int64_t)D * CVF_TILE + fo;
  float auxv[CVF_AUX_ROWS];
  float wv = 0.0f;
  if (fuse.on) {
    const int64_t frame = tile * CVF_TILE + lane;
    wv = fuse.w[frame < B ? frame : B - 1];
    if (frame >= B) wv = 0.0f;
  }
  const int ntab = 3 * nal + nc;
  float tabv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int j = tid + nthreads * i;
    const int jc = j < ntab ? j : ntab - 1;
    tabv[i] = jc < 3 * nal ? pp.ref_c[jc] : a[jc - 3 * nal];
  }
  if constexpr (!K1) {
    // the alignment kernel ran before: its rotation / centroid / K^-1 rows (lane = frame)
    const float* ax = aux_tiled + tile * CVF_AUX_ROWS * CVF_TILE + lane;
#pragma unroll
    for (int i = 0; i < CVF_AUX_ROWS; ++i) auxv[i] = ax[i * CVF_TILE];
  } else {
    // ---- K1 inside: stage the tile, split the align atoms over the block's waves for centroid + covariance (fp64
    // partial sums through the still unused g images), every wave solves the 64 rotations (lane = frame), then the
    // waves split the atoms again for the aligned positions = features, which land in image 0 (the forward chain's
    // B operand) and in feat_tiled (the backward kernel reads them)
    load_x_tile(x, B, nc, tile, lds, tid, nthreads);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int j = tid + nthreads * i;
      if (j < ntab) refL[j] = tabv[i];
    }
    for (int j = tid + 2 * nthreads; j < ntab; j += nthreads) refL[j] = j < 3 * nal ? pp.ref_c[j] : a[j - 3 * nal];
    __syncthreads();
    CVF_STAMP(51);
    const float* my = lds + lane * stride;
    const int nw = nthreads >> 6;
    // centroid + covariance over d = x - (atom 0 of the frame), fp32 (as k1_stream_kernel: the differences are
    // molecule-sized, packed products, no per-atom fp32 -> fp64 conversions); every wave over all align atoms -
    // splitting them cost more in the exchange than it saved
    const float p0 = my[0], p1 = my[1], p2 = my[2];
    f2 H01[3] = {{0, 0}, {0, 0}, {0, 0}};
    float H2[3] = {0, 0, 0}, sd[3] = {0, 0, 0};
#pragma unroll 2
    for (int b = 0; b < nal; ++b) {
      const float d0 = my[3 * b] - p0, d1 = my[3 * b + 1] - p1, d2 = my[3 * b + 2] - p2;
      const f2 r01 = f2{refL[3 * b], refL[3 * b + 1]};
      const float r2 = refL[3 * b + 2];
      sd[0] += d0; sd[1] += d1; sd[2] += d2;
      H01[0] = fma2(splat2(d0), r01, H01[0]); H2[0] = fmaf(d0, r2, H2[0]);
      H01[1] = fma2(splat2(d1), r01, H01[1]); H2[1] = fmaf(d1, r2, H2[1]);
      H01[2] = fma2(splat2(d2), r01, H01[2]); H2[2] = fmaf(d2, r2, H2[2]);
    }
    CVF_STAMP(52);
    const double inv = fast_rcp((double)nal);
    const double cr[3] = {(double)sd[0] * inv, (double)sd[1] * inv, (double)sd[2] * inv};
    double cd[3] = {(double)p0 + cr[0], (double)p1 + cr[1], (double)p2 + cr[2]};
    // (the term -(centroid - pivot) x sum(ref) is dropped: the reference is stored centred, its sum is the fp32
    // rounding residue of its mean, and against a molecule-sized factor that is 1e-8 of H - below the fp32 sums)
    double Hm[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      Hm[i][0] = (double)H01[i].x;
      Hm[i][1] = (double)H01[i].y;
      Hm[i][2] = (double)H2[i];
    }
    KabschOut ko;
    kabsch_from_H(Hm, ko);
    const Centre c = centre_of(cd);
    CVF_STAMP(53);
#pragma unroll
    for (int i = 0; i < 9; ++i) auxv[i] = ko.R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) auxv[9 + i] = c.hi[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) auxv[12 + i] = ko.Kinv[i];
    if (wave == 0 && aux_tiled != nullptr) {
      float* ax = aux_tiled + tile * CVF_AUX_ROWS * CVF_TILE + lane;
#pragma unroll
      for (int i = 0; i < CVF_AUX_ROWS; ++i) ax[i * CVF_TILE] = auxv[i];
    }
    CVF_STAMP(60);
    float* ft = feat + tile * (int64_t)D * CVF_TILE + lane;
#pragma unroll 2
    for (int at = wave; at < pp.n_rec; at += nw) {
      const V3 al = row_times(centred(my, at, c), ko.R);
      Ub[(3 * at) * CVF_TILE + lane] = al.x;
      Ub[(3 * at + 1) * CVF_TILE + lane] = al.y;
      Ub[(3 * at + 2) * CVF_TILE + lane] = al.z;
      ft[(3 * at) * CVF_TILE] = al.x;
      ft[(3 * at + 1) * CVF_TILE] = al.y;
      ft[(3 * at + 2) * CVF_TILE] = al.z;
    }
    CVF_STAMP(61);
    lds_barrier();   // the feature image is complete
    CVF_STAMP(54);
    in_lane = Ub + fo;
  }
  // ---- ... and forward part (matrix-core layout): every operand of the chain
  L0Chunk<H, FT, CH> c0, c1, c2;
  load_l0chunk<H, FT, CH>(c0, pk + L.f0(), D, S, in_lane, 0, lane);
  load_l0chunk<H, FT, CH>(c1, pk + L.f0(), D, S, in_lane, CH, lane);
  load_l0chunk<H, FT, CH>(c2, pk + L.f0(), D, S, in_lane, 2 * CH, lane);
  HConst<H> bias[NH];
#pragma unroll
  for (int l = 0; l < NH; ++l) load_hconst<H>(bias[l], theta + mlp.b_off[net][l], q);
  HFrag<H> hf[NH > 1 ? NH - 1 : 1];
#pragma unroll
  for (int l = 1; l < NH; ++l) load_hfrag<H>(hf[l - 1], pk + L.fh(l), lane);
  float wl[RT][4];
  load_hid_const<H>(theta + mlp.w_off[net][NH], q, wl);
  const float bL = theta[mlp.b_off[net][NH]];
  if constexpr (!K1) {
    load_x_tile(x, B, nc, tile, lds, tid, nthreads);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int j = tid + nthreads * i;
      if (j < ntab) refL[j] = tabv[i];
    }
    for (int j = tid + 2 * nthreads; j < ntab; j += nthreads) refL[j] = j < 3 * nal ? pp.ref_c[j] : a[j - 3 * nal];
  }
  CVF_STAMP(55);
  // ---- forward chain
  Vec<H, FT> h[NH];
  set_const<H, FT>(h[0], bias[0]);
  mul_l0chunk<H, FT, CH>(h[0], c0);
  mul_l0chunk<H, FT, CH>(h[0], c1);
  mul_l0chunk<H, FT, CH>(h[0], c2);
  CVF_STAMP(56);
  tanh_inplace<H, FT>(h[0], mlp.act[0]);
#pragma unroll
  for (int l = 1; l < NH; ++l) {
    set_const<H, FT>(h[l], bias[l]);
    hidden_mul<H, FT>(h[l], hf[l - 1], h[l - 1]);
    tanh_inplace<H, FT>(h[l], mlp.act[0]);
  }
  {
    float yv4[FT];
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) {
      float part = 0.0f;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) part = fmaf(wl[rt][r], h[NH - 1].v[rt][ft][r], part);
      yv4[ft] = sum_over_q(part) + bL;
    }
    if (q == 0) {
      store_frames<FT>(y_tiled + (tile * k + net) * CVF_TILE + fo, yv4);
      *reinterpret_cast<float4*>(yL + net * CVF_TILE + fo) = float4{yv4[0], yv4[1], yv4[2], yv4[3]};
    }
  }
  if (saved != nullptr) {
    float* sv = saved + (tile * k + net) * (int64_t)(NH * saved_per_vec<H>());
#pragma unroll
    for (int l = 0; l < NH; ++l) save_vec<H>(sv + l * saved_per_vec<H>(), h[l], lane);
  }
  CVF_STAMP(57);
  // ---- d chain and g = W_1^T d_1 -> this wave's LDS image [feature][frame]
  {
    HFrag<H> tf[NH > 1 ? NH - 1 : 1];
#pragma unroll
    for (int l = 1; l < NH; ++l) load_hfrag<H>(tf[l - 1], pk + L.th(l), lane);
    Vec<H, FT> d;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ft = 0; ft < FT; ++ft)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float hv = h[NH - 1].v[rt][ft][r];
          d.v[rt][ft][r] = wl[rt][r] * act_d1(mlp.act[0], hv);
        }
#pragma unroll
    for (int l = NH - 1; l >= 1; --l) {
      Vec<H, FT> e;
      init_bias<H, FT>(e, nullptr, q);
      hidden_mul<H, FT>(e, tf[l - 1], d);
      tangent_of<H, FT>(d, h[l - 1], e, mlp.act[0]);
    }
    const float* pT0 = pk + L.t0();
    float t0[CTMAX][NG];   // requested before the barrier: five dependent round trips otherwise
#pragma unroll
    for (int rt = 0; rt < CTMAX; ++rt)
#pragma unroll
      for (int s = 0; s < NG; ++s) t0[rt][s] = pT0[((rt < CT ? rt : CT - 1) * NG + s) * 64 + lane];
    CVF_STAMP(58);
    if constexpr (K1) lds_barrier();   // every wave has consumed the feature image: wave 0's g may replace it
#pragma unroll
    for (int rt = 0; rt < CTMAX; ++rt) {
      if (rt < CT) {  // wave-uniform
        f32x4 acc[FT];
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) acc[ft] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int s = 0; s < NG; ++s)
#pragma unroll
          for (int ft = 0; ft < FT; ++ft) acc[ft] = mfma4(t0[rt][s], d.v[s >> 2][ft][s & 3], acc[ft]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * rt + 4 * q + r;
          if (i < D) *reinterpret_cast<float4*>(Uw + i * CVF_TILE + fo) = float4{acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
        }
      }
    }
  }
  CVF_STAMP(59);
  lds_barrier();   // coordinate tile, tables and every net's y are in LDS; the g images are wave-private
  float yv[CVF_MAX_NETS];
#pragma unroll
  for (int j = 0; j < CVF_MAX_NETS; ++j) yv[j] = yL[(j < k ? j : k - 1) * CVF_TILE + lane];
  float cur[3 * kGChunk];
#pragma unroll
  for (int i = 0; i < 3 * kGChunk; ++i) cur[i] = 0.0f;
  float* qt = q_tiled + (tile * k + net) * (int64_t)pp.d_r * CVF_TILE + lane;
  metric_pure_passes<true>(pp, lane, net, k, tile, B, lds + lane * stride, refL, aL, Uw + lane, auxv, nullptr, qt, e_tiled, fuse, wv,
                           yv, cur);
}

// ------------------------------------------------------------------------------------------------------------------
// K1 + K4a in one launch for the transfer-operator mode (core.py:403,414: y = model(pp_layer(X)), y' on the lagged frames):
// the alignment part and the forward part of ef_fwd_metric_kernel without its derivative part.  Tiles 0..T1-1 take their
// frames from x, tiles T1.. from x2 (the lagged frames); block = tile, one wave per net; LDS = coordinate tile + the
// feature image only (34 KB at 22 atoms: four workgroups per CU, 2T tiles in one round).  Replaces two alignment launches
// and the 2T-tile forward launch (12 + 12 + 22 us at 20 000 frames) and the feature round trip between them.
// ------------------------------------------------------------------------------------------------------------------
template <int H, int NH>
__global__ __launch_bounds__(512) void ef_align_fwd_kernel(cvf_mlp_desc mlp, const float* __restrict__ theta,
                                                            const float* __restrict__ packed, float* __restrict__ feat,
                                                            cvf_pp_desc pp, const float* __restrict__ x,
                                                            const float* __restrict__ x2, int64_t T1, int64_t B,
                                                            float* __restrict__ y_tiled, float* __restrict__ saved) {
  constexpr int RT = Hid<H>::RT, FT = 4, CH = 6;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x;
  const int col = lane & 15, q = lane >> 4;
  const int64_t tile = blockIdx.x;
  const int net = wave;
  const int k = mlp.n_nets, D = mlp.dims[0];
  const int S = (D + 3) >> 2;
  const int nc = pp.n_coord, nal = pp.n_align;
  const int stride = x_tile_stride(nc);
  float* refL = lds + CVF_TILE * stride;                              // [3*nal]
  float* Ub = lds + ((CVF_TILE * stride + 3 * nal + 3) & ~3);         // feature image [D][64], 16-byte aligned
  const PackLayout L = pack_layout(H, NH, D);
  const float* pk = packed + (int64_t)net * L.per_net;
  const int fo = 4 * col;
  const bool second = tile >= T1;
  load_x_tile(second ? x2 : x, B, nc, second ? tile - T1 : tile, lds, tid, nthreads);
  for (int j = tid; j < 3 * nal; j += nthreads) refL[j] = pp.ref_c[j];
  __syncthreads();
  const float* my = lds + lane * stride;
  const int nw = nthreads >> 6;
  // centroid + covariance over d = x - (atom 0 of the frame), packed fp32 (as ef_fwd_metric_kernel)
  const float p0 = my[0], p1 = my[1], p2 = my[2];
  f2 H01[3] = {{0, 0}, {0, 0}, {0, 0}};
  float H2[3] = {0, 0, 0}, sd[3] = {0, 0, 0};
#pragma unroll 2
  for (int b = 0; b < nal; ++b) {
    const float d0 = my[3 * b] - p0, d1 = my[3 * b + 1] - p1, d2 = my[3 * b + 2] - p2;
    const f2 r01 = f2{refL[3 * b], refL[3 * b + 1]};
    const float r2 = refL[3 * b + 2];
    sd[0] += d0; sd[1] += d1; sd[2] += d2;
    H01[0] = fma2(splat2(d0), r01, H01[0]); H2[0] = fmaf(d0, r2, H2[0]);
    H01[1] = fma2(splat2(d1), r01, H01[1]); H2[1] = fmaf(d1, r2, H2[1]);
    H01[2] = fma2(splat2(d2), r01, H01[2]); H2[2] = fmaf(d2, r2, H2[2]);
  }
  const double inv = fast_rcp((double)nal);
  const double cd[3] = {(double)p0 + (double)sd[0] * inv, (double)p1 + (double)sd[1] * inv, (double)p2 + (double)sd[2] * inv};
  double Hm[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    Hm[i][0] = (double)H01[i].x;
    Hm[i][1] = (double)H01[i].y;
    Hm[i][2] = (double)H2[i];
  }
  KabschOut ko;
  kabsch_from_H(Hm, ko);
  const Centre c = centre_of(cd);
  float* ft = feat + tile * (int64_t)D * CVF_TILE + lane;
#pragma unroll 2
  for (int at = wave; at < pp.n_rec; at += nw) {
    const V3 al = row_times(centred(my, at, c), ko.R);
    Ub[(3 * at) * CVF_TILE + lane] = al.x;
    Ub[(3 * at + 1) * CVF_TILE + lane] = al.y;
    Ub[(3 * at + 2) * CVF_TILE + lane] = al.z;
    ft[(3 * at) * CVF_TILE] = al.x;
    ft[(3 * at + 1) * CVF_TILE] = al.y;
    ft[(3 * at + 2) * CVF_TILE] = al.z;
  }
  lds_barrier();   // the feature image is complete
  const float* in_lane = Ub + fo;
  L0Chunk<H, FT, CH> c0, c1, c2;
  load_l0chunk<H, FT, CH>(c0, pk + L.f0(), D, S, in_lane, 0, lane);
  load_l0chunk<H, FT, CH>(c1, pk + L.f0(), D, S, in_lane, CH, lane);
  load_l0chunk<H, FT, CH>(c2, pk + L.f0(), D, S, in_lane, 2 * CH, lane);
  HConst<H> bias[NH];
#pragma unroll
  for (int l = 0; l < NH; ++l) load_hconst<H>(bias[l], theta + mlp.b_off[net][l], q);
  HFrag<H> hf[NH > 1 ? NH - 1 : 1];
#pragma unroll
  for (int l = 1; l < NH; ++l) load_hfrag<H>(hf[l - 1], pk + L.fh(l), lane);
  float wl[RT][4];
  load_hid_const<H>(theta + mlp.w_off[net][NH], q, wl);
  const float bL = theta[mlp.b_off[net][NH]];
  Vec<H, FT> h[NH];
  set_const<H, FT>(h[0], bias[0]);
  mul_l0chunk<H, FT, CH>(h[0], c0);
  mul_l0chunk<H, FT, CH>(h[0], c1);
  mul_l0chunk<H, FT, CH>(h[0], c2);
  tanh_inplace<H, FT>(h[0], mlp.act[0]);
#pragma unroll
  for (int l = 1; l < NH; ++l) {
    set_const<H, FT>(h[l], bias[l]);
    hidden_mul<H, FT>(h[l], hf[l - 1], h[l - 1]);
    tanh_inplace<H, FT>(h[l], mlp.act[0]);
  }
  {
    float yv4[FT];
#pragma unroll
    for (int ft_ = 0; ft_ < FT; ++ft_) {
      float part = 0.0f;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) part = fmaf(wl[rt][r], h[NH - 1].v[rt][ft_][r], part);
      yv4[ft_] = sum_over_q(part) + bL;
    }
    if (q == 0) store_frames<FT>(y_tiled + (tile * k + net) * CVF_TILE + fo, yv4);
  }
  if (saved != nullptr) {
    float* sv = saved + (tile * k + net) * (int64_t)(NH * saved_per_vec<H>());
#pragma unroll
    for (int l = 0; l < NH; ++l) save_vec<H>(sv + l * saved_per_vec<H>(), h[l], lane);
  }
}

// K4a, workgroup form: four waves = four consecutive 64-frame tiles of ONE net.  The net's weight fragments
// (F0, Fh, Th, T0: ~25 KB for 66 -> 20 -> 20 -> 20 -> 1) are fetched ONCE per workgroup with 16-byte loads into
// LDS and read from there by all four waves; per-wave global traffic is then only its feature tile.  (The per-CU
// vector-memory path, not latency, bounded the wave-per-block kernels: every wave re-fetched the same 25 KB.)
template <int H, int NH>
__global__ __launch_bounds__(256) void ef_fwd_wg_kernel(cvf_mlp_desc mlp, const float* __restrict__ theta,
                                                         const float* __restrict__ packed,
                                                         const float* __restrict__ feat, int64_t n_tiles,
                                                         float* __restrict__ y_tiled, float* __restrict__ g_tiled,
                                                         float* __restrict__ saved) {
  constexpr int RT = Hid<H>::RT, NG = Hid<H>::NG, FT = 4, CH = 6;
  extern __shared__ float wL[];   // this net's packed fragments
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, q = lane >> 4;
  const int net = blockIdx.y;
  const int k = mlp.n_nets, D = mlp.dims[0];
  const int S = (D + 3) >> 2, CT = (D + 15) >> 4;
  const PackLayout L = pack_layout(H, NH, D);
  {
    const float4* src = reinterpret_cast<const float4*>(packed + (int64_t)net * L.per_net);
    float4* dst = reinterpret_cast<float4*>(wL);
    for (int i = tid; i < L.per_net / 4; i += 256) dst[i] = src[i];
  }
  int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  const bool live = tile < n_tiles;
  if (!live) tile = n_tiles - 1;   // keep the wave in step with the barriers; its stores are masked
  const int fo = 4 * col;
  const float* in_lane = feat + tile * (int64_t)D * CVF_TILE + fo;
  const bool want_g = g_tiled != nullptr;
  // this wave's global loads: feature tile slice (all k-steps), biases, last-layer weights
  float bfr[18][FT];
  const bool smallD = S <= 18;
#pragma unroll
  for (int s = 0; s < 18; ++s) {
    const int se = s < S ? s : S - 1;
    const int kf = 4 * se + q;
    load_frames<FT>(in_lane + (int64_t)(kf < D ? kf : 0) * CVF_TILE, bfr[s]);
  }
  HConst<H> bias[NH];
#pragma unroll
  for (int l = 0; l < NH; ++l) load_hconst<H>(bias[l], theta + mlp.b_off[net][l], q);
  float wl[RT][4];
  load_hid_const<H>(theta + mlp.w_off[net][NH], q, wl);
  const float bL = theta[mlp.b_off[net][NH]];
  __syncthreads();

  Vec<H, FT> h[NH];
  set_const<H, FT>(h[0], bias[0]);
  {
    const float* f0 = wL + L.f0();
#pragma unroll
    for (int s = 0; s < 18; ++s) {
      if (s < S) {  // wave-uniform
#pragma unroll
        for (int ft = 0; ft < FT; ++ft)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) h[0].v[rt][ft] = mfma4(f0[(s * RT + rt) * 64 + lane], bfr[s][ft], h[0].v[rt][ft]);
      }
    }
    for (int s = 18; s < S; ++s) {  // wider first layers: remaining k-steps straight from global
      const int kf = 4 * s + q;
      float b[FT];
      load_frames<FT>(in_lane + (int64_t)(kf < D ? kf : 0) * CVF_TILE, b);
#pragma unroll
      for (int ft = 0; ft < FT; ++ft)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) h[0].v[rt][ft] = mfma4(f0[(s * RT + rt) * 64 + lane], b[ft], h[0].v[rt][ft]);
    }
  }
  (void)smallD;
  tanh_inplace<H, FT>(h[0], mlp.act[0]);
#pragma unroll
  for (int l = 1; l < NH; ++l) {
    set_const<H, FT>(h[l], bias[l]);
    hidden_apply<H, FT>(h[l], wL + L.fh(l), h[l - 1], lane);
    tanh_inplace<H, FT>(h[l], mlp.act[0]);
  }
  {
    float yv[FT];
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) {
      float part = 0.0f;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) part = fmaf(wl[rt][r], h[NH - 1].v[rt][ft][r], part);
      yv[ft] = sum_over_q(part) + bL;
    }
    if (q == 0 && live) store_frames<FT>(y_tiled + (tile * k + net) * CVF_TILE + fo, yv);
  }
  float* sv = saved != nullptr && live ? saved + (tile * k + net) * (int64_t)(NH * saved_per_vec<H>()) : nullptr;
  if (sv != nullptr) {
#pragma unroll
    for (int l = 0; l < NH; ++l) save_vec<H>(sv + l * saved_per_vec<H>(), h[l], lane);
  }
  if (!want_g) return;
  Vec<H, FT> d;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float hv = h[NH - 1].v[rt][ft][r];
        d.v[rt][ft][r] = wl[rt][r] * act_d1(mlp.act[0], hv);
      }
#pragma unroll
  for (int l = NH - 1; l >= 1; --l) {
    Vec<H, FT> e;
    init_bias<H, FT>(e, nullptr, q);
    hidden_apply<H, FT>(e, wL + L.th(l), d, lane);
    tangent_of<H, FT>(d, h[l - 1], e, mlp.act[0]);
  }
  const float* pT0 = wL + L.t0();
  float* gout = g_tiled + (tile * k + net) * (int64_t)D * CVF_TILE + fo;
  for (int rt = 0; rt < CT; ++rt) {
    f32x4 acc[FT];
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) acc[ft] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int s = 0; s < NG; ++s) {
      const float a = pT0[(rt * NG + s) * 64 + lane];
#pragma unroll
      for (int ft = 0; ft < FT; ++ft) acc[ft] = mfma4(a, d.v[s >> 2][ft][s & 3], acc[ft]);
    }
    if (live) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * rt + 4 * q + r;
        if (i < D) {
          float gv[FT];
#pragma unroll
          for (int ft = 0; ft < FT; ++ft) gv[ft] = acc[ft][r];
          store_frames<FT>(gout + (int64_t)i * CVF_TILE, gv);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Wide first layers (d0 > kWideD: the large-molecule shapes, e.g. 384 features)
// ------------------------------------------------------------------------------------------------
// X += W0 [H x D] * in for one 64-frame tile, the k-steps shared out over the kWW waves of the block: a wave's operands
// (12 k-steps = 36 loads at d0 = 384) are requested in one go, the partial sums meet in LDS and are added in wave order.
// One wave per (half) tile walked the 96 k-steps of d0 = 384 behind a two-chunk operand ring - 16 dependent memory round
// trips, 45 k of the forward kernel's 79 k cycles; a wave cannot keep more than 63 loads in flight, so a deeper ring does
// not help, more waves do.  On return tot[((rt * 4 + ft) * 4 + r) * 64 + lane] holds element (rt, ft, r) of the sum (all 64 frames).
constexpr int kWW = 8;
template <int H>
__device__ __forceinline__ void wide_layer0(const float* __restrict__ pk0, int D, const float* __restrict__ in_lane, int lane, int wave,
                                            float* part, float* tot) {
  constexpr int RT = Hid<H>::RT, CH = 12, NV = RT * 16;
  const int S = (D + 3) >> 2;
  const int per = (S + kWW - 1) / kWW;
  const int s_end = (wave + 1) * per < S ? (wave + 1) * per : S;
  Vec<H, 4> P;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) P.v[rt][ft] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  for (int s0 = wave * per; s0 < s_end; s0 += CH) {
    L0Chunk<H, 4, CH> c;
    load_l0chunk<H, 4, CH>(c, pk0, D, s_end, in_lane, s0, lane);   // (steps past s_end: clamped loads, zeroed weights)
    __builtin_amdgcn_sched_barrier(0);   // all requests first: left alone, the scheduler sinks every load to its use (one round trip per k-step)
    mul_l0chunk<H, 4, CH>(P, c);
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[(wave * NV + (rt * 4 + ft) * 4 + r) * 64 + lane] = P.v[rt][ft][r];
  __syncthreads();
  for (int i = wave; i < NV; i += kWW) {
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < kWW; ++j) acc += part[(j * NV + i) * 64 + lane];
    tot[i * 64 + lane] = acc;
  }
  __syncthreads();
}
template <int H>
constexpr size_t wide_lds_bytes() { return (size_t)(kWW + 1) * Hid<H>::RT * 16 * 64 * sizeof(float); }

// K4a for wide first layers: one block of kWW waves per (tile, net).  First layer by wide_layer0 (all waves); the short rest of the
// chain (hidden layers, y, the d chain) runs ONCE, waves 0..3 each on one 16-frame group of the tile (as eight copies on all 64
// frames it was 13 k of the block's 31 k cycles and 96 registers of activations per wave); d_0 goes through LDS to all waves, which
// share out the rows of g = W0^T d_0 (24 row tiles at d0 = 384).  Hands the hidden activations to the backward kernel like
// ef_fwd_wg_kernel.
template <int H, int NH>
__global__ __launch_bounds__(64 * kWW) void ef_fwd_wide_kernel(cvf_mlp_desc mlp, const float* __restrict__ theta,
                                                               const float* __restrict__ packed,
                                                               const float* __restrict__ feat, float* __restrict__ y_tiled,
                                                               float* __restrict__ g_tiled, float* __restrict__ saved) {
  constexpr int RT = Hid<H>::RT, NG = Hid<H>::NG, FT = 4, kGT = 3;
  extern __shared__ float wide_lds[];
  float* part = wide_lds;
  float* tot = wide_lds + kWW * RT * 16 * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, q = lane >> 4;
  const int64_t tile = blockIdx.x;
  const int net = blockIdx.y;
  const int k = mlp.n_nets, D = mlp.dims[0];
  const int CT = (D + 15) >> 4;
  const PackLayout L = pack_layout(H, NH, D);
  const float* pk = packed + (int64_t)net * L.per_net;
  const int fo = 4 * col;
  const float* in_lane = feat + tile * (int64_t)D * CVF_TILE + fo;
  const bool want_g = g_tiled != nullptr;
  const bool chain = wave < 4;   // (uniform per wave) this wave runs the chain for frame group ft = wave: frames 4 col + ft
  CVF_STAMP(0);
  // everything the rest of the chain needs is requested before the first layer
  HConst<H> bias[NH];
  HFrag<H> hf[NH > 1 ? NH - 1 : 1], tf[NH > 1 ? NH - 1 : 1];
  float wl[RT][4];
  float bL = 0.0f;
  if (chain) {
#pragma unroll
    for (int l = 0; l < NH; ++l) load_hconst<H>(bias[l], theta + mlp.b_off[net][l], q);
#pragma unroll
    for (int l = 1; l < NH; ++l) load_hfrag<H>(hf[l - 1], pk + L.fh(l), lane);
    load_hid_const<H>(theta + mlp.w_off[net][NH], q, wl);
    bL = theta[mlp.b_off[net][NH]];
    if (want_g) {
#pragma unroll
      for (int l = 1; l < NH; ++l) load_hfrag<H>(tf[l - 1], pk + L.th(l), lane);
    }
  }
  const float* pT0 = pk + L.t0();
  float t0f[kGT][NG];   // W0^T fragments of this wave's first kGT row tiles of g
  if (want_g) {
#pragma unroll
    for (int i = 0; i < kGT; ++i) {
      const int rt = wave + kWW * i < CT ? wave + kWW * i : CT - 1;
#pragma unroll
      for (int s = 0; s < NG; ++s) t0f[i][s] = pT0[(rt * NG + s) * 64 + lane];
    }
  }
  CVF_STAMP(1);
  wide_layer0<H>(pk + L.f0(), D, in_lane, lane, wave, part, tot);
  CVF_STAMP(2);
  if (chain) {
    const int ft = wave;
    Vec<H, 1> h[NH];
    set_const<H, 1>(h[0], bias[0]);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) h[0].v[rt][0][r] += tot[((rt * 4 + ft) * 4 + r) * 64 + lane];
    tanh_inplace<H, 1>(h[0], mlp.act[0]);
#pragma unroll
    for (int l = 1; l < NH; ++l) {
      set_const<H, 1>(h[l], bias[l]);
      hidden_mul<H, 1>(h[l], hf[l - 1], h[l - 1]);
      tanh_inplace<H, 1>(h[l], mlp.act[0]);
    }
    CVF_STAMP(3);
    {
      float p = 0.0f;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) p = fmaf(wl[rt][r], h[NH - 1].v[rt][0][r], p);
      const float yv = sum_over_q(p) + bL;
      if (q == 0) y_tiled[(tile * k + net) * CVF_TILE + fo + ft] = yv;
    }
    if (saved != nullptr) {   // the layout of save_vec (ef_frag.hpp), this wave's frame group
      float* sv = saved + (tile * k + net) * (int64_t)(NH * saved_per_vec<H>());
#pragma unroll
      for (int l = 0; l < NH; ++l)
#pragma unroll
        for (int g = 0; g < NG; ++g) sv[l * saved_per_vec<H>() + (g * 2 + (ft >> 1)) * 128 + 2 * lane + (ft & 1)] = h[l].v[g >> 2][0][g & 3];
    }
    CVF_STAMP(4);
    if (want_g) {
      Vec<H, 1> d;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float hv = h[NH - 1].v[rt][0][r];
          d.v[rt][0][r] = wl[rt][r] * act_d1(mlp.act[0], hv);
        }
#pragma unroll
      for (int l = NH - 1; l >= 1; --l) {
        Vec<H, 1> e;
        init_bias<H, 1>(e, nullptr, q);
        hidden_mul<H, 1>(e, tf[l - 1], d);
        tangent_of<H, 1>(d, h[l - 1], e, mlp.act[0]);
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[((rt * 4 + ft) * 4 + r) * 64 + lane] = d.v[rt][0][r];   // (part: free since wide_layer0's sums)
    }
  }
  if (!want_g) return;
  __syncthreads();
  CVF_STAMP(5);
  Vec<H, FT> d;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int r = 0; r < 4; ++r) d.v[rt][ft][r] = part[((rt * 4 + ft) * 4 + r) * 64 + lane];
  float* gout = g_tiled + (tile * k + net) * (int64_t)D * CVF_TILE + fo;
  auto g_tile = [&](int rt, const float (&af)[NG]) {
    f32x4 acc[FT];
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) acc[ft] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int s = 0; s < NG; ++s)
#pragma unroll
      for (int ft = 0; ft < FT; ++ft) acc[ft] = mfma4(af[s], d.v[s >> 2][ft][s & 3], acc[ft]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * rt + 4 * q + r;
      if (i < D) {
        float gv[FT];
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) gv[ft] = acc[ft][r];
        store_frames<FT>(gout + (int64_t)i * CVF_TILE, gv);
      }
    }
  };
#pragma unroll
  for (int i = 0; i < kGT; ++i)
    if (wave + kWW * i < CT) g_tile(wave + kWW * i, t0f[i]);   // (wave-uniform)
  for (int rt = wave + kWW * kGT; rt < CT; rt += kWW) {   // wider still: the remaining row tiles just in time
    float af[NG];
#pragma unroll
    for (int s = 0; s < NG; ++s) af[s] = pT0[(rt * NG + s) * 64 + lane];
    g_tile(rt, af);
  }
  CVF_STAMP(6);
  CVF_STAMP(7);
}

// Generator mode: the tangent chain's first product t0 = W0 q (K = d0) for one (tile, net), ahead of the backward kernel -
// there it was a just-in-time double-buffered loop (40 k cycles at d0 = 384, and the kernel has no registers left for more
// operands in flight), repeated by every block that shares the (tile, net).  Output: one vector in the hand-off layout, behind
// the activations (see cvf_ef_saved_floats).
template <int H>
__global__ __launch_bounds__(64 * kWW) void ef_t0_kernel(cvf_mlp_desc mlp, const float* __restrict__ packed,
                                                         const float* __restrict__ q_tiled, float* __restrict__ t0_out) {
  extern __shared__ float wide_lds[];
  constexpr int RT = Hid<H>::RT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, q = lane >> 4;
  const int64_t tile = blockIdx.x;
  const int net = blockIdx.y;
  const int k = mlp.n_nets, D = mlp.dims[0];
  const PackLayout L = pack_layout(H, mlp.n_layers - 1, D);
  const float* pk = packed + (int64_t)net * L.per_net;
  const float* in_lane = q_tiled + (tile * k + net) * (int64_t)D * CVF_TILE + 4 * col;
  float* tot = wide_lds + kWW * RT * 16 * 64;
  wide_layer0<H>(pk + L.f0(), D, in_lane, lane, wave, wide_lds, tot);
  if (wave == 0) {
    Vec<H, 4> t;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int r = 0; r < 4; ++r) t.v[rt][ft][r] = tot[((rt * 4 + ft) * 4 + r) * 64 + lane];
    save_vec<H>(t0_out + (tile * k + net) * (int64_t)saved_per_vec<H>(), t, lane);
  }
}

// ------------------------------------------------------------------------------------------------
// K4b
// ------------------------------------------------------------------------------------------------
struct EfBwdArgs {
  int k;
  int lag_idx;
  int64_t B;
  int64_t T;
  int64_t n_tiles;
  int zs;             // blocks that share a (tile, net): block z takes the first layer's column tiles ct = z (mod zs) and the layers l = z (mod zs)
  int direct0;        // one tile per block (grid.x == n_tiles): first-layer gradient tiles go straight to the slab row, no LDS image of W0
  const float* t0;    // wide first layers: t0 = W0 q of every (tile, net), computed by ef_t0_kernel (else NULL)
};

// K4b.  Block = WPB waves sharing one 64-frame tile (and its LDS images); wave w owns the frames
// 4*col + w*FT .. + FT-1 (FT = 4/WPB) for the register-resident chains, and every WPB-th 16x16 tile of
// each weight-gradient product.  Splitting the tile over waves shortens each wave's dependent chain and
// puts two waves on every SIMD, which is what hides the LDS / L2 latencies at small batch sizes.
// WPB = 4 (cvf_ef16_backward): wave w owns the 16 CONSECUTIVE frames 16 w .. 16 w + 15 of the tile - the unit whose activations
// the 16-frames-per-wave front kernel (ef16_front.hip, ef16_back.hip) left behind (SAVED = 2: its hand-off layout) - a quarter of the dependent
// chain per wave, four blocks of four waves per CU.
template <int H, int NH, int WPB, int SAVED>
__global__ __launch_bounds__(64 * WPB, WPB == 4 ? 3 : WPB) void ef_bwd_mfma_kernel(EfBwdArgs args, cvf_mlp_desc mlp,
                                                               const float* __restrict__ theta,
                                                               const float* __restrict__ packed,
                                                               const float* __restrict__ w, const float* __restrict__ w_lag,
                                                               const float* __restrict__ feat,
                                                               const float* __restrict__ y_tiled,
                                                               const float* __restrict__ q_tiled,
                                                               const double* __restrict__ coef, float* __restrict__ slab,
                                                               int32_t* __restrict__ step, const float* __restrict__ saved) {
  constexpr int FT = 4 / WPB;
  constexpr int RT = Hid<H>::RT;
  constexpr int RTO = (H + 15) / 16;      // row tiles of an H-row image (natural order)
  constexpr int CTH = (H + 1 + 15) / 16;  // column tiles of [h ; 1]
  constexpr int NT = 64 * WPB;
  // Operand images, packed: an image owns H (A side) or H+1 (B side) rows; the MFMA reads up to the next
  // multiple of 16 rows, i.e. into the following image - finite values that only reach output rows/columns
  // which are discarded.  Keeps the block under 40 KiB of LDS (4 blocks per CU).
  constexpr int kRows = 2 * H + 2 * (H + 1) + 16;
  __shared__ __attribute__((aligned(16))) float IMG[kRows * kPitch];
  extern __shared__ float GI[];  // this block's partial gradient of `net` (flat parameter order)
  float* SA1 = IMG;
  float* SA2 = SA1 + H * kPitch;
  float* SB1 = SA2 + H * kPitch;
  float* SB2 = SB1 + (H + 1) * kPitch;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, q = lane >> 4, row16 = col, r0 = 4 * q;
  const int ft0 = wave * FT, fo = WPB == 4 ? 16 * wave + col : 4 * col + ft0;   // first frame (of the tile) this lane owns
  const int net = blockIdx.y;
  const int z = blockIdx.z, zs = args.zs;   // (zs > 1: wide first layers - this block adds only ITS column tiles of layer 0, block 0 everything else)
  const int k = args.k;
  const int D = mlp.dims[0];
  const int CT1 = (D + 1 + 15) / 16;
  const bool tangent = args.lag_idx == 0;
  const int gbase = mlp.w_off[net][0];
  const int gspan = mlp.b_off[net][NH] + 1 - gbase;
  // direct0: the image starts behind the first layer (W0, b0 lead a net's run - checked by the host); GIs is indexed like GI
  const int gi_off = args.direct0 ? mlp.b_off[net][0] - gbase + H : 0;
  float* GIs = GI - gi_off;
  float* out0 = slab + (int64_t)blockIdx.x * mlp.n_params + gbase;   // this block's slab row, this net's run

  for (int i = tid; i < kRows * kPitch; i += NT) IMG[i] = 0.0f;
  for (int i = tid; i < gspan - gi_off; i += NT) GI[i] = 0.0f;
  __syncthreads();
  if (tid < 64) SB1[H * kPitch + tid] = 1.0f;  // bias column of [h ; 1]  (row H of SB2 stays 0)

  const double* gS1 = coef;
  const double* gS2 = coef + k;
  const double* gEt = coef + k + k * k;
  const double* gS1l = coef + 2 * k + k * k;
  const double* gS2l = coef + 3 * k + k * k;

  // this net's coefficients d loss / d sums (wave-uniform, loaded once)
  const double cS1 = coef[net], cEt = coef[k + k * k + net];
  const double cS1l = coef[2 * k + k * k + net], cS2l = coef[3 * k + k * k + net];
  double cS2[CVF_MAX_NETS];
#pragma unroll
  for (int j = 0; j < CVF_MAX_NETS; ++j) cS2[j] = j < k ? (j == net ? 2.0 : 1.0) * coef[k + net * k + j] : 0.0;

  float wl[RT][4];
  load_hid_const<H>(theta + mlp.w_off[net][NH], q, wl);
  const PackLayout L = pack_layout(H, NH, D);
  const float* pk = packed + (int64_t)net * L.per_net;
  float one[FT];
#pragma unroll
  for (int ft = 0; ft < FT; ++ft) one[ft] = 1.0f;

  // add a finished 16x16 tile of layer `l` (rows = outputs, columns = inputs + bias) into the gradient image
  auto add_tile = [&](int l, int n_out, int n_in, int rt, int ct, const f32x4& acc) {
    const int wo = mlp.w_off[net][l] - gbase, bo = mlp.b_off[net][l] - gbase;
    const int i = 16 * ct + row16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = 16 * rt + r0 + r;
      if (o < n_out) {
        if (i < n_in) GIs[wo + o * n_in + i] += acc[r];
        else if (i == n_in) GIs[bo + o] += acc[r];
      }
    }
  };
  // first layer, one tile per block: the finished tile is this block's whole contribution - straight to the slab row
  auto put_tile0 = [&](int rt, int ct, const f32x4& acc) {
    const int wo = mlp.w_off[net][0] - gbase, bo = mlp.b_off[net][0] - gbase;
    const int i = 16 * ct + row16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = 16 * rt + r0 + r;
      if (o < H) {
        if (i < D) out0[wo + o * D + i] = acc[r];
        else if (i == D) out0[bo + o] = acc[r];
      }
    }
  };

  for (int64_t tile = blockIdx.x; tile < args.n_tiles; tile += gridDim.x) {
    const int pass = tile >= args.T ? 1 : 0;
    const int64_t t0 = pass ? tile - args.T : tile;
    CVF_STAMP(8);
    // ---- per-frame coefficients for the FT frames this lane owns
    float alpha[FT], gamma[FT];
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) {
      const int64_t frame = t0 * CVF_TILE + fo + ft;
      const bool valid = frame < args.B;
      const int64_t fc = valid ? frame : args.B - 1;
      const float wraw = w[fc];
      const float wb = valid ? wraw : 0.0f;
      const float* yb = y_tiled + t0 * k * CVF_TILE + fo + ft;
      gamma[ft] = 0.0f;
      if (args.lag_idx == 0) {
        double a = gS1[net];
        for (int j = 0; j < k; ++j) a += (j == net ? 2.0 : 1.0) * gS2[net * k + j] * (double)yb[j * CVF_TILE];
        alpha[ft] = (float)((double)wb * a);
        gamma[ft] = (float)(2.0 * (double)wb * gEt[net]);
      } else {
        const float* yl = y_tiled + (args.T + t0) * k * CVF_TILE + fo + ft;
        const double diff = (double)yl[net * CVF_TILE] - (double)yb[net * CVF_TILE];
        const double tterm = 2.0 * (double)wb * gEt[net] * diff;
        if (pass == 0) {
          double a = gS1[net];
          for (int j = 0; j < k; ++j) a += (j == net ? 2.0 : 1.0) * gS2[net * k + j] * (double)yb[j * CVF_TILE];
          alpha[ft] = (float)((double)wb * a - tterm);
        } else {
          const float wlraw = w_lag[fc];
          const float wlg = valid ? wlraw : 0.0f;
          alpha[ft] = (float)((double)wlg * (gS1l[net] + 2.0 * gS2l[net] * (double)yl[net * CVF_TILE]) + tterm);
        }
      }
    }
    const float* f_tile = feat + tile * (int64_t)D * CVF_TILE;
    const float* q_tile = tangent ? q_tiled + (tile * k + net) * (int64_t)D * CVF_TILE : nullptr;

    CVF_STAMP(9);
    // ---- chains (registers, MFMA) for this wave's frames
    Vec<H, FT> h[NH];
    Vec<H, FT> e[NH > 1 ? NH - 1 : 1];  // e[l] = W_{l+1}^T d_{l+1}, l = 0..NH-2  (e_{NH-1} = W_L is the constant wl)
    Vec<H, FT> t[NH];                   // t[l] = W_l tdot_{l-1}
    if (SAVED) {
      // the forward kernel left h_1..h_NH for this (tile, net): 15 coalesced 8-byte loads instead of recomputing the
      // forward chain (68 + 40 matrix instructions behind just-in-time weight loads).  (Handing over the d chain too
      // was measured: 10 MB more traffic each way and no change in this kernel's time.)
      static_assert(SAVED != 1 || FT == 2, "the saved layout pairs the frame groups of a two-wave block");
      static_assert(SAVED != 2 || FT == 1, "the 16-frame hand-off belongs to the four-wave block");
      const float* sv = saved + (tile * k + net) * (int64_t)(NH * saved_per_vec<H>());
      if constexpr (SAVED == 2) {   // [vector][group g][unit = wave][lane]: one coalesced 256-byte row per (vector, g)
#pragma unroll
        for (int l = 0; l < NH; ++l) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) h[l].v[rt][0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
          for (int g = 0; g < Hid<H>::NG; ++g) h[l].v[g >> 2][0][g & 3] = sv[((l * Hid<H>::NG + g) * 4 + wave) * 64 + lane];
        }
      } else if constexpr (SAVED == 1) {
#pragma unroll
        for (int l = 0; l < NH; ++l) load_vec<H>(sv + l * saved_per_vec<H>(), h[l], wave, lane);
      }
    } else {
      chain_forward<H, NH, FT, true>(mlp, theta, pk, L, net, f_tile + fo, lane, h);
    }
    CVF_STAMP(10);
    if (tangent) {
      // SAVED: the hidden layers' fragments (W^T for the d chain, W for the tangent chain) are requested here, ahead of
      // the first-layer product, instead of just in time at each layer (registers are free without the recompute)
      HFrag<H> tfr[NH > 1 ? NH - 1 : 1], ffr[NH > 1 ? NH - 1 : 1];
      if (SAVED == 1) {
#pragma unroll
        for (int l = 1; l < NH; ++l) {
          load_hfrag<H>(tfr[l - 1], pk + L.th(l), lane);
          load_hfrag<H>(ffr[l - 1], pk + L.fh(l), lane);
        }
      }
      // the tangent chain's first operands (weights, q rows) are requested before the d chain runs: one memory round trip
      // of this wave's dependent chain overlaps that chain's matrix instructions (18.3 k -> 17.0 k cycles for this phase)
      L0Chunk<H, FT, 3> tc0;
      const bool have_t0 = FT == 2 && args.t0 != nullptr;   // t0 = W0 q already computed (ef_t0_kernel)
      if (SAVED == 1 && !have_t0) load_l0chunk<H, FT, 3>(tc0, pk + L.f0(), D, (D + 3) >> 2, q_tile + fo, 0, lane);
      {
        Vec<H, FT> d;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int ft = 0; ft < FT; ++ft)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float hv = h[NH - 1].v[rt][ft][r];
              d.v[rt][ft][r] = wl[rt][r] * act_d1(mlp.act[0], hv);
            }
#pragma unroll
        for (int l = NH - 1; l >= 1; --l) {
          init_bias<H, FT>(e[l - 1], nullptr, q);
          if (SAVED == 1) hidden_mul<H, FT>(e[l - 1], tfr[l - 1], d);
          else hidden_apply<H, FT>(e[l - 1], pk + L.th(l), d, lane);
          tangent_of<H, FT>(d, h[l - 1], e[l - 1], mlp.act[0]);   // d_{l-1} = (1 - h^2) .* e_{l-1}
        }
      }
      init_bias<H, FT>(t[0], nullptr, q);
      // (batches of 4 / 6 / 9 k-steps: 20.1 / 20.6 / 22.0 k cycles for this phase against 18.3 k - spills; requesting the
      //  first layer's first column tile of B operands at the end of the layer-1 step: 73.7 k vs 69.0 k cycles in all - spills)
      if constexpr (FT == 2) {
        if (have_t0) load_vec<H>(args.t0 + (tile * k + net) * (int64_t)saved_per_vec<H>(), t[0], wave, lane);
      }
      if (have_t0) {
      } else if (SAVED == 1) layer0_apply_from<H, FT, 3>(t[0], pk + L.f0(), D, q_tile + fo, lane, tc0);
      else layer0_apply<H, FT, 3>(t[0], pk + L.f0(), D, q_tile + fo, lane);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ft = 0; ft < FT; ++ft)
#pragma unroll
          for (int r = 0; r < 4; ++r) t[0].v[rt][ft][r] *= gamma[ft];
#pragma unroll
      for (int l = 1; l < NH; ++l) {
        Vec<H, FT> td;
        tangent_of<H, FT>(td, h[l - 1], t[l - 1], mlp.act[0]);
        init_bias<H, FT>(t[l], nullptr, q);
        if (SAVED == 1) hidden_mul<H, FT>(t[l], ffr[l - 1], td);
        else hidden_apply<H, FT>(t[l], pk + L.fh(l), td, lane);
      }
    }

    CVF_STAMP(11);
    // ---- last layer (1 x H):  W_L += sum alpha h_{NH-1} + tdot_{NH-1} ; b_L += sum alpha
    {
      const bool mine = z == NH % zs;   // the block that owns the last layer's gradient
      if (mine) {
        if (q == 0) {
#pragma unroll
          for (int ft = 0; ft < FT; ++ft) {
            SA1[fo + ft] = alpha[ft];
            if (tangent) SA2[fo + ft] = 1.0f;
          }
        }
        store_image<H, FT, false>(SB1, h[NH - 1], one, lane, fo);
        if (tangent) {
          Vec<H, FT> td;
          tangent_of<H, FT>(td, h[NH - 1], t[NH - 1], mlp.act[0]);
          store_image<H, FT, false>(SB2, td, one, lane, fo);
        }
      }
      __syncthreads();
      for (int ct = wave; ct < CTH && mine; ct += WPB) {
        const f32x4 acc = outer_tile(SA1, SB1, SA2, SB2, 0, ct, tangent, lane);
        if (q == 0) {  // output row 0 lives in register 0 of lanes 0..15
          const int wo = mlp.w_off[net][NH] - gbase, bo = mlp.b_off[net][NH] - gbase;
          const int i = 16 * ct + row16;
          if (i < H) GIs[wo + i] += acc[0];
          else if (i == H) GIs[bo] += acc[0];
        }
      }
      __syncthreads();
    }

    CVF_STAMP(12);
    // ---- reverse sweep
    Vec<H, FT> hbar;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ft = 0; ft < FT; ++ft)
#pragma unroll
        for (int r = 0; r < 4; ++r) hbar.v[rt][ft][r] = alpha[ft] * wl[rt][r];
    const int act0 = mlp.act[0];
#pragma unroll
    for (int l = NH - 1; l >= 0; --l) {
      CVF_STAMP(13 + (NH - 1 - l));
      Vec<H, FT> zbar, dl;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ft = 0; ft < FT; ++ft)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float hv = h[l].v[rt][ft][r];
            if (act0 == CVF_ACT_TANH) {
              const float om = 1.0f - hv * hv;
              float hb = hbar.v[rt][ft][r];
              if (tangent) {
                const float ev = (l == NH - 1) ? wl[rt][r] : e[l < NH - 1 ? l : 0].v[rt][ft][r];
                hb = fmaf(-2.0f * hv * t[l].v[rt][ft][r], ev, hb);
                dl.v[rt][ft][r] = ev * om;
              }
              zbar.v[rt][ft][r] = om * hb;
            } else {   // zbar = f' hbar + f'' t e  (the derivatives through the stored output)
              const float d1 = ef_act_d1_slow(act0, hv);
              float zb = d1 * hbar.v[rt][ft][r];
              if (tangent) {
                const float ev = (l == NH - 1) ? wl[rt][r] : e[l < NH - 1 ? l : 0].v[rt][ft][r];
                zb = fmaf(ef_act_d2_slow(act0, hv) * t[l].v[rt][ft][r], ev, zb);
                dl.v[rt][ft][r] = ev * d1;
              }
              zbar.v[rt][ft][r] = zb;
            }
          }
      const bool mine = z == l % zs;   // the block that owns layer l's gradient (l >= 1; the first layer is shared by column tile)
      if (l == 0 || mine) {
        store_image<H, FT, false>(SA1, zbar, one, lane, fo);
        if (tangent) {
          if (l == 0) store_image<H, FT, true>(SA2, dl, gamma, lane, fo);
          else store_image<H, FT, false>(SA2, dl, one, lane, fo);
        }
      }
      if (l > 0) {
        if (mine) {
          store_image<H, FT, false>(SB1, h[l - 1], one, lane, fo);
          if (tangent) {
            Vec<H, FT> td;
            tangent_of<H, FT>(td, h[l - 1], t[l - 1], mlp.act[0]);
            store_image<H, FT, false>(SB2, td, one, lane, fo);
          }
        }
        __syncthreads();
        for (int pr = wave; pr < RTO * CTH && mine; pr += WPB) {
          const int rt = pr / CTH, ct = pr - rt * CTH;
          add_tile(l, H, H, rt, ct, outer_tile(SA1, SB1, SA2, SB2, rt, ct, tangent, lane));
        }
        // hbar_{l-1} = W_l^T zbar_l  (registers; overlaps the other wave's outer products)
        init_bias<H, FT>(hbar, nullptr, q);
        hidden_apply<H, FT>(hbar, pk + L.th(l), zbar, lane);
        __syncthreads();
      } else {
        __syncthreads();
        // first layer: the B operands are the feature tile (+ ones row) and q, read straight from global memory.
        // The order in which the 64 frames are summed is free as long as A and B agree, so k-slot q of k-step
        // (j, c) is frame 16 j + 4 q + c: a lane's sixteen B values are then four 16-byte loads, and one load
        // instruction covers 64 contiguous bytes of each of its 16 feature rows (with the natural order 4 s + q
        // every 4-byte load touched 16 different sectors: 12k cycles per tile pair, half of this kernel).
        // A wave owns whole column tiles (both row tiles reuse the B registers); an odd last column tile is split by
        // row tile.  (Fetching the next column tile during the current one's MFMAs was tried: the 32 extra live
        // registers spill elsewhere in the kernel and cost more than the overlap gains.)
        static_assert(WPB == 2 || WPB == 4, "the column-tile schedules below are written for two or four waves per block");
        // column tile `ct` for the row tiles rt0, rt0 + rstep, ...: first the [f ; 1] half of the contraction for
        // all of them, then the q half - sixteen B registers live at a time
        auto outer0 = [&](int ct, int rt0, int rstep) {
          const int i = 16 * ct + row16;
          const int ic = i < D ? i : D - 1;
          // (no MFMA under lane-divergent control flow: operand values are selected per lane, the MFMAs are uniform)
          const float pad1 = i == D ? 1.0f : 0.0f;   // bias column; columns past it stay 0
          f32x4 acc[RTO];
          // the q half's operands are requested together with the feature half's (SAVED: the registers the forward
          // recompute used to hold are free): one memory round trip per column tile instead of two
          float4 bq[4];
          if (SAVED == 1 && tangent) {
            const float4* qb = reinterpret_cast<const float4*>(q_tile + (int64_t)ic * CVF_TILE + 4 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) bq[j] = qb[4 * j];
          }
          {
            const float4* fb = reinterpret_cast<const float4*>(f_tile + (int64_t)ic * CVF_TILE + 4 * q);
            float4 b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = fb[4 * j];
#pragma unroll
            for (int rt = 0; rt < RTO; ++rt) {
              acc[rt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
              if (rt >= rt0 && (rt - rt0) % rstep == 0) {
                const float4* a1 = reinterpret_cast<const float4*>(SA1 + (16 * rt + row16) * kPitch + 4 * q);
                float4 av[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) av[j] = a1[4 * j];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  acc[rt] = mfma4(av[j].x, i < D ? b[j].x : pad1, acc[rt]);
                  acc[rt] = mfma4(av[j].y, i < D ? b[j].y : pad1, acc[rt]);
                  acc[rt] = mfma4(av[j].z, i < D ? b[j].z : pad1, acc[rt]);
                  acc[rt] = mfma4(av[j].w, i < D ? b[j].w : pad1, acc[rt]);
                }
              }
            }
          }
          if (tangent) {
            if (SAVED != 1) {
              const float4* qb = reinterpret_cast<const float4*>(q_tile + (int64_t)ic * CVF_TILE + 4 * q);
#pragma unroll
              for (int j = 0; j < 4; ++j) bq[j] = qb[4 * j];
            }
            const float4 (&b)[4] = bq;
#pragma unroll
            for (int rt = 0; rt < RTO; ++rt) {
              if (rt >= rt0 && (rt - rt0) % rstep == 0) {
                const float4* a2 = reinterpret_cast<const float4*>(SA2 + (16 * rt + row16) * kPitch + 4 * q);
                float4 av[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) av[j] = a2[4 * j];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  acc[rt] = mfma4(av[j].x, i < D ? b[j].x : 0.0f, acc[rt]);
                  acc[rt] = mfma4(av[j].y, i < D ? b[j].y : 0.0f, acc[rt]);
                  acc[rt] = mfma4(av[j].z, i < D ? b[j].z : 0.0f, acc[rt]);
                  acc[rt] = mfma4(av[j].w, i < D ? b[j].w : 0.0f, acc[rt]);
                }
              }
            }
          }
#pragma unroll
          for (int rt = 0; rt < RTO; ++rt)
            if (rt >= rt0 && (rt - rt0) % rstep == 0) {
              if (args.direct0) put_tile0(rt, ct, acc[rt]);
              else add_tile(0, H, D, rt, ct, acc[rt]);
            }
        };
        if constexpr (WPB == 2) {   // this block's column tiles z, z + zs, ...: whole ones dealt to the waves, an odd last one split by row tile
          const int nz = z < CT1 ? (CT1 - z + zs - 1) / zs : 0;
          const int nfull = nz & ~1;
          for (int j = wave; j < nfull; j += WPB) outer0(z + j * zs, 0, 1);
          if (nz & 1) outer0(z + (nz - 1) * zs, wave, WPB);
        } else {   // four waves: the (column tile, row tile) pairs dealt round-robin
          for (int pr = wave; pr < RTO * CT1; pr += WPB) outer0(pr / RTO, pr % RTO, RTO);
        }
        __syncthreads();
      }
    }
  }

  CVF_STAMP(17);
  // ---- flush this block's partial gradient of `net` into its slab row
  __syncthreads();
  float* out = out0;
  if (zs == 1 && !args.direct0) {
    for (int i = tid; i < gspan; i += NT) out[i] = GI[i];
  } else {   // the zs blocks of this (row, net) write disjoint parts of the row
    const int w0 = mlp.w_off[net][0] - gbase, b0 = mlp.b_off[net][0] - gbase;
    if (!args.direct0)   // first layer by column tile
      for (int ct = z; ct < CT1; ct += zs)
        for (int e = tid; e < 16 * H; e += NT) {
          const int o = e >> 4, i = 16 * ct + (e & 15);
          if (i < D) out[w0 + o * D + i] = GI[w0 + o * D + i];
          else if (i == D) out[b0 + o] = GI[b0 + o];
        }
    for (int l = 1; l <= NH; ++l)   // the layers this block owns
      if (z == l % zs) {
        const int wo = mlp.w_off[net][l] - gbase, bo = mlp.b_off[net][l] - gbase, n_out = l < NH ? H : 1;
        for (int i = tid; i < n_out * H; i += NT) out[wo + i] = GIs[wo + i];
        for (int i = tid; i < n_out; i += NT) out[bo + i] = GIs[bo + i];
      }
  }
  // one gradient per optimiser step: advance the step counter read by the Adam that follows
  if (step != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && z == 0 && tid == 0) *step += 1;
  CVF_STAMP(18);
}

// grad[p] = sum over slab rows, fixed order: 32 row groups (strided) per parameter, then the 32
// sub-sums in sequence -> bitwise reproducible without atomics.  With `adam` set (single-process runs:
// no cross-rank reduction of the gradient in between) the same thread applies the Adam update.
// kSlabGY row groups: 32 for the hundreds of rows of a dipeptide-sized batch (one round of loads), 4 when there are at most 64
// rows (large molecules: few tiles, ~50 000 parameters - 1605 blocks of 1024 threads that each read ONE row were 22 us, mostly
// wave launches)
constexpr int kSlabPX = 32, kSlabU = 10;   // parameters per block, loads in flight per thread
template <int kSlabGY>
__global__ __launch_bounds__(kSlabPX * kSlabGY) void slab_reduce_kernel(const float* __restrict__ slab, int64_t nrows, int P,
                                                                        float* __restrict__ grad, const float* __restrict__ mask,
                                                                        int use_adam, AdamDev adam, cvf_mlp_desc mlp,
                                                                        const double* __restrict__ pair_partial, int n_pair,
                                                                        double* __restrict__ pair_out, P2PLL ll) {
  // (optional rider: one extra block adds n_pair rows of [a, b] partial sums in a fixed order -> pair_out = [a, b, a / b];
  //  the autoencoder step's loss, which would otherwise be a launch of its own between the step kernel and this one)
  if (pair_partial != nullptr && blockIdx.x == gridDim.x - 1) {
    const int t = threadIdx.y * kSlabPX + threadIdx.x;
    if (t < CVF_WAVE) {
      double a0 = 0.0, a1 = 0.0;
      for (int g = t; g < n_pair; g += CVF_WAVE) {
        a0 += pair_partial[2 * g];
        a1 += pair_partial[2 * g + 1];
      }
      a0 = wave_sum(a0);
      a1 = wave_sum(a1);
      if (t == 0) {
        pair_out[0] = a0;
        pair_out[1] = a1;
        pair_out[2] = a0 / a1;
      }
    }
    return;
  }
  __shared__ float sub[kSlabGY][kSlabPX];
  __shared__ PackTab tab;
  const int px = threadIdx.x, gy = threadIdx.y;
  __shared__ AdamScalars scal;
  // the offset table and the step's bias corrections (two fp64 pow calls) are prepared by idle lanes while the rows load
  if (use_adam && adam.packed != nullptr && px == 0 && gy == 1) pack_tab_fill(tab, mlp);
  if (use_adam && px == 0 && gy == 2) scal = adam_scalars(adam);
  const int p = blockIdx.x * kSlabPX + px;
  // data-parallel step (ll.world > 0): the gradient exchange carries the number of the step's STATISTICS exchange (collective #1
  // ran in an earlier launch of this step: the word is constant while this kernel runs, so every workgroup reads the same number
  // without a ticket - 207 agent-scope atomic adds on one word were 3 us at the end of this launch).  That exchange is also what
  // keeps a fast rank from overwriting words a slow rank has not read yet: to finish it the fast rank needed the slow rank's
  // sums, which the slow rank sent after its previous gradient exchange.
  const unsigned ex = ll.world > 0 ? __hip_atomic_load(ll.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  // kSlabU independent loads in flight per thread (two left every thread with ~10 dependent round trips at 313 rows; with
  // 32 row groups a 20 000-frame batch is one round); rows past the end re-read the group's first row with weight 0: no
  // branch around the loads
  float m0 = 0.0f, v0 = 0.0f, th0 = 0.0f;   // the optimiser state of this thread's parameter, requested with the rows
  if (use_adam && gy == 0 && p < P) {
    m0 = adam.m[p];
    v0 = adam.v[p];
    th0 = adam.theta[p];
  }
  float au[kSlabU];
#pragma unroll
  for (int u = 0; u < kSlabU; ++u) au[u] = 0.0f;
  if (p < P) {
    for (int64_t g0 = gy; g0 < nrows; g0 += kSlabGY * kSlabU) {
      float v[kSlabU];
#pragma unroll
      for (int u = 0; u < kSlabU; ++u) {
        const int64_t g = g0 + kSlabGY * u;
        v[u] = slab[(g < nrows ? g : g0) * P + p];
      }
#pragma unroll
      for (int u = 0; u < kSlabU; ++u) au[u] += (g0 + kSlabGY * u < nrows) ? v[u] : 0.0f;
    }
  }
  float acc = 0.0f;
#pragma unroll
  for (int u = 0; u < kSlabU; ++u) acc += au[u];
  sub[gy][px] = acc;
  __syncthreads();
  if (gy == 0 && p < P) {
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < kSlabGY; ++t) s += sub[t][px];
    if (mask != nullptr) s *= mask[p];   // structural zeros of block-structured layers / frozen parameters
    // data-parallel step: this rank's share -> every peer's window, the `world` shares of the entry from my own window, added in
    // rank order (cvf_p2p.hpp) - collective #2 of SURVEY.md section 8e inside this launch, identical Adam on every rank behind it
    if (ll.world > 0) s = p2p_ll_allreduce_grad(ll, ex, p, s);
    grad[p] = s;
    if (use_adam) adam_apply(adam, scal, tab, p, s, m0, v0, th0);
  }
  // (misuse guard: two gradient exchanges with no statistics exchange between them would carry the same number and read each
  //  other's words - one thread notes the number this launch used and flags a repeat in the communicator's error word)
  if (ll.world > 0 && blockIdx.x == 0 && px == 0 && gy == 0) {
    if (__hip_atomic_load(ll.epoch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ex)
      __hip_atomic_store(ll.error, 0x40000000u | ex, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(ll.epoch + 1, ex, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

bool ef_shape(const cvf_mlp_desc* m, int* H, int* NH) {
  if (m->n_layers < 2 || m->n_layers > CVF_MAX_LAYERS || m->dims[m->n_layers] != 1) return false;
  *H = m->dims[1];
  *NH = m->n_layers - 1;
  for (int l = 1; l < m->n_layers; ++l)
    if (m->dims[l] != *H) return false;
  // one activation for all hidden layers (any code of include/cvf.h with two derivatives through its output), none after the last
  const int a0 = m->act[0];
  if (a0 < CVF_ACT_TANH || a0 > CVF_ACT_SOFTPLUS) return false;
  for (int l = 0; l < m->n_layers; ++l)
    if (m->act[l] != (l + 1 < m->n_layers ? a0 : 0)) return false;
  return true;
}

int64_t bwd_grid(int64_t n_tiles) { return n_tiles < 1024 ? n_tiles : 1024; }

constexpr int kFusedMaxH = 32;   // widest hidden layer of the fused / register-resident launches (see ef_dispatch)
template <class F>
bool ef_dispatch(int H, int NH, F&& f) {
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
  EF_CASE(32, 2) EF_CASE(32, 3)
  // wider hidden layers (up to 64 units; other widths are zero-padded to 48 / 64 by the host): the plain forward and backward
  // kernels only (kWideH) - the fused launches keep a net's whole chain in registers and are instantiated up to 32 units
  EF_CASE(48, 2) EF_CASE(48, 3)
  EF_CASE(64, 2) EF_CASE(64, 3)
  // four and five hidden layers (e.g. regulariser o encoder chains of RegAutoEncoderTask's generator mode): two widths,
  // narrower nets are zero-padded to them by the host
  EF_CASE(20, 4) EF_CASE(20, 5)
  EF_CASE(32, 4) EF_CASE(32, 5)
#undef EF_CASE
  return false;
}

}  // namespace

__global__ void ef_pack_kernel(cvf_mlp_desc mlp, const float* __restrict__ theta, float* __restrict__ packed) {
  __shared__ PackTab tab;
  if (threadIdx.x == 0) pack_tab_fill(tab, mlp);
  __syncthreads();
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < mlp.n_params) pack_scatter(tab, p, theta[p], packed);
}

extern "C" int64_t cvf_ef_pack_floats(const cvf_mlp_desc* mlp) {
  int H, NH;
  if (!mlp || !ef_shape(mlp, &H, &NH) || !ef_dispatch(H, NH, [](auto, auto) {})) return 0;   // 0: no kernel instance for this shape
  return (int64_t)mlp->n_nets * pack_layout(H, NH, mlp->dims[0]).per_net;
}

extern "C" int cvf_ef_pack(const cvf_mlp_desc* mlp, const float* theta, float* packed, void* stream) {
  CVF_REQUIRE(mlp && theta && packed, "cvf_ef_pack: bad argument");
  int H, NH;
  CVF_REQUIRE(ef_shape(mlp, &H, &NH), "cvf_ef_pack: unsupported net shape");
  (void)hipMemsetAsync(packed, 0, sizeof(float) * cvf_ef_pack_floats(mlp), (hipStream_t)stream);
  hipLaunchKernelGGL(ef_pack_kernel, dim3((mlp->n_params + 255) / 256), dim3(256), 0, (hipStream_t)stream, *mlp, theta, packed);
  return cvf_check_launch("ef_pack_kernel");
}

// The forward kernel that shares one weight fetch among four tiles can also hand its activations to the backward kernel.
static bool fwd_wg_ok(const cvf_mlp_desc* mlp, int H, int NH) {
  bool wg = (size_t)pack_layout(H, NH, mlp->dims[0]).per_net * sizeof(float) <= 64 * 1024;
  if (getenv("CVF_FWD_WG")) wg = wg && atoi(getenv("CVF_FWD_WG")) != 0;   // developer override
  return wg;
}

extern "C" int64_t cvf_ef_saved_floats(const cvf_mlp_desc* mlp, int64_t n_tiles) {
  int H, NH;
  if (!mlp || !ef_shape(mlp, &H, &NH) || getenv("CVF_NO_SAVED")) return 0;
  if (H > kFusedMaxH) return 0;   // (wider layers: the backward kernel runs the chain forward again)
  const bool wide = mlp->dims[0] > kWideD;   // (+ one vector per (tile, net) for t0 = W0 q, see ef_backward_impl)
  if (!fwd_wg_ok(mlp, H, NH) && !wide) return 0;
  int64_t per_vec = 0;
  const bool ok = ef_dispatch(H, NH, [&](auto h_, auto) { per_vec = saved_per_vec<decltype(h_)::value>(); });
  return ok ? n_tiles * mlp->n_nets * (NH + (wide ? 1 : 0)) * per_vec : 0;
}

extern "C" int cvf_ef_mlp_fwd(const cvf_mlp_desc* mlp, const float* theta, const float* packed, const float* feat_tiled,
                              int64_t n_tiles, float* y_tiled, float* g_tiled, float* saved, void* stream) {
  CVF_REQUIRE(mlp && theta && packed && feat_tiled && y_tiled && n_tiles > 0, "cvf_ef_mlp_fwd: bad argument");
  int H, NH;
  CVF_REQUIRE(ef_shape(mlp, &H, &NH),
              "cvf_ef_mlp_fwd: nets must be d0->H->..->H->1 with one activation of include/cvf.h after every hidden layer (got %d layers)", mlp->n_layers);
  CVF_REQUIRE(mlp->n_nets >= 1 && mlp->n_nets <= CVF_MAX_NETS, "cvf_ef_mlp_fwd: k=%d out of range", mlp->n_nets);
  // few tiles: split each 64-frame tile over two waves so that the launch still fills the 1024 SIMDs
  bool split = n_tiles * mlp->n_nets < 2048;
  if (getenv("CVF_FWD_SPLIT")) split = atoi(getenv("CVF_FWD_SPLIT")) != 0;   // developer override
  const bool pre = mlp->dims[0] <= 72 && saved == nullptr;   // every load up front (see ef_fwd_pre_kernel)
  const size_t wlds = (size_t)pack_layout(H, NH, mlp->dims[0]).per_net * sizeof(float);
  const bool wg = fwd_wg_ok(mlp, H, NH);  // one fetch of the weights per four tiles (see ef_fwd_wg_kernel)
  CVF_REQUIRE(saved == nullptr || cvf_ef_saved_floats(mlp, 1) > 0, "cvf_ef_mlp_fwd: this shape has no activation hand-off (cvf_ef_saved_floats() == 0)");
  const bool launched = ef_dispatch(H, NH, [&](auto h_, auto nh_) {
    constexpr int kH = decltype(h_)::value, kNH = decltype(nh_)::value;
    bool done = false;
    if constexpr (kH <= kFusedMaxH) {
      if (mlp->dims[0] > kWideD && getenv("CVF_NO_FWD_WIDE") == nullptr) {
        (void)hipFuncSetAttribute((const void*)ef_fwd_wide_kernel<kH, kNH>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)wide_lds_bytes<kH>());
        hipLaunchKernelGGL((ef_fwd_wide_kernel<kH, kNH>), dim3((unsigned)n_tiles, mlp->n_nets), dim3(64 * kWW), wide_lds_bytes<kH>(),
                           (hipStream_t)stream, *mlp, theta, packed, feat_tiled, y_tiled, g_tiled, saved);
        done = true;
      } else if (!wg && pre) {
        if (split)
          hipLaunchKernelGGL((ef_fwd_pre_kernel<kH, kNH, 2>), dim3((unsigned)(2 * n_tiles), mlp->n_nets), dim3(64), 0,
                             (hipStream_t)stream, *mlp, theta, packed, feat_tiled, y_tiled, g_tiled);
        else
          hipLaunchKernelGGL((ef_fwd_pre_kernel<kH, kNH, 4>), dim3((unsigned)n_tiles, mlp->n_nets), dim3(64), 0,
                             (hipStream_t)stream, *mlp, theta, packed, feat_tiled, y_tiled, g_tiled);
        done = true;
      }
    }
    if (done) {
    } else if (wg) {
      if (wlds > 48 * 1024)
        (void)hipFuncSetAttribute((const void*)ef_fwd_wg_kernel<kH, kNH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds);
      hipLaunchKernelGGL((ef_fwd_wg_kernel<kH, kNH>), dim3((unsigned)((n_tiles + 3) / 4), mlp->n_nets), dim3(256), wlds,
                         (hipStream_t)stream, *mlp, theta, packed, feat_tiled, n_tiles, y_tiled, g_tiled, saved);
    } else if (split)
      hipLaunchKernelGGL((ef_fwd_mfma_kernel<kH, kNH, 2>), dim3((unsigned)(2 * n_tiles), mlp->n_nets), dim3(64), 0,
                         (hipStream_t)stream, *mlp, theta, packed, feat_tiled, y_tiled, g_tiled, saved);
    else
      hipLaunchKernelGGL((ef_fwd_mfma_kernel<kH, kNH, 4>), dim3((unsigned)n_tiles, mlp->n_nets), dim3(64), 0,
                         (hipStream_t)stream, *mlp, theta, packed, feat_tiled, y_tiled, g_tiled, saved);
  });
  CVF_REQUIRE(launched, "cvf_ef_mlp_fwd: no kernel instance for hidden width %d x %d layers", H, NH);
  return cvf_check_launch("ef_fwd_mfma_kernel");
}

int cvf_ef_stats_finish(const cvf_ef_cfg* cfg, int n_rows, const double* partial, double* stats, double* loss_vec, double* coef,
                        hipStream_t s);

static size_t fwd_metric_lds(const cvf_pp_desc* pp, int k) {
  const size_t head = ((size_t)CVF_TILE * x_tile_stride(pp->n_coord) + 3 * (size_t)pp->n_align + pp->n_coord + 3) & ~(size_t)3;
  return (head + (size_t)k * pp->n_coord * CVF_TILE + (size_t)k * CVF_TILE) * sizeof(float);
}

extern "C" int cvf_ef_fwd_metric_supported(const cvf_mlp_desc* mlp, const cvf_pp_desc* pp) {
  int H, NH;
  if (!mlp || !pp || !ef_shape(mlp, &H, &NH) || getenv("CVF_NO_FWD_METRIC")) return 0;
  if (!ef_dispatch(H, NH, [](auto, auto) {}) || H > kFusedMaxH) return 0;
  const int fast = CVF_PP_ALIGN_CONTIG | CVF_PP_PURE_POSITION;
  if (pp->mode != CVF_PP_ALIGN || pp->align_w || (pp->flags & fast) != fast || pp->n_align > pp->n_rec) return 0;
  if (pp->d_r != 3 * pp->n_rec || pp->d_r != mlp->dims[0] || pp->d_r > 72 || pp->n_coord > 192) return 0;
  return fwd_metric_lds(pp, mlp->n_nets) <= 80 * 1024 && cvf_ef_saved_floats(mlp, 1) > 0;
}

static int fwd_metric_launch(bool with_k1, const cvf_mlp_desc* mlp, const float* theta, const float* packed, float* feat_tiled,
                             const cvf_pp_desc* pp, const float* x, int64_t B, float* aux_tiled, const float* a, float* y_tiled,
                             float* saved, float* q_tiled, float* e_tiled, const cvf_ef_cfg* cfg, const float* w, double* scratch,
                             double* stats, double* loss_vec, double* coef, void* stream) {
  CVF_REQUIRE(cvf_ef_fwd_metric_supported(mlp, pp), "cvf_ef_fwd_metric_stats: shape not covered (cvf_ef_fwd_metric_supported() == 0)");
  CVF_REQUIRE(theta && packed && feat_tiled && x && a && y_tiled && q_tiled && e_tiled && cfg && w && scratch && B > 0,
              "cvf_ef_fwd_metric_stats: bad argument");
  CVF_REQUIRE(stats != nullptr || cvf_ntiles(B) <= kFuseMaxTiles,
              "cvf_ef_fwd_metric_stats: stats == NULL (rows left for cvf_ef_stats_finish_rows) needs cvf_ef_fused_stats_rows() > 0");
  CVF_REQUIRE(with_k1 || aux_tiled, "cvf_ef_fwd_metric_stats: aux_tiled missing");
  CVF_REQUIRE(cfg->k == mlp->n_nets && cfg->lag_idx == 0, "cvf_ef_fwd_metric_stats: generator mode only, cfg.k must equal the number of nets");
  CVF_REQUIRE(loss_vec == nullptr || coef != nullptr, "cvf_ef_fwd_metric_stats: loss_vec without coef");
  int H, NH;
  ef_shape(mlp, &H, &NH);
  const int k = mlp->n_nets;
  const int64_t T = cvf_ntiles(B);
  const int ns = cvf_ef_nstats(k, 0);
  MetricFuse f = {};
  f.on = T <= kFuseMaxTiles ? 1 : 0;
  f.ns = ns;
  f.w = w;
  f.y_tiled = y_tiled;
  f.partial = scratch;
  // (A variant with two waves per net - half the forward chain and two lanes per frame in the derivative part - was
  // built and measured at 52.8 us against 45.5: five block-wide barriers among six waves and the pair-wise bank
  // conflicts cost more than the shorter chains save.)
  const size_t lds = fwd_metric_lds(pp, k);
  ef_dispatch(H, NH, [&](auto h_, auto nh_) {
    constexpr int kH = decltype(h_)::value, kNH = decltype(nh_)::value;
    auto go = [&](auto kernel) {
      if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(kernel, dim3((unsigned)T), dim3(64 * k), lds, (hipStream_t)stream, *mlp, theta, packed, feat_tiled, *pp, x, B,
                         aux_tiled, a, y_tiled, saved, q_tiled, e_tiled, f);
    };
    if constexpr (kH <= kFusedMaxH) {
      if (with_k1) go(ef_fwd_metric_kernel<kH, kNH, true>);
      else go(ef_fwd_metric_kernel<kH, kNH, false>);
    }
  });
  int rc = cvf_check_launch("ef_fwd_metric_kernel");
  if (rc || stats == nullptr) return rc;   // stats == NULL: the caller finishes the rows itself (cvf_ef_stats_finish_rows)
  if (f.on) return cvf_ef_stats_finish(cfg, (int)T, scratch, stats, loss_vec, coef, (hipStream_t)stream);
  return cvf_ef_stats(cfg, B, w, y_tiled, e_tiled, nullptr, nullptr, scratch + T * ns, stats, loss_vec, coef, stream);
}

extern "C" int cvf_ef_fwd_metric_stats(const cvf_mlp_desc* mlp, const float* theta, const float* packed, const float* feat_tiled,
                                       const cvf_pp_desc* pp, const float* x, int64_t B, const float* aux_tiled,
                                       const float* a, float* y_tiled, float* saved, float* q_tiled, float* e_tiled,
                                       const cvf_ef_cfg* cfg, const float* w, double* scratch, double* stats,
                                       double* loss_vec, double* coef, void* stream) {
  return fwd_metric_launch(false, mlp, theta, packed, const_cast<float*>(feat_tiled), pp, x, B, const_cast<float*>(aux_tiled), a,
                           y_tiled, saved, q_tiled, e_tiled, cfg, w, scratch, stats, loss_vec, coef, stream);
}

extern "C" int cvf_ef_align_fwd(const cvf_mlp_desc* mlp, const float* theta, const float* packed, float* feat_tiled,
                                const cvf_pp_desc* pp, const float* x, const float* x_lag, int64_t B, float* y_tiled, float* saved,
                                void* stream) {
  CVF_REQUIRE(cvf_ef_align_fwd_metric_supported(mlp, pp), "cvf_ef_align_fwd: shape not covered (cvf_ef_align_fwd_metric_supported() == 0)");
  CVF_REQUIRE(theta && packed && feat_tiled && x && y_tiled && B > 0, "cvf_ef_align_fwd: bad argument");
  int H, NH;
  ef_shape(mlp, &H, &NH);
  const int k = mlp->n_nets;
  const int64_t T = cvf_ntiles(B), n_tiles = x_lag ? 2 * T : T;
  const size_t lds = (((size_t)CVF_TILE * x_tile_stride(pp->n_coord) + 3 * (size_t)pp->n_align + 3) & ~(size_t)3) * sizeof(float) +
                     (size_t)mlp->dims[0] * CVF_TILE * sizeof(float);
  CVF_REQUIRE(lds <= 160 * 1024, "cvf_ef_align_fwd: %zu B of LDS per workgroup (> 160 KiB)", lds);
  ef_dispatch(H, NH, [&](auto h_, auto nh_) {
    constexpr int kH = decltype(h_)::value, kNH = decltype(nh_)::value;
    if constexpr (kH <= kFusedMaxH) {
      auto kernel = ef_align_fwd_kernel<kH, kNH>;
      if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(kernel, dim3((unsigned)n_tiles), dim3(64 * k), lds, (hipStream_t)stream, *mlp, theta, packed, feat_tiled, *pp, x,
                         x_lag, T, B, y_tiled, saved);
    }
  });
  return cvf_check_launch("ef_align_fwd_kernel");
}

// Rows of per-tile partial sums the fused launches leave in `scratch` (0: the batch is too large for the fused sums and
// the launch must be given `stats` so that it runs the two-stage reduction itself), and the launch that adds them.
extern "C" int64_t cvf_ef_fused_stats_rows(const cvf_mlp_desc* mlp, const cvf_pp_desc* pp, int64_t B, int with_align) {
  (void)mlp; (void)pp; (void)with_align;   // one row per tile for every fused launch at present
  return cvf_ntiles(B) <= kFuseMaxTiles ? cvf_ntiles(B) : 0;
}
extern "C" int cvf_ef_stats_finish_rows(const cvf_ef_cfg* cfg, int64_t n_rows, const double* partial, double* stats,
                                        double* loss_vec, double* coef, void* stream) {
  CVF_REQUIRE(cfg && partial && stats && n_rows > 0, "cvf_ef_stats_finish_rows: bad argument");
  CVF_REQUIRE(loss_vec == nullptr || coef != nullptr, "cvf_ef_stats_finish_rows: loss_vec without coef");
  return cvf_ef_stats_finish(cfg, (int)n_rows, partial, stats, loss_vec, coef, (hipStream_t)stream);
}

// the alignment kernel folded in as well: one launch from coordinates to q, E and the batch sums
extern "C" int cvf_ef_align_fwd_metric_supported(const cvf_mlp_desc* mlp, const cvf_pp_desc* pp) {
  // (the covariance partial sums pass through the g images: 15 doubles per frame must fit an image of n_coord floats)
  return cvf_ef_fwd_metric_supported(mlp, pp) && pp->n_coord >= 30 && getenv("CVF_NO_ALIGN_FUSED") == nullptr;
}

extern "C" int cvf_ef_align_fwd_metric_stats(const cvf_mlp_desc* mlp, const float* theta, const float* packed, float* feat_tiled,
                                             const cvf_pp_desc* pp, const float* x, int64_t B, float* aux_tiled, const float* a,
                                             float* y_tiled, float* saved, float* q_tiled, float* e_tiled,
                                             const cvf_ef_cfg* cfg, const float* w, double* scratch, double* stats,
                                             double* loss_vec, double* coef, void* stream) {
  CVF_REQUIRE(cvf_ef_align_fwd_metric_supported(mlp, pp), "cvf_ef_align_fwd_metric_stats: shape not covered");
  return fwd_metric_launch(true, mlp, theta, packed, feat_tiled, pp, x, B, aux_tiled, a, y_tiled, saved, q_tiled, e_tiled, cfg, w,
                           scratch, stats, loss_vec, coef, stream);
}

extern "C" int64_t cvf_ef_backward_slab_rows(int64_t n_tiles) { return bwd_grid(n_tiles); }

static int ef_backward_impl(const cvf_ef_cfg* cfg, const cvf_mlp_desc* mlp, const float* theta, const float* packed,
                            int64_t B, const float* w, const float* w_lag, const float* feat_tiled,
                            const float* y_tiled, const float* q_tiled, const double* coef, float* slab,
                            int32_t* step_count, float* saved, void* stream) {
  CVF_REQUIRE(cfg && mlp && theta && packed && w && feat_tiled && y_tiled && coef && slab && B > 0,
              "cvf_ef_backward: bad argument");
  CVF_REQUIRE(cfg->lag_idx > 0 || q_tiled, "cvf_ef_backward: generator mode needs q");
  CVF_REQUIRE(cfg->lag_idx == 0 || w_lag, "cvf_ef_backward: transfer mode needs w_lag");
  int H, NH;
  CVF_REQUIRE(ef_shape(mlp, &H, &NH), "cvf_ef_backward: unsupported net shape");
  EfBwdArgs a;
  a.k = cfg->k;
  a.lag_idx = cfg->lag_idx;
  a.B = B;
  a.T = cvf_ntiles(B);
  a.n_tiles = cfg->lag_idx > 0 ? 2 * a.T : a.T;
  const int64_t G = bwd_grid(a.n_tiles);
  // wide first layers (d0 > kWideD): the first layer's gradient - 2 x 2 x 32 matrix instructions per column tile, 25 tiles at
  // d0 = 384, 100 k of one SIMD's cycles per (tile, net) - is shared out over up to four blocks per (tile, net) when the grid
  // would otherwise leave most of the chip idle; with the activation hand-off they also get t0 = W0 q from a launch of its own
  const bool wide = mlp->dims[0] > kWideD;
  // (measured at the config-5 shape, backward us for zs = 1 / 2 / 3 / 4: 2000 frames 65 / 50 / 48 / 42, 4000 frames 81 / 62 / 83 / 79,
  //  6000 frames 89 / 107 / 94 / 110, 16 000 frames 197 / 203 / 220 / 231 - every block repeats the chains, so sharing pays only
  //  while the blocks are few: as many as keep the grid at or under 768 blocks)
  const int64_t want = 768 / (G * cfg->k);
  a.zs = wide ? (int)(want < 1 ? 1 : want > 4 ? 4 : want) : 1;
  if (getenv("CVF_BWD_ZS")) a.zs = atoi(getenv("CVF_BWD_ZS")) > 0 && wide ? atoi(getenv("CVF_BWD_ZS")) : a.zs;   // developer switch
  a.t0 = nullptr;
  a.direct0 = wide && G == a.n_tiles && getenv("CVF_BWD_NO_DIRECT") == nullptr ? 1 : 0;
  dim3 grid((unsigned)G, cfg->k, a.zs);
  // the LDS gradient image relies on each net's parameters being one contiguous run of the flat buffer
  const int span = mlp->b_off[0][NH] + 1 - mlp->w_off[0][0];
  int covered = 0;
  for (int n = 0; n < mlp->n_nets; ++n) {
    CVF_REQUIRE(mlp->b_off[n][NH] + 1 - mlp->w_off[n][0] == span, "cvf_ef_backward: nets are not laid out contiguously");
    for (int l = 0; l <= NH; ++l)
      CVF_REQUIRE(mlp->w_off[n][l] >= mlp->w_off[n][0] && mlp->b_off[n][l] < mlp->w_off[n][0] + span,
                  "cvf_ef_backward: nets are not laid out contiguously");
    covered += span;
  }
  CVF_REQUIRE(covered == mlp->n_params, "cvf_ef_backward: flat buffer holds parameters outside the nets");
  // (the shared-out flush assumes the usual order W0, b0, W1, ... inside a net's run)
  for (int n = 0; n < mlp->n_nets && (a.zs > 1 || a.direct0); ++n) {
    bool usual = mlp->b_off[n][0] == mlp->w_off[n][0] + H * mlp->dims[0];
    for (int l = 1; l <= NH; ++l)
      usual = usual && mlp->w_off[n][l] >= mlp->b_off[n][0] + H && mlp->b_off[n][l] >= mlp->b_off[n][0] + H;
    if (!usual) {
      a.zs = 1;
      a.direct0 = 0;
    }
  }
  grid.z = a.zs;
  const size_t lds_dyn = (size_t)(span - (a.direct0 ? H * mlp->dims[0] + H : 0)) * sizeof(float);
  const bool launched = ef_dispatch(H, NH, [&](auto h_, auto nh_) {
    constexpr int kH = decltype(h_)::value, kNH = decltype(nh_)::value;
    if (kH <= kFusedMaxH && saved != nullptr && wide && cfg->lag_idx == 0 && getenv("CVF_NO_T0") == nullptr) {
      float* t0 = saved + a.n_tiles * cfg->k * (int64_t)(kNH * saved_per_vec<kH>());   // (behind the activations)
      (void)hipFuncSetAttribute((const void*)ef_t0_kernel<kH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wide_lds_bytes<kH>());
      hipLaunchKernelGGL((ef_t0_kernel<kH>), dim3((unsigned)a.T, cfg->k), dim3(64 * kWW), wide_lds_bytes<kH>(), (hipStream_t)stream,
                         *mlp, packed, q_tiled, t0);
      a.t0 = t0;
    }
    if (saved != nullptr)
      hipLaunchKernelGGL((ef_bwd_mfma_kernel<kH, kNH, 2, 1>), grid, dim3(128), lds_dyn, (hipStream_t)stream, a, *mlp, theta,
                         packed, w, w_lag, feat_tiled, y_tiled, q_tiled, coef, slab, step_count, saved);
    else
      hipLaunchKernelGGL((ef_bwd_mfma_kernel<kH, kNH, 2, 0>), grid, dim3(128), lds_dyn, (hipStream_t)stream, a, *mlp, theta,
                         packed, w, w_lag, feat_tiled, y_tiled, q_tiled, coef, slab, step_count, saved);
  });
  CVF_REQUIRE(launched, "cvf_ef_backward: no kernel instance for hidden width %d x %d layers", H, NH);
  return cvf_check_launch("ef_bwd_mfma_kernel");
}

extern "C" int cvf_ef_backward(const cvf_ef_cfg* cfg, const cvf_mlp_desc* mlp, const float* theta, const float* packed,
                               int64_t B, const float* w, const float* w_lag, const float* feat_tiled,
                               const float* y_tiled, const float* q_tiled, const double* coef, float* slab,
                               int32_t* step_count, float* saved, void* stream) {
  return ef_backward_impl(cfg, mlp, theta, packed, B, w, w_lag, feat_tiled, y_tiled, q_tiled, coef, slab, step_count, saved,
                          stream);
}

int cvf_slab_reduce_impl(const float* slab, int64_t n_rows, int64_t n_params, float* grad, const float* mask,
                         const cvf_adam_args* adam, void* stream, const double* pair_partial = nullptr, int n_pair = 0,
                         double* pair_out = nullptr);
static int slab_reduce_ll(const float* slab, int64_t n_rows, int64_t n_params, float* grad, const float* mask,
                          const cvf_adam_args* adam, void* stream, const double* pair_partial, int n_pair, double* pair_out,
                          const P2PLL* ll);
extern "C" int cvf_slab_reduce(const float* slab, int64_t n_rows, int64_t n_params, float* grad, const cvf_adam_args* adam,
                               void* stream) {
  return cvf_slab_reduce_impl(slab, n_rows, n_params, grad, nullptr, adam, stream);
}
// data-parallel step: sum of the slab rows -> sum over ranks (peer-to-peer exchange inside the launch) -> Adam + fragment refresh.
// One launch where the step had three (cvf_slab_reduce, all-reduce, cvf_adam_step).  Must follow a statistics exchange of the same
// step (cvf_ef16_finish_dp / cvf_ef_stats_dp / cvf_ef_loss_dp / cvf_p2p_exchange_f64): it borrows that exchange's number, and a
// second call without one in between sets the communicator's error word.  One process per GPU: every workgroup waits
// for the peers' words of ITS parameters only, so the launch needs no workgroup of its own to make progress; several ranks on
// ONE GPU (the tests) must fit their workgroups on the chip together.
extern "C" int cvf_slab_reduce_dp(const float* slab, int64_t n_rows, int64_t n_params, float* grad, const cvf_adam_args* adam,
                                  void* p2p_comm, void* stream) {
  const P2PLL* ll = cvf_p2p_ll(p2p_comm, n_params);
  if (ll == nullptr) return -1;
  return slab_reduce_ll(slab, n_rows, n_params, grad, nullptr, adam, stream, nullptr, 0, nullptr, ll);
}
int cvf_slab_reduce_impl(const float* slab, int64_t n_rows, int64_t n_params, float* grad, const float* mask,
                         const cvf_adam_args* adam, void* stream, const double* pair_partial, int n_pair, double* pair_out) {
  return slab_reduce_ll(slab, n_rows, n_params, grad, mask, adam, stream, pair_partial, n_pair, pair_out, nullptr);
}
static int slab_reduce_ll(const float* slab, int64_t n_rows, int64_t n_params, float* grad, const float* mask,
                          const cvf_adam_args* adam, void* stream, const double* pair_partial, int n_pair, double* pair_out,
                          const P2PLL* llp) {
  CVF_REQUIRE(slab && grad && n_rows > 0 && n_params > 0, "cvf_slab_reduce: bad argument");
  CVF_REQUIRE(llp == nullptr || pair_partial == nullptr, "cvf_slab_reduce: the pair rider does not combine with the cross-rank exchange");
  P2PLL ll = {};
  if (llp != nullptr) ll = *llp;
  AdamDev ad{};
  cvf_mlp_desc md = {};
  if (adam != nullptr) {
    CVF_REQUIRE(adam->theta && adam->m && adam->v && adam->step_count, "cvf_slab_reduce: incomplete adam arguments");
    CVF_REQUIRE(adam->packed == nullptr || (adam->mlp != nullptr && adam->mlp->n_params == n_params),
                "cvf_slab_reduce: packed buffer needs its mlp desc");
    ad = AdamDev{adam->theta, adam->m, adam->v, (float)adam->lr, (float)adam->beta1, (float)adam->beta2, (float)adam->eps,
                 adam->step_count, adam->packed, adam->lr_dev};
    if (adam->packed) md = *adam->mlp;
  }
  const unsigned nb = (unsigned)((n_params + kSlabPX - 1) / kSlabPX) + (pair_partial != nullptr ? 1u : 0u);
  if (n_rows > 64)
    hipLaunchKernelGGL(slab_reduce_kernel<32>, dim3(nb), dim3(kSlabPX, 32), 0, (hipStream_t)stream, slab,
                       n_rows, (int)n_params, grad, mask, adam != nullptr ? 1 : 0, ad, md, pair_partial, n_pair, pair_out, ll);
  else
    hipLaunchKernelGGL(slab_reduce_kernel<4>, dim3(nb), dim3(kSlabPX, 4), 0, (hipStream_t)stream, slab,
                       n_rows, (int)n_params, grad, mask, adam != nullptr ? 1 : 0, ad, md, pair_partial, n_pair, pair_out, ll);
  return cvf_check_launch("slab_reduce_kernel");
}
