// EigenFunctionTask generator-mode step for the fast layout (pure position features on a contiguous align set, d_r <= 72)
// with SIXTEEN frames per wave: the matrix instruction's N is 16 frames, so a wave that owns 16 frames (one "unit") runs a
// quarter of the dependent chain of the 64-frames-per-wave kernels in ef_mfma.hip, the launch has four times as many waves
// and each needs half the registers - three to five waves share a SIMD and cover each other's LDS / L2 round trips
// (the 64-frame kernels ran one 65-72 k-cycle chain per SIMD at the reference's batch size: 10 % / 14 % of the fp32 peak).
//
//   front  (cvf_ef16_front):    block = unit of 16 consecutive frames, one wave per net.
//     stage the 16 x 3N coordinates (one contiguous run) -> wave 0: centroid + covariance with FOUR lanes per frame
//     (lane = 4 f + p takes atoms = p mod 4, quad-permute DPP sums) and the 3x3 solve -> all waves: aligned positions =
//     features into an LDS image [frame][feature] (+ the tiled copy the backward kernel reads) -> wave = net: forward chain,
//     d chain and g = W_1^T d_1 on the matrix cores (activations never leave registers; g lands in the wave's own LDS image
//     as 16-byte writes) -> the three passes of q = J A J^T g, E = g^T J A J^T g with four lanes per frame on that image ->
//     wave 0: the unit's row of batch sums (fp64, fixed order).
//   back   (cvf_ef16_backward): ef_bwd_mfma_kernel<H, NH, 4, SAVED16> in ef_mfma.hip - four waves per 64-frame tile, 16
//     frames each - reading the hidden activations in the layout this file's front kernel leaves them.
//
// Replaces, for these shapes, cvf_ef_align_fwd_metric_stats / cvf_ef_backward (core.py:403-457, 517).
#include "cvf_metric.hpp"
#include "ef_frag.hpp"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int kU = 16;       // frames per unit
constexpr int kImgP = 76;    // pitch of the [frame][feature] images: = 12 (mod 32), so the four-lanes-per-frame reads (address
                             // 76 f + 3 p + c) and the matrix cores' 16-byte row writes (76 col + 4 q) touch every bank once
constexpr int kAuxP = 21;    // pitch of the per-frame alignment record: R (9), centroid hi (3), K^-1 (6), centroid lo (3)
constexpr int kMaxRows16 = 16384;   // units whose rows of batch sums one finishing launch adds (above: cvf_ef_stats)
template <int NH>
__host__ __device__ constexpr int kHand() { return 2 * NH; }   // vectors of the front -> back hand-off per (tile, net)

__device__ __forceinline__ float quad_sumf16(float v) {
  v += dpp_movf<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dpp_movf<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  return v;
}
__device__ __forceinline__ double quad_sumd16(double v) {
  v += dpp_movd<0xB1, 0xf>(v);
  v += dpp_movd<0x4E, 0xf>(v);
  return v;
}

struct Front16Lds {   // offsets in floats
  int ref, a, aux, w, rs, y, e, feat, g, total;
};
__host__ __device__ inline Front16Lds front16_lds(int nc, int nal, int k) {
  Front16Lds L;
  const int stride = x_tile_stride(nc);
  L.ref = kU * stride;
  L.a = L.ref + 3 * nal;
  L.aux = (L.a + nc + 3) & ~3;
  L.w = L.aux + ((kU * kAuxP + 3) & ~3);
  L.rs = L.w + kU;      // sum of the (centred) reference over the align atoms: 3 floats (+ 1 pad)
  L.y = L.rs + 4;
  L.e = L.y + k * kU;
  L.feat = L.e + k * kU;              // 16-byte aligned: every term above is a multiple of 4 floats
  L.g = L.feat + kU * kImgP;
  L.total = L.g + k * kU * kImgP;
  return L;
}

// ------------------------------------------------------------------------------------------------------------------
// front
// ------------------------------------------------------------------------------------------------------------------
// NIT = ceil(N / 4) exactly (atoms per lane in the four-lanes-per-frame passes): every iteration but the last is complete, so
// only the last one carries the masks of the ragged end.  ALLAL: every feature atom is an align atom (n_align == n_rec).
template <int H, int NH, int NIT, bool ALLAL>
__global__ __launch_bounds__(512, 4) void ef16_front_kernel(cvf_mlp_desc mlp, const float* __restrict__ theta,
                                                             const float* __restrict__ packed, cvf_pp_desc pp,
                                                             const float* __restrict__ x, int64_t B,
                                                             const float* __restrict__ a, const float* __restrict__ w,
                                                             float* __restrict__ feat_tiled, float* __restrict__ y_tiled,
                                                             float* __restrict__ saved, float* __restrict__ q_tiled,
                                                             float* __restrict__ e_tiled, double* __restrict__ partial, int ns,
                                                             const float* __restrict__ x_lag, int64_t units_x) {
  constexpr int RT = Hid<H>::RT, NG = Hid<H>::NG, SMAX = 18, CTMAX = 5;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, nthreads = blockDim.x, nw = nthreads >> 6;
  // (the wave number through an SGPR: derived from threadIdx.x alone the compiler treats it - and every address formed
  //  with it, i.e. all of this net's weights and images - as lane-varying, in VGPR pairs)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int net = wave, k = mlp.n_nets, D = mlp.dims[0];
  // transfer-operator mode (x_lag != NULL): the grid holds the units of x, then the units of the lagged frames, whose tiles
  // follow the tiles of x in every tiled output; the block stops after y and the hand-off of the hidden activations
  const bool lagged = x_lag != nullptr && (int64_t)blockIdx.x >= units_x;   // (uniform)
  const int64_t unit = lagged ? (int64_t)blockIdx.x - units_x : (int64_t)blockIdx.x;   // unit within its frame set
  const int64_t tile = (unit >> 2) + (lagged ? (units_x >> 2) : 0);
  const int sub = (int)(unit & 3);
  if (lagged) x = x_lag;
  const int nc = pp.n_coord, nal = pp.n_align, N = pp.n_rec;
  const int stride = x_tile_stride(nc);
  const Front16Lds Lo = front16_lds(nc, nal, k);
  float* xt = lds;
  float* refL = lds + Lo.ref;
  float* aL = lds + Lo.a;
  float* auxL = lds + Lo.aux;
  float* wL = lds + Lo.w;
  float* rsL = lds + Lo.rs;
  float* yL = lds + Lo.y;
  float* eL = lds + Lo.e;
  float* featI = lds + Lo.feat;
  float* gI = lds + Lo.g + net * (kU * kImgP);
  const int f = lane >> 2, p = lane & 3;       // four-lanes-per-frame phases: frame f of the unit, part p
  const int col = lane & 15, q = lane >> 4;    // matrix-core phases: frame col of the unit, k-slot / row group q

  CVF_STAMP(20);
  // ---- stage the unit's coordinates (16 x nc floats, one contiguous run), the tables and the weights
  if (stride == nc && (unit + 1) * kU <= B && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    // the unit as it lies in memory (3N = 2 mod 4: plain copy, 16 x 3N floats = a whole number of 16-byte pieces; at most two
    // per thread here) - the general stager's index arithmetic (integer divisions by 3N) was a tenth of this kernel's
    // vector instructions
    const float4* src = reinterpret_cast<const float4*>(x + unit * (int64_t)(kU * nc));
    float4* dst = reinterpret_cast<float4*>(xt);
    const int n4 = (kU * nc) >> 2;
    for (int v = tid; v < n4; v += nthreads) dst[v] = src[v];
  } else {
    load_x_tile<6>(x, B, nc, unit, xt, tid, nthreads, kU);
  }
  for (int j = tid; j < 3 * nal + nc; j += nthreads) refL[j] = j < 3 * nal ? pp.ref_c[j] : (a != nullptr ? a[j - 3 * nal] : 0.0f);   // refL | aL
  if (tid < kU) {
    const int64_t frame = unit * kU + tid;
    wL[tid] = (w != nullptr && frame < B) ? w[frame] : 0.0f;      // frames past the batch replicate the last one with weight 0
  }
  // ---- the first layer's weights are requested here, behind the coordinates and the tables (vector memory returns in issue
  //      order: in front of them they delayed the staging by 7 k cycles): their round trip runs beside the barrier and the
  //      alignment (requested after the alignment, layer 0 began with a wait of ~2 k cycles per wave)
  const PackLayout L = pack_layout(H, NH, D);
  const URows pk = urows(packed + (int64_t)net * L.per_net, L.per_net, lane);   // this net's fragments (see URows)
  const int S = (D + 3) >> 2, CT = (D + 15) >> 4;
  float a0[SMAX][RT];
  float bias[NH][RT][4];
  auto request_layer0 = [&]() {
#pragma unroll
    for (int s = 0; s < SMAX; ++s) {
      const int se = s < S ? s : S - 1;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) a0[s][rt] = pk.ld(L.f0() + (se * RT + rt) * 64);   // (k-steps past S are skipped below)
    }
    load_hid_const_u<H>(urows(theta + mlp.b_off[net][0], H, q), bias[0]);
  };
  // (wave 0 asks after its alignment: the solve's fp64 state and these 36 + 8 registers do not fit 128 together, and
  //  a spilled fragment is stored behind a wait for ALL outstanding loads)
  if (wave != 0) request_layer0();
  __syncthreads();
  const float* my = xt + f * stride;
  CVF_STAMP(21);

  // ---- wave 0: centroid, covariance (this lane's quarter of the align atoms, fp64) and the rotation of the 16 frames
  if (wave == 0) {
    double acc[15];
#pragma unroll
    for (int i = 0; i < 15; ++i) acc[i] = 0.0;
#pragma unroll 2
    for (int b = p; b < nal; b += 4) {
      const double x0 = (double)my[3 * b], x1 = (double)my[3 * b + 1], x2 = (double)my[3 * b + 2];
      const double r0 = (double)refL[3 * b], r1 = (double)refL[3 * b + 1], r2 = (double)refL[3 * b + 2];
      acc[0] += x0; acc[1] += x1; acc[2] += x2;
      acc[3] = fma(x0, r0, acc[3]); acc[4] = fma(x0, r1, acc[4]); acc[5] = fma(x0, r2, acc[5]);
      acc[6] = fma(x1, r0, acc[6]); acc[7] = fma(x1, r1, acc[7]); acc[8] = fma(x1, r2, acc[8]);
      acc[9] = fma(x2, r0, acc[9]); acc[10] = fma(x2, r1, acc[10]); acc[11] = fma(x2, r2, acc[11]);
      acc[12] += r0; acc[13] += r1; acc[14] += r2;
    }
#pragma unroll
    for (int i = 0; i < 15; ++i) acc[i] = quad_sumd16(acc[i]);
    const double inv = fast_rcp((double)nal);
    const double cd[3] = {acc[0] * inv, acc[1] * inv, acc[2] * inv};
    double Hm[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) Hm[i][j] = fma(-cd[i], acc[12 + j], acc[3 + 3 * i + j]);
    KabschOut ko;
    kabsch_from_H(Hm, ko);
    const Centre c = centre_of(cd);
    const float av[kAuxP] = {ko.R[0], ko.R[1], ko.R[2], ko.R[3], ko.R[4], ko.R[5], ko.R[6], ko.R[7], ko.R[8], c.hi[0], c.hi[1], c.hi[2],
                             ko.Kinv[0], ko.Kinv[1], ko.Kinv[2], ko.Kinv[3], ko.Kinv[4], ko.Kinv[5], c.lo[0], c.lo[1], c.lo[2]};
#pragma unroll
    for (int i = 0; i < kAuxP; ++i)
      if ((i & 3) == p) auxL[f * kAuxP + i] = av[i];   // the four lanes of a frame hold the same record: each writes a quarter
    if (lane < 3) rsL[lane] = (float)(lane == 0 ? acc[12] : lane == 1 ? acc[13] : acc[14]);
    request_layer0();
  }
  lds_barrier();
  CVF_STAMP(22);

  // ---- aligned positions = features: the block's waves split the atoms (lane p of wave v: atoms p + 4 v + 4 nw i)
  {
    float R[9];
    const float* ar = auxL + f * kAuxP;
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = ar[i];
    Centre c;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      c.hi[i] = ar[9 + i];
      c.lo[i] = ar[18 + i];
    }
    // (LDS only here: the tiled copy for the backward kernel leaves at the end of the kernel - vector-memory operations
    //  return in issue order, so global stores at this point would sit in front of every weight fragment requested above)
    for (int at = p + 4 * wave; at < N; at += 4 * nw) {
      const V3 al = row_times(centred(my, at, c), R);
      float* fi = featI + f * kImgP + 3 * at;
      fi[0] = al.x;
      fi[1] = al.y;
      fi[2] = al.z;
    }
  }
  lds_barrier();
  CVF_STAMP(23);

  // ---- forward chain of this wave's net on the matrix cores: h_l = tanh(W_l h_{l-1} + b_l), 16 frames = the MFMA's N
  Vec<H, 1> h[NH];
  auto set_bias = [&](Vec<H, 1>& X, const float (&b)[RT][4]) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) X.v[rt][0][r] = b[rt][r];
  };
  set_bias(h[0], bias[0]);
  {
    float bf[SMAX];
    const float* fr = featI + col * kImgP;
#pragma unroll
    for (int s = 0; s < SMAX; ++s) {
      const int kf = 4 * (s < S ? s : S - 1) + q;
      bf[s] = fr[kf < D ? kf : D - 1];   // rows past D meet zero weights
    }
    HFrag<H> hf[NH > 1 ? NH - 1 : 1];
#pragma unroll
    for (int l = 1; l < NH; ++l) {
      load_hfrag_u<H>(hf[l - 1], pk, L.fh(l));
      load_hid_const_u<H>(urows(theta + mlp.b_off[net][l], H, q), bias[l]);
    }
#pragma unroll
    for (int s = 0; s < SMAX; ++s) {
      if (s < S) {   // wave-uniform
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) h[0].v[rt][0] = mfma4(a0[s][rt], bf[s], h[0].v[rt][0]);
      }
    }
    CVF_STAMP(24);
    tanh_inplace<H, 1>(h[0]);
#pragma unroll
    for (int l = 1; l < NH; ++l) {
      set_bias(h[l], bias[l]);
      hidden_mul<H, 1>(h[l], hf[l - 1], h[l - 1]);
      tanh_inplace<H, 1>(h[l]);
    }
  }
  float wl[RT][4];
  load_hid_const_u<H>(urows(theta + mlp.w_off[net][NH], H, q), wl);
  const float bL = theta[mlp.b_off[net][NH]];
  HFrag<H> tf[NH > 1 ? NH - 1 : 1];
#pragma unroll
  for (int l = 1; l < NH; ++l) load_hfrag_u<H>(tf[l - 1], pk, L.th(l));

  // hand-off to the backward kernel, per (tile, net): 2 NH vectors in the register layout both kernels use, as
  // [vector][group g][unit of the tile][lane] (every (vector, g, unit) one coalesced 256-byte row):
  //   h_1..h_NH | e_1..e_{NH-1} (e_l = W_{l+1}^T d_{l+1}: the d chain) | s = W_1 q (the tangent chain's first product)
  // (stored after g below: vector-memory operations return in issue order, and the fragment loads of the d chain and of g -
  //  which the compiler places just in time - must not queue behind 25 stores)
  const URows sv = urows(saved + (tile * k + net) * (int64_t)(kHand<NH>() * NG * 256) + sub * 64, kHand<NH>() * NG * 256 - sub * 64, lane);
  {
    float part = 0.0f;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) part = fmaf(wl[rt][r], h[NH - 1].v[rt][0][r], part);
    const float yv = sum_over_q(part) + bL;
    // (all four q groups hold the same value and store it to the same place: no lane-divergent branch around a store,
    //  behind which the compiler could no longer count the outstanding memory operations and would drain them all)
    yL[net * kU + col] = yv;
    y_tiled[(tile * k + net) * CVF_TILE + kU * sub + col] = yv;
  }
  CVF_STAMP(25);
  if (x_lag != nullptr) {   // transfer-operator mode: y, h_1..h_NH and the feature tile are all the backward pass needs
#pragma unroll
    for (int l = 0; l < NH; ++l)
#pragma unroll
      for (int g = 0; g < NG; ++g) sv.st((l * NG + g) * 256, h[l].v[g >> 2][0][g & 3]);
    float* ft = feat_tiled + tile * (int64_t)D * CVF_TILE + kU * sub + f;
    const float* fi = featI + f * kImgP;
    for (int j = p + 4 * wave; j < D; j += 4 * nw) ft[j * CVF_TILE] = fi[j];
    return;
  }
  // ---- d chain and g = W_1^T d_1 -> this wave's image [frame][feature]
  {
    // (requested behind the hand-off stores of h - vector-memory operations return in issue order - but the d chain below
    //  runs on fragments requested before them and covers that)
    float t0[CTMAX][NG];
#pragma unroll
    for (int rt = 0; rt < CTMAX; ++rt)
#pragma unroll
      for (int s = 0; s < NG; ++s) t0[rt][s] = pk.ld(L.t0() + ((rt < CT ? rt : CT - 1) * NG + s) * 64);
    Vec<H, 1> d;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float hv = h[NH - 1].v[rt][0][r];
        d.v[rt][0][r] = wl[rt][r] * (1.0f - hv * hv);
      }
    Vec<H, 1> ev[NH > 1 ? NH - 1 : 1];
#pragma unroll
    for (int l = NH - 1; l >= 1; --l) {
      init_bias<H, 1>(ev[l - 1], nullptr, q);
      hidden_mul<H, 1>(ev[l - 1], tf[l - 1], d);
      tangent_of<H, 1>(d, h[l - 1], ev[l - 1]);
    }
#pragma unroll
    for (int rt = 0; rt < CTMAX; ++rt) {
      if (rt < CT) {   // wave-uniform
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int s = 0; s < NG; ++s) acc = mfma4(t0[rt][s], d.v[s >> 2][0][s & 3], acc);
        // rows 16 rt + 4 q .. + 3 of column col: four consecutive features of one frame = one 16-byte write
        if (16 * rt + 4 * q < D) *reinterpret_cast<float4*>(gI + col * kImgP + 16 * rt + 4 * q) = float4{acc[0], acc[1], acc[2], acc[3]};
      }
    }
    // the hand-off rows h_1..h_NH and e_1..e_{NH-1}
#pragma unroll
    for (int l = 0; l < NH; ++l)
#pragma unroll
      for (int g = 0; g < NG; ++g) sv.st((l * NG + g) * 256, h[l].v[g >> 2][0][g & 3]);
#pragma unroll
    for (int l = 1; l < NH; ++l)
#pragma unroll
      for (int g = 0; g < NG; ++g) sv.st(((NH + l - 1) * NG + g) * 256, ev[l - 1].v[g >> 2][0][g & 3]);
  }
  CVF_STAMP(26);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the image is this wave's own: LDS keeps a wave's accesses in order

  float f0b[SMAX][RT];   // the first layer's fragments once more, for s = W_1 q after the passes
  // ---- q = J A J^T g and E with four lanes per frame (the three passes of cvf_metric.hpp; lane p of a frame takes the atoms
  // p, p + 4, ..).  NIT = ceil(N / 4) rounded up to 2 / 4 / 6 is a template parameter: the lane's g rows and centred
  // coordinates are read from LDS ONCE into registers (clamped index + mask for the ragged end, no branches), u = a .* G
  // replaces g in those registers and q is formed from them - one LDS round trip for the three passes instead of three.
  {
    float R[9];
    const float* ar = auxL + f * kAuxP;
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = ar[i];
    const Centre c = centre_of(ar[9], ar[10], ar[11]);
    float* Ul = gI + f * kImgP;
    V3 gv[NIT];
    // pass 1: sum_b g_b and M = sum_b (x_b - c) (x) g_b
    V3 gsum = v3(0.0f, 0.0f, 0.0f);
    Outer3 Mo = {{{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}}, {0.0f, 0.0f, 0.0f}};
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int at = p + 4 * it;
      asm volatile("" : "+v"(at));   // (opaque: the atom's addresses are formed here, not hoisted and kept across the passes)
      const bool last = it == NIT - 1;           // (compile-time: the only iteration that can run past the last atom)
      const int ac = (last && at >= N) ? N - 1 : at;
      const float lv = (last && at >= N) ? 0.0f : 1.0f;
      // (a lane past the last atom works on a DUPLICATE of atom N - 1 - masked out of every sum, but carried through so that
      //  it computes and stores the same u and q as that atom's owner: no lane-divergent branch around the stores)
      gv[it] = v3(Ul[3 * ac], Ul[3 * ac + 1], Ul[3 * ac + 2]);
      const V3 gm = last ? lv * gv[it] : gv[it];
      gsum = gsum + gm;
      outer_acc(Mo, centred(my, ac, c), gm);
    }
    CVF_STAMP(27);
    gsum = v3(quad_sumf16(gsum.x), quad_sumf16(gsum.y), quad_sumf16(gsum.z));
    const V3 sump = mat_times(R, gsum);   // sum_b R g_b
    float M[9];
    outer_to_array(Mo, M);
#pragma unroll
    for (int i = 0; i < 9; ++i) M[i] = quad_sumf16(M[i]);
    float T[9], Z[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) T[3 * i + j] = R[i] * M[j] + R[3 + i] * M[3 + j] + R[6 + i] * M[6 + j];
    float Kinv[6];   // (read where it is used, twice: six registers less across the atom loops)
#pragma unroll
    for (int i = 0; i < 6; ++i) Kinv[i] = ar[12 + i];
    const V3 s = sym_times(Kinv, v3(T[7] - T[5], T[2] - T[6], T[3] - T[1]));
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      Z[3 * i + 0] = R[3 * i + 1] * s.z - R[3 * i + 2] * s.y;
      Z[3 * i + 1] = -R[3 * i + 0] * s.z + R[3 * i + 2] * s.x;
      Z[3 * i + 2] = R[3 * i + 0] * s.y - R[3 * i + 1] * s.x;
    }
    const float inv_nal = 1.0f / (float)nal;
    const V3 shift = inv_nal * sump;
    // pass 2: G = R g (+ Z ref - shift on the align atoms), u = a .* G, E = u . G; u replaces g in the registers
    const MatCols Rc = mat_cols(R), Zc = mat_cols(Z);
    f2 E2 = {0.0f, 0.0f};
    float Ez = 0.0f;
    f2 usum_xy = {0.0f, 0.0f};
    float usum_z = 0.0f;
    Outer3 dHo = {{{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}}, {0.0f, 0.0f, 0.0f}};
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int at = p + 4 * it;
      asm volatile("" : "+v"(at));   // (opaque: the atom's addresses are formed here, not hoisted and kept across the passes)
      const bool last = it == NIT - 1;
      const int ac = (last && at >= N) ? N - 1 : at;
      const float lv = (last && at >= N) ? 0.0f : 1.0f;             // alive (not a duplicate): counts in the sums
      const float ma = ALLAL ? 1.0f : (ac < nal ? 1.0f : 0.0f);      // align atom: carries the rotation's and the centroid's derivative
      const int ar_ = ALLAL ? ac : (ac < nal ? ac : 0);
      V3 rf = v3(refL[3 * ar_], refL[3 * ar_ + 1], refL[3 * ar_ + 2]);
      if (!ALLAL) rf = ma * rf;
      f2 Gxy = ALLAL ? f2{-shift.x, -shift.y} : f2{-ma * shift.x, -ma * shift.y};
      float Gz = ALLAL ? -shift.z : -ma * shift.z;
      mat_times_acc(Rc, gv[it], Gxy, Gz);
      mat_times_acc(Zc, rf, Gxy, Gz);
      const f2 uxy = f2{aL[3 * ac], aL[3 * ac + 1]} * Gxy;
      const float uz = aL[3 * ac + 2] * Gz;
      gv[it] = v3(uxy.x, uxy.y, uz);
      const bool masked = last || !ALLAL;                            // (compile-time)
      const float m = ma * lv;
      const f2 uxm = masked ? m * uxy : uxy;
      const float uzm = masked ? m * uz : uz;
      E2 = fma2(last ? lv * uxy : uxy, Gxy, E2);
      Ez = fmaf(last ? lv * uz : uz, Gz, Ez);
      usum_xy += uxm; usum_z += uzm;
      outer_acc(dHo, v3(uxm.x, uxm.y, uzm), rf);
    }
    CVF_STAMP(28);
    const float E = quad_sumf16((E2.x + E2.y) + Ez);
    const V3 usum = v3(quad_sumf16(usum_xy.x), quad_sumf16(usum_xy.y), quad_sumf16(usum_z));
    const V3 rsum = v3(rsL[0], rsL[1], rsL[2]);   // sum of the reference over the align atoms (the fp32 residue of its centring)
    float dH[9];
    outer_to_array(dHo, dH);
#pragma unroll
    for (int i = 0; i < 9; ++i) dH[i] = quad_sumf16(dH[i]);
    eL[net * kU + f] = E;   // (the four lanes of a frame hold the same sum and store it to the same place)
    e_tiled[(tile * k + net) * CVF_TILE + kU * sub + f] = E;
    const V3 ubar = inv_nal * usum;
    dH[0] -= ubar.x * rsum.x; dH[1] -= ubar.x * rsum.y; dH[2] -= ubar.x * rsum.z;
    dH[3] -= ubar.y * rsum.x; dH[4] -= ubar.y * rsum.y; dH[5] -= ubar.y * rsum.z;
    dH[6] -= ubar.z * rsum.x; dH[7] -= ubar.z * rsum.y; dH[8] -= ubar.z * rsum.z;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) T[3 * i + j] = R[i] * dH[j] + R[3 + i] * dH[3 + j] + R[6 + i] * dH[6 + j];
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 6; ++i) Kinv[i] = ar[12 + i];
    const V3 om = sym_times(Kinv, v3(T[7] - T[5], T[2] - T[6], T[3] - T[1]));
    float dR[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      dR[3 * i + 0] = R[3 * i + 1] * om.z - R[3 * i + 2] * om.y;
      dR[3 * i + 1] = -R[3 * i + 0] * om.z + R[3 * i + 2] * om.x;
      dR[3 * i + 2] = R[3 * i + 0] * om.y - R[3 * i + 1] * om.x;
    }
    // pass 3: q_b = (u_b - ubar) R + (x_b - c) dR  -> in place of g, this wave's image (the B operand of s = W_1 q below)
    CVF_STAMP(31);
    const MatRows Rr = mat_rows(R), dRr = mat_rows(dR);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int at = p + 4 * it;
      asm volatile("" : "+v"(at));   // (opaque: the atom's addresses are formed here, not hoisted and kept across the passes)
      const int ac = (it == NIT - 1 && at >= N) ? N - 1 : at;
      f2 qxy = {0.0f, 0.0f};
      float qz = 0.0f;
      row_times_acc(Rr, gv[it] - ubar, qxy, qz);
      row_times_acc(dRr, centred(my, ac, c), qxy, qz);
      gv[it] = v3(qxy.x, qxy.y, qz);   // (kept: q leaves for global memory at the very end, behind every load of this kernel)
      Ul[3 * ac] = qxy.x;
      Ul[3 * ac + 1] = qxy.y;
      Ul[3 * ac + 2] = qz;
    }
    CVF_STAMP(32);
    asm volatile("" ::: "memory");   // (not earlier: 36 more live registers during the passes spill, and a spill's reload
                                     //  drains every outstanding store)
#pragma unroll
    for (int s_ = 0; s_ < SMAX; ++s_) {
      const int se = s_ < S ? s_ : S - 1;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) f0b[s_][rt] = pk.ld(L.f0() + (se * RT + rt) * 64);
    }
    // ---- s = W_1 q (what the backward kernel's tangent chain starts from, up to the per-frame factor 2 w dL/dE it only
    // knows after the batch sums are reduced): the first layer's fragments once more, q from this wave's image
    {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      Vec<H, 1> sv0;
      init_bias<H, 1>(sv0, nullptr, q);
      const float* qr = gI + col * kImgP;
#pragma unroll
      for (int s = 0; s < SMAX; ++s) {
        if (s < S) {   // wave-uniform
          const int kf = 4 * s + q;
          const float b = qr[kf < D ? kf : D - 1];
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) sv0.v[rt][0] = mfma4(f0b[s][rt], b, sv0.v[rt][0]);
        }
      }
#pragma unroll
      for (int g = 0; g < NG; ++g) sv.st(((2 * NH - 1) * NG + g) * 256, sv0.v[g >> 2][0][g & 3]);
    }
    CVF_STAMP(33);
    // ---- q -> the tiled hand-off (the first layer's weight-gradient operand of the backward kernel)
    float* qt = q_tiled + (tile * k + net) * (int64_t)D * CVF_TILE + kU * sub + f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int at = p + 4 * it;
      asm volatile("" : "+v"(at));
      const int ac = (it == NIT - 1 && at >= N) ? N - 1 : at;
      qt[(3 * ac) * CVF_TILE] = gv[it].x;
      qt[(3 * ac + 1) * CVF_TILE] = gv[it].y;
      qt[(3 * ac + 2) * CVF_TILE] = gv[it].z;
    }
  }
  CVF_STAMP(34);
  // ---- the feature tile for the backward kernel (the first layer's weight-gradient operand), from the LDS image
  {
    float* ft = feat_tiled + tile * (int64_t)D * CVF_TILE + kU * sub + f;
    const float* fi = featI + f * kImgP;
    for (int j = p + 4 * wave; j < D; j += 4 * nw) ft[j * CVF_TILE] = fi[j];
  }
  CVF_STAMP(29);
  if (partial == nullptr) return;   // (uniform) large batches: the caller reduces y / E with cvf_ef_stats
  lds_barrier();                    // y and E of every net are in LDS

  // ---- wave 0: this unit's row of the batch sums [W | S1(k) | S2(i<=j) | E(k)], fp64: lane = (statistic, frame) in rows of
  // 16 lanes, the 16 frames of a row added by the in-row DPP scan (fixed order); cvf_ef_stats_finish adds the units' rows
  if (wave == 0) {
    const int row = lane >> 4, fr = lane & 15;
    const double wb = (double)wL[fr];
    const int np = CVF_NPAIR(k);
    for (int t0 = 0; t0 < ns; t0 += 4) {
      const int t = t0 + row;
      double term = 0.0;
      if (t == 0) term = 1.0;
      else if (t <= k) term = (double)yL[(t - 1) * kU + fr];
      else if (t <= k + np) {
        int pi = t - 1 - k, i = 0;
        while (pi >= k - i) {
          pi -= k - i;
          ++i;
        }
        term = (double)yL[i * kU + fr] * (double)yL[(i + pi) * kU + fr];
      } else if (t < ns) term = (double)eL[(t - 1 - k - np) * kU + fr];
      double v = wb * term;
      v += dpp_movd<0x111, 0xf>(v);   // row_shr:1
      v += dpp_movd<0x112, 0xf>(v);   // row_shr:2
      v += dpp_movd<0x114, 0xf>(v);   // row_shr:4
      v += dpp_movd<0x118, 0xf>(v);   // row_shr:8 -> lane 15 of the row holds the sum of its 16 frames
      if (fr == 15 && t < ns) partial[t * (int64_t)gridDim.x + unit] = v;   // [statistic][unit]: coalesced for the finishing launch
    }
  }
  CVF_STAMP(30);
}

// ------------------------------------------------------------------------------------------------------------------
// back: parameter gradient of the loss given the coefficients d loss / d sums (what loss.backward() does at core.py:517).
// Block = (64-frame tile, net), four waves; wave w owns the unit w of the tile (frames 16 w .. 16 w + 15) for the
// register-resident chains and a share of every weight-gradient product, whose K dimension is the tile's 64 frames
// (operands transposed through LDS images [feature][frame], as in ef_bwd_mfma_kernel).
// Against that kernel: the forward chain, the d chain AND the first product of the tangent chain arrive from the front
// kernel (30 coalesced 256-byte rows per wave), every weight fragment and the first layer's feature / q operands are
// requested before the first matrix instruction - no global round trip is left inside the dependent chain, which at these
// batch sizes was two thirds of the old kernel's time (tools/ef16_probe.hip: d + tangent chains 32 k, reverse l=0 19 k
// of 69 k cycles per wave with just-in-time loads).
// ------------------------------------------------------------------------------------------------------------------
struct Back16Args {
  int k;
  int64_t B;
  int64_t n_tiles;
  int64_t T;              // transfer-operator mode: tiles 0..T-1 hold the frames, T..2T-1 their lagged partners
  const float* w_lag;     // ... and the partners' weights
};

// GEN: generator mode (tangent chain, second operands).  !GEN: transfer-operator mode - the plain backward pass of y and y'
// with the coefficients of the time-lagged loss (as ef_bwd_mfma_kernel's lag_idx > 0 branch), no tangent chain.
template <int H, int NH, bool MULTI, bool GEN>
__global__ __launch_bounds__(256, 4) void ef16_back_kernel(Back16Args args, cvf_mlp_desc mlp, const float* __restrict__ theta,
                                                            const float* __restrict__ packed, const float* __restrict__ w,
                                                            const float* __restrict__ feat, const float* __restrict__ y_tiled,
                                                            const float* __restrict__ q_tiled, const double* __restrict__ coef,
                                                            float* __restrict__ slab, int32_t* __restrict__ step,
                                                            const float* __restrict__ saved) {
  constexpr int RT = Hid<H>::RT, NG = Hid<H>::NG;
  constexpr int RTO = (H + 15) / 16;      // row tiles of an H-row image (natural order)
  constexpr int CTH = (H + 1 + 15) / 16;  // column tiles of [h ; 1]
  constexpr int NT = 256, WPB = 4;
  constexpr int kRows = 2 * H + 2 * (H + 1) + 16;   // packed images: reads past an image's rows meet finite values whose products are discarded
  __shared__ __attribute__((aligned(16))) float IMG[kRows * kPitch];
  extern __shared__ float GI[];  // MULTI (a block walks several tiles): its partial gradient of `net`, flat parameter order
  float* SA1 = IMG;
  float* SA2 = SA1 + H * kPitch;
  float* SB1 = SA2 + H * kPitch;
  float* SB2 = SB1 + (H + 1) * kPitch;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform values in SGPRs (see the front kernel)
  const int col = lane & 15, q = lane >> 4, row16 = col, r0 = 4 * q;
  const int fo = 16 * wave + col;   // this lane's frame of the tile
  const int net = blockIdx.y, k = args.k, D = mlp.dims[0];
  const int CT1 = (D + 1 + 15) / 16;
  const int gbase = mlp.w_off[net][0];
  const int gspan = mlp.b_off[net][NH] + 1 - gbase;
  float* out = slab + (int64_t)blockIdx.x * mlp.n_params + gbase;   // this block's slab row, this net's span

  for (int i = tid; i < kRows * kPitch; i += NT) IMG[i] = 0.0f;
  if (MULTI)
    for (int i = tid; i < gspan; i += NT) GI[i] = 0.0f;
  __syncthreads();
  if (tid < 64) SB1[H * kPitch + tid] = 1.0f;  // bias column of [h ; 1]  (row H of SB2 stays 0)

  const PackLayout L = pack_layout(H, NH, D);
  const URows pk = urows(packed + (int64_t)net * L.per_net, L.per_net, lane);   // this net's fragments (see URows)
  float wl[RT][4];
  load_hid_const_u<H>(urows(theta + mlp.w_off[net][NH], H, q), wl);
  const double gS1n = coef[net], gEtn = coef[k + k * k + net];
  const double gS1ln = GEN ? 0.0 : coef[2 * k + k * k + net], gS2ln = GEN ? 0.0 : coef[3 * k + k * k + net];
  const float one[1] = {1.0f};

  // a finished 16x16 tile of layer `l` (rows = outputs, columns = inputs + bias).  Every tile of the gradient is produced
  // by exactly one wave, so a block that handles ONE tile of frames stores it straight into its slab row; a block that
  // walks several accumulates in the LDS image and flushes at the end.
  // (direct stores go through a buffer descriptor of the row's span: entries outside the layer get an offset past its end and
  //  are dropped by the hardware - no lane-divergent branch around the stores, so the compiler can count them when it waits
  //  for the weight fragments requested before them instead of waiting for everything; and the row offset is made opaque, or
  //  the four offsets of every call site are hoisted out of the layer loops, spilled, and reloaded behind a wait for ALL
  //  outstanding memory operations - i.e. for the previous store's completion, eight times per tile)
  const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, gspan * 4, 0x00020000);
  auto emit_tile = [&](int l, int n_out, int n_in, int rt, int ct, const f32x4& acc) {
    const int wo = mlp.w_off[net][l] - gbase, bo = mlp.b_off[net][l] - gbase;
    const int i = 16 * ct + row16;
    int ob = 16 * rt + r0;
    asm volatile("" : "+v"(ob));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = ob + r;
      const bool isw = o < n_out && i < n_in, isb = o < n_out && i == n_in;
      const int idx = isw ? wo + o * n_in + i : bo + (o < n_out ? o : 0);
      const float av = acc[r];   // (by value: __builtin_bit_cast applied to the vector-element lvalue acc[r] read element 0 four times)
      if (MULTI) {
        if (isw || isb) GI[idx] += av;
      } else {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, av), out_rs, (isw || isb) ? idx * 4 : 0x7ffffff0, 0, 0);
      }
    }
  };
  // one 16x16 tile of  A1 B1^T + A2 B2^T  over the tile's 64 frames, operands in LDS images (eight 16-byte reads at a time)
  auto outer2 = [&](const float* A1, const float* B1, const float* A2, const float* B2, int rt, int ct) {
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    acc = outer_half(A1, B1, rt, ct, lane, acc);
    if constexpr (GEN) acc = outer_half(A2, B2, rt, ct, lane, acc);
    return acc;
  };

  for (int64_t tile = blockIdx.x; tile < args.n_tiles; tile += gridDim.x) {
    CVF_STAMP(8);
    // ---- everything the chains need from memory is requested here
    const bool lagged = !GEN && tile >= args.T;              // (uniform) a tile of lagged partners
    const int64_t t0 = lagged ? tile - args.T : tile;        // the tile of the frames themselves
    const int64_t frame = t0 * CVF_TILE + fo;
    const bool valid = frame < args.B;
    const float wraw = w[valid ? frame : args.B - 1];
    float yb[CVF_MAX_NETS];
#pragma unroll
    for (int j = 0; j < CVF_MAX_NETS; ++j) yb[j] = y_tiled[(t0 * k + (j < k ? j : k - 1)) * CVF_TILE + fo];
    float ylag = 0.0f, wlraw = 0.0f;
    if constexpr (!GEN) {
      ylag = y_tiled[((args.T + t0) * k + net) * CVF_TILE + fo];
      wlraw = args.w_lag[valid ? frame : args.B - 1];
    }
    const URows sv = urows(saved + (tile * k + net) * (int64_t)(kHand<NH>() * NG * 256) + wave * 64, kHand<NH>() * NG * 256 - wave * 64, lane);
    Vec<H, 1> h[NH], e[NH > 1 ? NH - 1 : 1], t[NH];
#pragma unroll
    for (int l = 0; l < NH; ++l) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) h[l].v[rt][0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int g = 0; g < NG; ++g) h[l].v[g >> 2][0][g & 3] = sv.ld((l * NG + g) * 256);
    }
    if constexpr (GEN) {
#pragma unroll
      for (int l = 0; l + 1 < NH; ++l) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) e[l].v[rt][0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int g = 0; g < NG; ++g) e[l].v[g >> 2][0][g & 3] = sv.ld(((NH + l) * NG + g) * 256);
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) t[0].v[rt][0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int g = 0; g < NG; ++g) t[0].v[g >> 2][0][g & 3] = sv.ld(((2 * NH - 1) * NG + g) * 256);
    }
    // the tangent chain's weight fragments W_l, l = 2..NH, with the same round trip (the hbar chain's W_l^T are requested at
    // the top of each reverse step, a phase of outer products ahead of their use)
    HFrag<H> ffr[NH > 1 ? NH - 1 : 1];
    if constexpr (GEN) {
#pragma unroll
      for (int l = 1; l < NH; ++l) load_hfrag_u<H>(ffr[l - 1], pk, L.fh(l));
    }
    const float* f_tile = feat + tile * (int64_t)D * CVF_TILE;
    const float* q_tile = GEN ? q_tiled + (tile * k + net) * (int64_t)D * CVF_TILE : nullptr;
    // ---- per-frame coefficients
    const float wb = valid ? wraw : 0.0f;
    float alpha, gamma = 0.0f;
    float ynet = 0.0f;   // this net's y of the frame (selected, not indexed: yb[] lives in registers)
#pragma unroll
    for (int j = 0; j < CVF_MAX_NETS; ++j)
      if (j == net) ynet = yb[j];
    if (GEN || !lagged) {
      double a = gS1n;
#pragma unroll
      for (int j = 0; j < CVF_MAX_NETS; ++j)
        if (j < k) a += (j == net ? 2.0 : 1.0) * coef[k + net * k + j] * (double)yb[j];
      alpha = (float)((double)wb * a);
      if constexpr (GEN) gamma = (float)(2.0 * (double)wb * gEtn);
      else alpha = (float)((double)wb * a - 2.0 * (double)wb * gEtn * ((double)ylag - (double)ynet));   // ... and d/dy of gT sum w (y' - y)^2
    } else {   // a lagged partner: d/dy' of the primed sums and of gT sum w (y' - y)^2
      const double wlg = valid ? (double)wlraw : 0.0;
      alpha = (float)(wlg * (gS1ln + 2.0 * gS2ln * (double)ylag) + 2.0 * (double)wb * gEtn * ((double)ylag - (double)ynet));
    }
    CVF_STAMP(9);
    if constexpr (GEN) {
      // ---- tangent chain: t_1 = gamma s, t_l = W_l tdot_{l-1}
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) t[0].v[rt][0][r] *= gamma;
#pragma unroll
      for (int l = 1; l < NH; ++l) {
        Vec<H, 1> td;
        tangent_of<H, 1>(td, h[l - 1], t[l - 1]);
        init_bias<H, 1>(t[l], nullptr, q);
        hidden_mul<H, 1>(t[l], ffr[l - 1], td);
      }
    }
    CVF_STAMP(11);
    // ---- last layer (1 x H):  W_L += sum alpha h_{NH} + tdot_{NH} ; b_L += sum alpha
    {
      if (q == 0) {
        SA1[fo] = alpha;
        if constexpr (GEN) SA2[fo] = 1.0f;
      }
      store_image<H, 1, false>(SB1, h[NH - 1], one, lane, fo);
      if constexpr (GEN) {
        Vec<H, 1> td;
        tangent_of<H, 1>(td, h[NH - 1], t[NH - 1]);
        store_image<H, 1, false>(SB2, td, one, lane, fo);
      }
      __syncthreads();
      for (int ct = wave; ct < CTH; ct += WPB) {
        const f32x4 acc = outer2(SA1, SB1, SA2, SB2, 0, ct);
        if (q == 0) {  // output row 0 lives in register 0 of lanes 0..15
          const int wo = mlp.w_off[net][NH] - gbase, bo = mlp.b_off[net][NH] - gbase;
          const int i = 16 * ct + row16;
          if (i <= H) {
            const int idx = i < H ? wo + i : bo;
            if (MULTI) GI[idx] += acc[0];
            else out[idx] = acc[0];
          }
        }
      }
      __syncthreads();
    }
    CVF_STAMP(12);
    // ---- reverse sweep
    Vec<H, 1> hbar;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) hbar.v[rt][0][r] = alpha * wl[rt][r];
    // Every step's operands from memory (the transposed weights of a hidden layer; the feature / q rows of the first) are
    // requested one step ahead, BEFORE the step's slab stores: vector memory returns in issue order, and requests issued
    // behind the stores (at the top of the next step) waited for the stores' acknowledgements first.
    HFrag<H> tfl;
    float4 bA[4], bB[4];
    auto request = [&](float4 (&dst)[4], const float* src_tile, int ct) {
      const int i = 16 * ct + row16;
      const float4* p = reinterpret_cast<const float4*>(src_tile + (int64_t)(i < D ? i : D - 1) * CVF_TILE + 4 * q);
#pragma unroll
      for (int j = 0; j < 4; ++j) dst[j] = p[4 * j];
    };
    auto request_step = [&](int l) {   // l: compile-time after unrolling
      if (l > 0) {
        load_hfrag_u<H>(tfl, pk, L.th(l));
      } else {
        request(bA, f_tile, wave);
        if constexpr (GEN) request(bB, q_tile, wave);
      }
    };
    request_step(NH - 1);
#pragma unroll
    for (int l = NH - 1; l >= 0; --l) {
      CVF_STAMP(13 + (NH - 1 - l));
      // the first layer's B operands come straight from memory (the feature tile + ones row, then q): wave w owns column
      // tile w; k-slot kq of k-step (j, c) is frame 16 j + 4 kq + c, so a lane's sixteen values of one operand row are four
      // 16-byte loads.  The [f ; 1] rows are requested here, q's after the barrier, behind the matrix instructions of the first half.
      Vec<H, 1> zbar, dl;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float hv = h[l].v[rt][0][r];
          const float om = 1.0f - hv * hv;
          if constexpr (GEN) {
            const float ev = (l == NH - 1) ? wl[rt][r] : e[l < NH - 1 ? l : 0].v[rt][0][r];
            const float hb = fmaf(-2.0f * hv * t[l].v[rt][0][r], ev, hbar.v[rt][0][r]);
            dl.v[rt][0][r] = ev * om;
            zbar.v[rt][0][r] = om * hb;
          } else {
            zbar.v[rt][0][r] = om * hbar.v[rt][0][r];
          }
        }
      store_image<H, 1, false>(SA1, zbar, one, lane, fo);
      if constexpr (GEN) {
        const float sc[1] = {l == 0 ? gamma : 1.0f};
        store_image<H, 1, true>(SA2, dl, sc, lane, fo);
      }
      if (l > 0) {
        store_image<H, 1, false>(SB1, h[l - 1], one, lane, fo);
        if constexpr (GEN) {
          Vec<H, 1> td;
          tangent_of<H, 1>(td, h[l - 1], t[l - 1]);
          store_image<H, 1, false>(SB2, td, one, lane, fo);
        }
        __syncthreads();
        // hbar_{l-1} = W_l^T zbar_l  (registers), then the next step's requests, then this step's tiles
        init_bias<H, 1>(hbar, nullptr, q);
        hidden_mul<H, 1>(hbar, tfl, zbar);
        request_step(l - 1);
        for (int pr = wave; pr < RTO * CTH; pr += WPB) {
          const int rt = pr / CTH, ct = pr - rt * CTH;
          emit_tile(l, H, H, rt, ct, outer2(SA1, SB1, SA2, SB2, rt, ct));
        }
        __syncthreads();
      } else {
        __syncthreads();
        // one half of the contraction of column tile `ct` for the row tiles rt0, rt0 + rstep, ..: A rows from the LDS image
        // `SA`, B rows in registers; columns: features, then the ones (bias) column when `ones`, zeros past it
        // (no MFMA under lane-divergent control flow: operand values are selected per lane, the MFMAs are uniform)
        auto half0 = [&](f32x4 (&acc)[RTO], const float* SA, const float4 (&b)[4], int ct, int rt0, int rstep, bool ones) {
          const int i = 16 * ct + row16;
          const float pad = (ones && i == D) ? 1.0f : 0.0f;
#pragma unroll
          for (int rt = 0; rt < RTO; ++rt) {
            if (rt >= rt0 && (rt - rt0) % rstep == 0) {
              const float4* a1 = reinterpret_cast<const float4*>(SA + (16 * rt + row16) * kPitch + 4 * q);
              float4 av[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) av[j] = a1[4 * j];
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                acc[rt] = mfma4(av[j].x, i < D ? b[j].x : pad, acc[rt]);
                acc[rt] = mfma4(av[j].y, i < D ? b[j].y : pad, acc[rt]);
                acc[rt] = mfma4(av[j].z, i < D ? b[j].z : pad, acc[rt]);
                acc[rt] = mfma4(av[j].w, i < D ? b[j].w : pad, acc[rt]);
              }
            }
          }
        };
        // column tiles beyond the first four (a fifth, ragged one for D = 66: two features and the bias column) are dealt
        // by (column tile, row tile) pairs: pair p -> wave p % 4
        const int extra = (CT1 - WPB) * RTO;
        f32x4 acc[RTO];
#pragma unroll
        for (int rt = 0; rt < RTO; ++rt) acc[rt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (wave < CT1) {   // wave-uniform
          half0(acc, SA1, bA, wave, 0, 1, true);
          if (wave < extra) request(bA, f_tile, WPB + wave / RTO);   // the extra pair's rows, behind the second half
          if constexpr (GEN) half0(acc, SA2, bB, wave, 0, 1, false);
#pragma unroll
          for (int rt = 0; rt < RTO; ++rt) emit_tile(0, H, D, rt, wave, acc[rt]);
        }
        for (int pr = wave; pr < extra; pr += WPB) {
          const int ct = WPB + pr / RTO, rt = pr % RTO;
          if (pr != wave || wave >= CT1) request(bA, f_tile, ct);
          if constexpr (GEN) request(bB, q_tile, ct);
#pragma unroll
          for (int r_ = 0; r_ < RTO; ++r_) acc[r_] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
          half0(acc, SA1, bA, ct, rt, RTO, true);
          if constexpr (GEN) half0(acc, SA2, bB, ct, rt, RTO, false);
#pragma unroll
          for (int r_ = 0; r_ < RTO; ++r_)
            if (r_ == rt) emit_tile(0, H, D, r_, ct, acc[r_]);
        }
        __syncthreads();
      }
    }
  }
  CVF_STAMP(17);
  if (MULTI) {   // flush this block's partial gradient of `net` into its slab row
    __syncthreads();
    for (int i = tid; i < gspan; i += NT) out[i] = GI[i];
  }
  // one gradient per optimiser step: advance the step counter read by the Adam that follows
  if (step != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *step += 1;
  CVF_STAMP(18);
}

bool ef16_shape(const cvf_mlp_desc* m, int* H, int* NH) {
  if (m->n_layers < 2 || m->n_layers > 4 || m->dims[m->n_layers] != 1) return false;
  *H = m->dims[1];
  *NH = m->n_layers - 1;
  for (int l = 1; l < m->n_layers; ++l)
    if (m->dims[l] != *H) return false;
  for (int l = 0; l < m->n_layers; ++l)
    if (m->act[l] != (l + 1 < m->n_layers ? 1 : 0)) return false;
  return true;
}

template <class F>
bool ef16_dispatch(int H, int NH, F&& f) {
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
#undef EF_CASE
  return false;
}

}  // namespace

int cvf_ef_stats_finish_impl(const cvf_ef_cfg* cfg, int n_rows, int stat_major, const double* partial, double* stats,
                             double* loss_vec, double* coef, hipStream_t s);

extern "C" int cvf_ef16_supported(const cvf_mlp_desc* mlp, const cvf_pp_desc* pp) {
  int H, NH;
  if (!mlp || !pp || !ef16_shape(mlp, &H, &NH) || getenv("CVF_NO_EF16")) return 0;
  if (!ef16_dispatch(H, NH, [](auto, auto) {})) return 0;
  const int fast = CVF_PP_ALIGN_CONTIG | CVF_PP_PURE_POSITION;
  if (pp->mode != CVF_PP_ALIGN || pp->align_w || (pp->flags & fast) != fast || pp->n_align > pp->n_rec || pp->n_align < 3) return 0;
  if (pp->d_r != 3 * pp->n_rec || pp->d_r != mlp->dims[0] || pp->d_r > 72 || pp->n_coord > 192 || pp->n_coord < pp->d_r) return 0;
  if (mlp->n_nets < 1 || mlp->n_nets > CVF_MAX_NETS) return 0;
  return (size_t)front16_lds(pp->n_coord, pp->n_align, mlp->n_nets).total * sizeof(float) <= 64 * 1024;
}

// rows of per-unit batch sums the front launch leaves in `scratch` (+ room for the two-stage fallback of large batches)
extern "C" int64_t cvf_ef16_scratch_doubles(int64_t B, int k) {
  const int64_t units = 4 * cvf_ntiles(B);
  const int64_t rows = units <= kMaxRows16 ? units : 0;
  return rows * cvf_ef_nstats(k, 0) + cvf_ef_stats_scratch_doubles(k, 0);
}

// units whose rows of batch sums cvf_ef16_front leaves in `scratch` for cvf_ef16_finish (0: the batch is too large for one
// finishing launch - give cvf_ef16_front `stats` and it runs the two-stage reduction itself)
extern "C" int64_t cvf_ef16_rows(int64_t B) {
  const int64_t units = 4 * cvf_ntiles(B);
  return units <= kMaxRows16 ? units : 0;
}
extern "C" int cvf_ef16_finish(const cvf_ef_cfg* cfg, int64_t B, const double* scratch, double* stats, double* loss_vec, double* coef,
                               void* stream) {
  CVF_REQUIRE(cfg && scratch && stats && cvf_ef16_rows(B) > 0, "cvf_ef16_finish: bad argument");
  CVF_REQUIRE(loss_vec == nullptr || coef != nullptr, "cvf_ef16_finish: loss_vec without coef");
  return cvf_ef_stats_finish_impl(cfg, (int)cvf_ef16_rows(B), 1, scratch, stats, loss_vec, coef, (hipStream_t)stream);
}

extern "C" int64_t cvf_ef16_saved_floats(const cvf_mlp_desc* mlp, int64_t n_tiles) {
  int H, NH;
  if (!mlp || !ef16_shape(mlp, &H, &NH)) return 0;
  return n_tiles * mlp->n_nets * (2 * NH) * (int64_t)((H + 3) / 4) * 256;
}

extern "C" int cvf_ef16_front(const cvf_mlp_desc* mlp, const float* theta, const float* packed, float* feat_tiled,
                              const cvf_pp_desc* pp, const float* x, int64_t B, const float* a, float* y_tiled, float* saved,
                              float* q_tiled, float* e_tiled, const cvf_ef_cfg* cfg, const float* w, double* scratch, double* stats,
                              double* loss_vec, double* coef, void* stream) {
  CVF_REQUIRE(cvf_ef16_supported(mlp, pp), "cvf_ef16_front: shape not covered (cvf_ef16_supported() == 0)");
  CVF_REQUIRE(theta && packed && feat_tiled && x && a && y_tiled && saved && q_tiled && e_tiled && cfg && w && scratch && B > 0,
              "cvf_ef16_front: bad argument");
  CVF_REQUIRE(stats != nullptr || cvf_ef16_rows(B) > 0, "cvf_ef16_front: stats == NULL (rows left for cvf_ef16_finish) needs cvf_ef16_rows(B) > 0");
  CVF_REQUIRE(cfg->k == mlp->n_nets && cfg->lag_idx == 0, "cvf_ef16_front: generator mode only, cfg.k must equal the number of nets");
  CVF_REQUIRE(loss_vec == nullptr || coef != nullptr, "cvf_ef16_front: loss_vec without coef");
  int H, NH;
  ef16_shape(mlp, &H, &NH);
  const int k = mlp->n_nets;
  const int64_t T = cvf_ntiles(B), units = 4 * T;
  const int ns = cvf_ef_nstats(k, 0);
  const bool rows = units <= kMaxRows16;
  const size_t lds = (size_t)front16_lds(pp->n_coord, pp->n_align, k).total * sizeof(float);
  ef16_dispatch(H, NH, [&](auto h_, auto nh_) {
    constexpr int kH = decltype(h_)::value, kNH = decltype(nh_)::value;
    auto go = [&](auto kernel) {
      if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(kernel, dim3((unsigned)units), dim3(64 * k), lds, (hipStream_t)stream, *mlp, theta, packed, *pp, x, B, a, w,
                         feat_tiled, y_tiled, saved, q_tiled, e_tiled, rows ? scratch : nullptr, ns, (const float*)nullptr, units);
    };
    const int nit = (pp->n_rec + 3) / 4;   // atoms per lane in the four-lanes-per-frame passes (1..6: d_r <= 72)
    const bool allal = pp->n_align == pp->n_rec;
#define EF16_GO(NIT_)                                                  \
    case NIT_:                                                          \
      if (allal) go(ef16_front_kernel<kH, kNH, NIT_, true>);            \
      else go(ef16_front_kernel<kH, kNH, NIT_, false>);                 \
      break;
    switch (nit) {
      EF16_GO(1) EF16_GO(2) EF16_GO(3) EF16_GO(4) EF16_GO(5)
      default:
        if (allal) go(ef16_front_kernel<kH, kNH, 6, true>);
        else go(ef16_front_kernel<kH, kNH, 6, false>);
    }
#undef EF16_GO
  });
  int rc = cvf_check_launch("ef16_front_kernel");
  if (rc || stats == nullptr) return rc;   // stats == NULL: the caller adds the units' rows itself (cvf_ef16_finish)
  if (rows) return cvf_ef_stats_finish_impl(cfg, (int)units, 1, scratch, stats, loss_vec, coef, (hipStream_t)stream);
  return cvf_ef_stats(cfg, B, w, y_tiled, e_tiled, nullptr, nullptr, scratch, stats, loss_vec, coef, stream);
}

extern "C" int64_t cvf_ef16_backward_slab_rows(int64_t n_tiles) { return n_tiles < 1024 ? n_tiles : 1024; }

static int ef16_backward_impl(const cvf_ef_cfg* cfg, const cvf_mlp_desc* mlp, const float* theta, const float* packed,
                              int64_t B, const float* w, const float* w_lag, const float* feat_tiled, const float* y_tiled,
                              const float* q_tiled, const double* coef, float* slab, int32_t* step_count, const float* saved,
                              void* stream) {
  const bool transfer = cfg != nullptr && cfg->lag_idx > 0;
  CVF_REQUIRE(cfg && mlp && theta && packed && w && feat_tiled && y_tiled && (transfer ? w_lag != nullptr : q_tiled != nullptr) && coef &&
              slab && saved && B > 0, "cvf_ef16_backward: bad argument");
  CVF_REQUIRE(cfg->k == mlp->n_nets, "cvf_ef16_backward: cfg.k must equal the number of nets");
  int H, NH;
  CVF_REQUIRE(ef16_shape(mlp, &H, &NH), "cvf_ef16_backward: unsupported net shape");
  CVF_REQUIRE(mlp->dims[0] <= 8 * 16 - 1, "cvf_ef16_backward: first layer wider than the column-tile schedule covers");
  // the LDS gradient image relies on each net's parameters being one contiguous run of the flat buffer
  const int span = mlp->b_off[0][NH] + 1 - mlp->w_off[0][0];
  int covered = 0;
  for (int n = 0; n < mlp->n_nets; ++n) {
    CVF_REQUIRE(mlp->b_off[n][NH] + 1 - mlp->w_off[n][0] == span, "cvf_ef16_backward: nets are not laid out contiguously");
    for (int l = 0; l <= NH; ++l)
      CVF_REQUIRE(mlp->w_off[n][l] >= mlp->w_off[n][0] && mlp->b_off[n][l] < mlp->w_off[n][0] + span,
                  "cvf_ef16_backward: nets are not laid out contiguously");
    covered += span;
  }
  CVF_REQUIRE(covered == mlp->n_params, "cvf_ef16_backward: flat buffer holds parameters outside the nets");
  Back16Args a;
  a.k = cfg->k;
  a.B = B;
  a.T = cvf_ntiles(B);
  a.n_tiles = transfer ? 2 * a.T : a.T;
  a.w_lag = w_lag;
  const int64_t G = cvf_ef16_backward_slab_rows(a.n_tiles);
  const bool launched = ef16_dispatch(H, NH, [&](auto h_, auto nh_) {
    constexpr int kH = decltype(h_)::value, kNH = decltype(nh_)::value;
    auto go = [&](auto kernel, size_t lds) {
      hipLaunchKernelGGL(kernel, dim3((unsigned)G, cfg->k), dim3(256), lds, (hipStream_t)stream, a, *mlp, theta, packed, w, feat_tiled,
                         y_tiled, q_tiled, coef, slab, step_count, saved);
    };
    const size_t gi = (size_t)span * sizeof(float);
    // a.n_tiles > G: blocks walk several tiles (partial gradient in LDS, flushed once); else one tile per block, every
    // gradient tile goes straight to the block's slab row
    if (transfer) {
      if (a.n_tiles > G) go(ef16_back_kernel<kH, kNH, true, false>, gi);
      else go(ef16_back_kernel<kH, kNH, false, false>, 0);
    } else {
      if (a.n_tiles > G) go(ef16_back_kernel<kH, kNH, true, true>, gi);
      else go(ef16_back_kernel<kH, kNH, false, true>, 0);
    }
  });
  CVF_REQUIRE(launched, "cvf_ef16_backward: no kernel instance for hidden width %d x %d layers", H, NH);
  return cvf_check_launch("ef16_back_kernel");
}

extern "C" int cvf_ef16_backward(const cvf_ef_cfg* cfg, const cvf_mlp_desc* mlp, const float* theta, const float* packed,
                                 int64_t B, const float* w, const float* feat_tiled, const float* y_tiled, const float* q_tiled,
                                 const double* coef, float* slab, int32_t* step_count, const float* saved, void* stream) {
  CVF_REQUIRE(cfg && cfg->lag_idx == 0, "cvf_ef16_backward: generator mode (transfer-operator mode: cvf_ef16_backward_transfer)");
  return ef16_backward_impl(cfg, mlp, theta, packed, B, w, nullptr, feat_tiled, y_tiled, q_tiled, coef, slab, step_count, saved, stream);
}

// ------------------------------------------------------------------------------------------------------------------
// transfer-operator mode (lag_tau > 0; core.py:403,414,420-431,440): y on the frames and on their lagged partners, then the
// backward pass of both with the coefficients of the time-lagged loss.  Same kernels, same hand-off layout; the front
// kernel stops after y (no derivative passes), the backward kernel is compiled without the tangent chain.
// ------------------------------------------------------------------------------------------------------------------
extern "C" int cvf_ef16_front_transfer(const cvf_mlp_desc* mlp, const float* theta, const float* packed, float* feat_tiled,
                                       const cvf_pp_desc* pp, const float* x, const float* x_lag, int64_t B, float* y_tiled,
                                       float* saved, void* stream) {
  CVF_REQUIRE(cvf_ef16_supported(mlp, pp), "cvf_ef16_front_transfer: shape not covered (cvf_ef16_supported() == 0)");
  CVF_REQUIRE(theta && packed && feat_tiled && x && x_lag && y_tiled && saved && B > 0, "cvf_ef16_front_transfer: bad argument");
  int H, NH;
  ef16_shape(mlp, &H, &NH);
  const int k = mlp->n_nets;
  const int64_t T = cvf_ntiles(B), units = 4 * T;
  CVF_REQUIRE(2 * units < (int64_t)1 << 31, "cvf_ef16_front_transfer: batch too large for one launch");
  const size_t lds = (size_t)front16_lds(pp->n_coord, pp->n_align, k).total * sizeof(float);
  ef16_dispatch(H, NH, [&](auto h_, auto nh_) {
    constexpr int kH = decltype(h_)::value, kNH = decltype(nh_)::value;
    // (the passes' template parameters do not matter here - the block leaves before them: one instance serves)
    auto kernel = ef16_front_kernel<kH, kNH, 6, true>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kernel, dim3((unsigned)(2 * units)), dim3(64 * k), lds, (hipStream_t)stream, *mlp, theta, packed, *pp, x, B,
                       (const float*)nullptr, (const float*)nullptr, feat_tiled, y_tiled, saved, (float*)nullptr, (float*)nullptr,
                       (double*)nullptr, 0, x_lag, units);
  });
  return cvf_check_launch("ef16_front_kernel");
}

extern "C" int cvf_ef16_backward_transfer(const cvf_ef_cfg* cfg, const cvf_mlp_desc* mlp, const float* theta, const float* packed,
                                          int64_t B, const float* w, const float* w_lag, const float* feat_tiled,
                                          const float* y_tiled, const double* coef, float* slab, int32_t* step_count,
                                          const float* saved, void* stream) {
  CVF_REQUIRE(cfg && cfg->lag_idx > 0, "cvf_ef16_backward_transfer: transfer-operator mode (cfg.lag_idx > 0)");
  return ef16_backward_impl(cfg, mlp, theta, packed, B, w, w_lag, feat_tiled, y_tiled, nullptr, coef, slab, step_count, saved, stream);
}
