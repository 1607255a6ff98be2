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
#include "ef16_common.hpp"
#include "cvf_p2p.hpp"

namespace {
template <int H, int NH, int NIT, bool ALLAL>
__global__ __launch_bounds__(1024, 4) void ef16_front_kernel(cvf_mlp_desc mlp, const float* __restrict__ theta,
                                                             const float* __restrict__ packed, cvf_pp_desc pp,
                                                             const float* __restrict__ x, int64_t B,
                                                             const float* __restrict__ a, const float* __restrict__ w,
                                                             float* __restrict__ feat_tiled, float* __restrict__ y_tiled,
                                                             float* __restrict__ saved, float* __restrict__ q_tiled,
                                                             float* __restrict__ e_tiled, double* __restrict__ partial, int launch,
                                                             const float* __restrict__ x_lag, int64_t units_x,
                                                             const float* __restrict__ w_lag) {
  constexpr int RT = Hid<H>::RT, NG = Hid<H>::NG, SMAX = 18, CTMAX = 5;
  // NIT == 0: the TRANSFER-OPERATOR instance (cvf_ef16_front_transfer) - the block leaves after y and the hand-off of the hidden
  // activations, and everything behind that point is compiled out: no g images in LDS (11 KB per block instead of 26) and fewer
  // registers, so that the 2 x units of the frames and their lagged partners are resident in ONE round (the generator instance
  // held them in two: 5 blocks per CU by registers and LDS)
  constexpr bool kTransfer = NIT == 0;
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  // `launch` = units per workgroup | paired << 8.  Several units per workgroup (a developer switch, off by default - measured
  // slower, see ef16_units_per_wg): a workgroup of upb k waves takes upb consecutive units, each on its own k waves and its own
  // copy of the LDS layout, exactly as upb workgroups of one unit would - the units only share the workgroup's barriers.
  const int upb = launch & 0xff;
  const int k = mlp.n_nets, D = mlp.dims[0];
  // (the wave number through an SGPR: derived from threadIdx.x alone the compiler treats it - and every address formed
  //  with it, i.e. all of this net's weights and images - as lane-varying, in VGPR pairs)
  const int wave_b = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int usub = upb > 1 ? wave_b / k : 0;                 // this wave's unit within the workgroup (wave-uniform)
  const int wave = wave_b - usub * k;
  const int tid = (int)threadIdx.x - usub * (64 * k), lane_in = tid & 63;
  const int nthreads = upb > 1 ? 64 * k : (int)blockDim.x, nw = nthreads >> 6;
  const int net = wave;
  const int64_t ublock = (int64_t)blockIdx.x * upb + usub;   // what blockIdx.x is with one unit per workgroup
  // transfer-operator mode (x_lag != NULL), two launch forms:
  //   PAIRED (launch bit 8): the block runs unit u of the frames and then, with the same waves, unit u of their lagged
  //     partners, so that y and y' of the same 16 frame indices meet in one
  //     block and wave 0 can form the unit's row of the TIME-LAGGED batch sums (sum w (y' - y)^2 pairs a frame with its
  //     partner); cvf_ef16_finish then adds the rows as in generator mode (was: cvf_ef_stats, two launches, 15 us).  The
  //     launch is one round of units_x blocks doing two units each instead of two rounds of 2 units_x blocks.
  //   unpaired: the units of x, then the units of the lagged frames, one per block.
  // In both the lagged frames' tiles follow the tiles of x in every tiled output, and a pass stops after y and the hand-off.
  const bool paired = kTransfer && x_lag != nullptr && (launch >> 8) != 0;   // (uniform)
  const int nc = pp.n_coord, nal = pp.n_align, N = pp.n_rec;
  const int stride = x_tile_stride(nc);
  const Front16Lds Lo = front16_lds(nc, nal, k);   // (the transfer instance is launched with Lo.g + 16 floats per unit: no g images)
  float* lds = lds_all + usub * (kTransfer ? Lo.g + 16 : Lo.total);
  // units past the last one (the last workgroup of a launch whose unit count is not a multiple of upb) repeat the last unit: the
  // same values into the same places, and every wave meets the workgroup's barriers
  const int64_t n_ublock = (kTransfer && !paired) ? 2 * units_x : units_x;
  const int64_t ub = ublock < n_ublock ? ublock : n_ublock - 1;
  const float* const x_first = x;
  const float* const w_first = w;
  // one unit from its coordinates to y (transfer instance) / to the unit's row of batch sums (generator instances).  A lambda so that
  // a paired transfer block can run it twice as STRAIGHT-LINE code: written as a loop, the by-value descriptors stay in scalar
  // registers across the iterations, the scalar file overflows into vector lanes and the kernel spills (128 + 37 against 109)
  auto run_pass = [&](const int pass) __attribute__((always_inline)) {
  const int lane = lane_in;
  // ---- this net's weights (requested in every pass, behind the coordinates)
  const PackLayout L = pack_layout(H, NH, D);
  const URows pk = urows(packed + (int64_t)net * L.per_net, L.per_net, lane);   // this net's fragments (see URows)
  const int S = (D + 3) >> 2, CT = (D + 15) >> 4;
  float a0[SMAX][RT];
  float bias[NH][RT][4];
  const int q_ = lane >> 4;
  auto request_layer0 = [&]() {
#pragma unroll
    for (int s = 0; s < SMAX; ++s) {
      const int se = s < S ? s : S - 1;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) a0[s][rt] = pk.ld(L.f0() + (se * RT + rt) * 64);   // (k-steps past S are skipped below)
    }
    load_hid_const_u<H>(urows(theta + mlp.b_off[net][0], H, q_), bias[0]);
  };
  const bool lagged = x_lag != nullptr && (paired ? pass == 1 : ub >= units_x);   // (wave-uniform)
  const int64_t unit = (lagged && !paired) ? ub - units_x : ub;   // unit within its frame set
  const int64_t tile = (unit >> 2) + (lagged ? (units_x >> 2) : 0);
  const int sub = (int)(unit & 3);
  x = lagged ? x_lag : x_first;
  w = lagged ? w_lag : w_first;
  float* xt = lds;
  float* refL = lds + Lo.ref;
  float* aL = lds + Lo.a;
  float* auxL = lds + Lo.aux;
  float* wL = (kTransfer && pass == 1) ? lds + Lo.g : lds + Lo.w;   // (second pass: the partners' weights behind the layout)
  float* rsL = lds + Lo.rs;
  float* yL = lds + Lo.y + (kTransfer ? pass * (k * kU) : 0);   // (second pass: y' goes where the generator instance keeps E)
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
  // (wave 0 asks after its alignment: the solve's fp64 state and these 36 + 8 registers do not fit 128 together, and
  //  a spilled fragment is stored behind a wait for ALL outstanding loads)
  if (wave != 0) request_layer0();   // (every pass: kept across wave 0's solve of the second pass they would spill)
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
    if constexpr (kTransfer) {   // transfer-operator mode: no derivative through the alignment, the rotation alone
      kabsch_from_H<false>(Hm, ko);
#pragma unroll
      for (int i = 0; i < 6; ++i) ko.Kinv[i] = 0.0f;
    } else {
      kabsch_from_H<true>(Hm, ko);
    }
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
  if constexpr (kTransfer) {   // transfer-operator mode: y, h_1..h_NH and the feature tile are all the backward pass needs
#pragma unroll
    for (int l = 0; l < NH; ++l)
#pragma unroll
      for (int g = 0; g < NG; ++g) sv.st((l * NG + g) * 256, h[l].v[g >> 2][0][g & 3]);
    float* ft = feat_tiled + tile * (int64_t)D * CVF_TILE + kU * sub + f;
    const float* fi = featI + f * kImgP;
#pragma unroll 1
    for (int j = p + 4 * wave; j < D; j += 4 * nw) ft[j * CVF_TILE] = fi[j];
    return;
  } else {

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
    // (a plain counted loop: left to itself the compiler unrolls and vectorises this run-time trip count into ~200 vector
    //  instructions with three remainder paths and spills an address across them - each reload behind a wait for ALL stores)
#pragma unroll 1
    for (int j = p + 4 * wave; j < D; j += 4 * nw) ft[j * CVF_TILE] = fi[j];
  }
  CVF_STAMP(29);
  if (partial == nullptr) return;   // (uniform) large batches: the caller reduces y / E with cvf_ef_stats
  lds_barrier();                    // y and E of every net are in LDS

  // ---- wave 0: this unit's row of the batch sums [W | S1(k) | S2(i<=j) | E(k)] in fp64 on the matrix cores: with the frames
  // as the contraction index (four k-steps of four frames), D1 = [1, y_1..y_k] x [w, w y_1..w y_k] holds W (0,0), S1_j (0,j)
  // and S2_ij (i,j), D2 = [E_1..E_k] x [w, ..] holds E_i in column 0 - eight v_mfma_f64_16x16x4_f64 instead of the 13
  // statistic-by-statistic DPP scans (~570 vector instructions, 3-4 k cycles at the end of every block).  Products of two
  // floats are exact in fp64 and the hardware adds the k-steps in a fixed order: bitwise reproducible, as before.
  // cvf_ef_stats_finish adds the units' rows.
  if (wave == 0) {
    typedef double f64x4 __attribute__((ext_vector_type(4)));
    const int np = CVF_NPAIR(k);
    const int i = lane & 15, kq = lane >> 4;   // operand row / column, k-slot
    f64x4 d1 = {0.0, 0.0, 0.0, 0.0}, d2 = {0.0, 0.0, 0.0, 0.0};
    // (yL and eL are adjacent: rows 1..k of the A operand are y, the same index k rows further is E)
    const float* yrow = yL + (i >= 1 && i <= k ? i - 1 : 0) * kU;
    const float* erow = eL + (i < k ? i : 0) * kU;
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
      const int fr = 4 * s_ + kq;
      const double wb = (double)wL[fr];
      const double yv = (double)yrow[fr];
      const double a1 = i == 0 ? 1.0 : (i <= k ? yv : 0.0);
      const double b = i == 0 ? wb : (i <= k ? wb * yv : 0.0);
      const double a2 = i < k ? (double)erow[fr] : 0.0;
      d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b, d1, 0, 0, 0);
      d2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b, d2, 0, 0, 0);
    }
    // C/D of the f64 form: column = lane & 15, row = (lane >> 4) + 4 r
    const int cj = lane & 15;
    const int64_t G = units_x;   // rows of the launch = its units
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ri = (lane >> 4) + 4 * r;
      // D1: (0, 0) -> W; (0, j) -> S1_j; (i, j), 1 <= i <= j <= k -> S2 pair (i-1, j-1) in row-major i <= j order
      int t = -1;
      if (ri == 0 && cj <= k) t = cj;
      else if (ri >= 1 && ri <= cj && cj <= k) {
        const int a_ = ri - 1, b_ = cj - 1;
        t = 1 + k + a_ * k - (a_ * (a_ - 1)) / 2 + (b_ - a_);
      }
      if (t >= 0) partial[t * G + unit] = d1[r];
      if (cj == 0 && ri < k) partial[(1 + k + np + ri) * G + unit] = d2[r];
    }
  }
  CVF_STAMP(30);
  }   // (generator instance)
  };  // run_pass
  run_pass(0);
  if constexpr (kTransfer) {
    if (!paired) return;                         // (uniform)
    run_pass(1);                                 // the same waves, the lagged partners of these 16 frames
    if (partial == nullptr) return;              // (uniform)
    const int lane = lane_in;
    const int64_t unit = ub;
    lds_barrier();                               // y and y' of every net are in LDS
    // ---- wave 0: this unit's row of the time-lagged batch sums [W | S1 | S2(i<=j) | W' | S1' | S2'_ii | T] (cvf_ef_nstats, lag > 0)
    // in fp64 on the matrix cores, the 16 frames as the contraction index: D1 = [1, y] x [w, w y] holds W, S1_j, S2_ij;
    // D2 = [1, y'] x [w', w' y'] holds W', S1'_j and S2'_jj on its diagonal; D3 = [y' - y] x [w (y' - y)] holds T_i = sum w (y'_i - y_i)^2
    // on its diagonal (core.py:412-416, 428).  Products of two floats are exact in fp64, the k-steps add in a fixed order.
    if (wave == 0) {
      typedef double f64x4 __attribute__((ext_vector_type(4)));
      const int np = CVF_NPAIR(k);
      const int i = lane & 15, kq = lane >> 4;
      const float* y0 = lds + Lo.y;                  // y of the frames | y' of the partners (the two passes' yL)
      const float* y1 = y0 + k * kU;
      const float* w0 = lds + Lo.w;
      const float* w1 = lds + Lo.g;
      f64x4 d1 = {0.0, 0.0, 0.0, 0.0}, d2 = {0.0, 0.0, 0.0, 0.0}, d3 = {0.0, 0.0, 0.0, 0.0};
      const int yi = (i >= 1 && i <= k ? i - 1 : 0) * kU, ti = (i < k ? i : 0) * kU;
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
        const int fr = 4 * s_ + kq;
        const double wb = (double)w0[fr], wl_ = (double)w1[fr];
        const double yv = (double)y0[yi + fr], ylv = (double)y1[yi + fr];
        const double a1 = i == 0 ? 1.0 : (i <= k ? yv : 0.0), b1 = i == 0 ? wb : (i <= k ? wb * yv : 0.0);
        const double a2 = i == 0 ? 1.0 : (i <= k ? ylv : 0.0), b2 = i == 0 ? wl_ : (i <= k ? wl_ * ylv : 0.0);
        const double df = i < k ? (double)y1[ti + fr] - (double)y0[ti + fr] : 0.0;
        d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, d1, 0, 0, 0);
        d2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, d2, 0, 0, 0);
        d3 = __builtin_amdgcn_mfma_f64_16x16x4f64(df, wb * df, d3, 0, 0, 0);
      }
      const int cj = lane & 15, o = 1 + k + np;
      const int64_t G = units_x;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ri = (lane >> 4) + 4 * r;
        int t = -1;
        if (ri == 0 && cj <= k) t = cj;
        else if (ri >= 1 && ri <= cj && cj <= k) {
          const int a_ = ri - 1, b_ = cj - 1;
          t = 1 + k + a_ * k - (a_ * (a_ - 1)) / 2 + (b_ - a_);
        }
        if (t >= 0) partial[t * G + unit] = d1[r];
        if (ri == 0 && cj <= k) partial[(o + cj) * G + unit] = d2[r];                             // W', S1'_j
        if (ri >= 1 && ri == cj && cj <= k) partial[(o + 1 + k + (cj - 1)) * G + unit] = d2[r];   // S2'_jj
        if (ri == cj && cj < k) partial[(o + 1 + 2 * k + cj) * G + unit] = d3[r];                 // T_i
      }
    }
  }
}
}  // namespace


int cvf_ef_stats_finish_impl(const cvf_ef_cfg* cfg, int n_rows, int stat_major, const double* partial, double* stats,
                             double* loss_vec, double* coef, hipStream_t s);
int cvf_ef_stats_finish_ll(const cvf_ef_cfg* cfg, int n_rows, int stat_major, const double* partial, double* stats,
                           double* loss_vec, double* coef, hipStream_t s, const P2PLL* ll);

// Units per workgroup of a front launch (see the kernel).  ONE by default.  Several (CVF_EF16_UPB = 2..5, a developer switch, results
// bit for bit those of one: tools/upb_check.py) were built on the observation that the 1250 one-unit workgroups of a 20 000-frame
// batch reach their first stamp over 4.4 us - and measured SLOWER: 37.2 us per launch with five units per workgroup against 34.6
// (two or three: 47-49 us, a second round under the 16-waves-per-CU limit).  tools/dispatch_probe.hip settled why: starting 1250
// workgroups of 3 waves costs the same as 250 of 15 (2.5-2.8 us per empty launch, identical with work inside), so the spread of
// the first stamps is the cold instruction cache and kernel-argument fetch, which every wave pays wherever it sits; what the
// large workgroup adds is five units waiting at each other's barriers (two of the five alignment solves share a SIMD).
static int ef16_units_per_wg(int64_t units, int k, size_t lds_unit_bytes) {
  int cap = 16 / k;
  if (cap > 5) cap = 5;
  while (cap > 1 && (size_t)cap * lds_unit_bytes > 150 * 1024) --cap;
  if (cap < 1) cap = 1;
  int upb = 1;
  (void)units;
  if (const char* e = getenv("CVF_EF16_UPB")) {
    const int v = atoi(e);
    if (v >= 1 && v <= cap) upb = v;
  }
  return upb;
}

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
  return rows * cvf_ef_nstats(k, 1) + cvf_ef_stats_scratch_doubles(k, 1);   // (the time-lagged sums are the longer vector)
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

// data-parallel step: the units' rows -> this rank's sums -> the sum over ranks (peer-to-peer exchange inside the launch,
// cvf_p2p.hpp) -> loss tail and coefficients.  One launch where the step had three (finish, all-reduce, cvf_ef_loss).
extern "C" int cvf_ef16_finish_dp(const cvf_ef_cfg* cfg, int64_t B, const double* scratch, double* stats, double* loss_vec, double* coef,
                                  void* p2p_comm, void* stream) {
  CVF_REQUIRE(cfg && scratch && stats && loss_vec && coef && cvf_ef16_rows(B) > 0, "cvf_ef16_finish_dp: bad argument");
  const P2PLL* ll = cvf_p2p_ll(p2p_comm, 0);
  if (ll == nullptr) return -1;
  return cvf_ef_stats_finish_ll(cfg, (int)cvf_ef16_rows(B), 1, scratch, stats, loss_vec, coef, (hipStream_t)stream, ll);
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
  const size_t lds1 = (size_t)front16_lds(pp->n_coord, pp->n_align, k).total * sizeof(float);
  const int upb = ef16_units_per_wg(units, k, lds1);
  const size_t lds = lds1 * upb;
  (void)ns;
  ef16_dispatch(H, NH, [&](auto h_, auto nh_) {
    constexpr int kH = decltype(h_)::value, kNH = decltype(nh_)::value;
    auto go = [&](auto kernel) {
      if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(kernel, dim3((unsigned)((units + upb - 1) / upb)), dim3(64 * k * upb), lds, (hipStream_t)stream, *mlp, theta, packed, *pp,
                         x, B, a, w, feat_tiled, y_tiled, saved, q_tiled, e_tiled, rows ? scratch : nullptr, upb, (const float*)nullptr, units,
                         (const float*)nullptr);
    };
    const int nit = (pp->n_rec + 3) / 4;   // atoms per lane in the four-lanes-per-frame passes (1..6: d_r <= 72)
    const bool allal = pp->n_align == pp->n_rec;
#define EF16_GO(NIT_)                                                  \
    case NIT_:                                                          \
      if (allal) go(ef16_front_kernel<kH, kNH, NIT_, true>);            \
      else go(ef16_front_kernel<kH, kNH, NIT_, false>);                 \
      break;
    switch (nit) {
#ifndef CVF_DEV_SHAPES
      EF16_GO(1) EF16_GO(2) EF16_GO(3) EF16_GO(4) EF16_GO(5)
#endif
      default:
        if (allal) go(ef16_front_kernel<kH, kNH, 6, true>);
#ifndef CVF_DEV_SHAPES
        else go(ef16_front_kernel<kH, kNH, 6, false>);
#endif
    }
#undef EF16_GO
  });
  int rc = cvf_check_launch("ef16_front_kernel");
  if (rc || stats == nullptr) return rc;   // stats == NULL: the caller adds the units' rows itself (cvf_ef16_finish)
  if (rows) return cvf_ef_stats_finish_impl(cfg, (int)units, 1, scratch, stats, loss_vec, coef, (hipStream_t)stream);
  return cvf_ef_stats(cfg, B, w, y_tiled, e_tiled, nullptr, nullptr, scratch, stats, loss_vec, coef, stream);
}


// ------------------------------------------------------------------------------------------------------------------
// transfer-operator mode (lag_tau > 0; core.py:403,414,420-431,440): y on the frames and on their lagged partners, then the
// backward pass of both with the coefficients of the time-lagged loss.  Same kernels, same hand-off layout; the front
// kernel stops after y (no derivative passes), the backward kernel is compiled without the tangent chain.
// ------------------------------------------------------------------------------------------------------------------
static int ef16_front_transfer_impl(const cvf_mlp_desc* mlp, const float* theta, const float* packed, float* feat_tiled,
                                    const cvf_pp_desc* pp, const float* x, const float* x_lag, int64_t B, float* y_tiled, float* saved,
                                    const float* w, const float* w_lag, double* rows_out, void* stream, const char* what) {
  CVF_REQUIRE(cvf_ef16_supported(mlp, pp), "%s: shape not covered (cvf_ef16_supported() == 0)", what);
  CVF_REQUIRE(theta && packed && feat_tiled && x && x_lag && y_tiled && saved && B > 0, "%s: bad argument", what);
  int H, NH;
  ef16_shape(mlp, &H, &NH);
  const int k = mlp->n_nets;
  const int64_t T = cvf_ntiles(B), units = 4 * T;
  CVF_REQUIRE(2 * units < (int64_t)1 << 31, "%s: batch too large for one launch", what);
  // a unit and its lagged partner in one block, one after the other (CVF_EF16_UNPAIRED=1: one unit per block, the two sets one after the
  // other in the grid - the launch of rounds 2-3, kept as a developer switch)
  const bool paired = rows_out != nullptr || getenv("CVF_EF16_UNPAIRED") == nullptr;
  const size_t lds1 = ((size_t)front16_lds(pp->n_coord, pp->n_align, k).g + 16) * sizeof(float);   // (no g images in this instance; + the partners' weights)
  const int upb = paired ? ef16_units_per_wg(units, k, lds1) : 1;
  const size_t lds = lds1 * upb;
  ef16_dispatch(H, NH, [&](auto h_, auto nh_) {
    constexpr int kH = decltype(h_)::value, kNH = decltype(nh_)::value;
    auto kernel = ef16_front_kernel<kH, kNH, 0, true>;   // NIT = 0: the transfer-operator instance
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kernel, dim3((unsigned)(paired ? (units + upb - 1) / upb : 2 * units)), dim3(64 * k * upb), lds, (hipStream_t)stream, *mlp,
                       theta, packed, *pp, x, B, (const float*)nullptr, w, feat_tiled, y_tiled, saved, (float*)nullptr, (float*)nullptr,
                       rows_out, upb | (paired ? 256 : 0), x_lag, units, w_lag);
  });
  return cvf_check_launch("ef16_front_kernel");
}

extern "C" int cvf_ef16_front_transfer(const cvf_mlp_desc* mlp, const float* theta, const float* packed, float* feat_tiled,
                                       const cvf_pp_desc* pp, const float* x, const float* x_lag, int64_t B, float* y_tiled,
                                       float* saved, void* stream) {
  return ef16_front_transfer_impl(mlp, theta, packed, feat_tiled, pp, x, x_lag, B, y_tiled, saved, nullptr, nullptr, nullptr, stream,
                                  "cvf_ef16_front_transfer");
}

// ... and the units' rows of the time-lagged batch sums in the same launch (cvf_ef16_transfer_rows(B, k) > 0): follow with
// cvf_ef16_finish / cvf_ef16_finish_dp (cfg.lag_idx > 0) instead of cvf_ef_stats
extern "C" int64_t cvf_ef16_transfer_rows(int64_t B, int k) { return k >= 1 ? cvf_ef16_rows(B) : 0; }
extern "C" int cvf_ef16_front_transfer_rows(const cvf_mlp_desc* mlp, const float* theta, const float* packed, float* feat_tiled,
                                            const cvf_pp_desc* pp, const float* x, const float* x_lag, int64_t B, float* y_tiled,
                                            float* saved, const float* w, const float* w_lag, double* scratch, void* stream) {
  CVF_REQUIRE(w && w_lag && scratch && mlp && cvf_ef16_transfer_rows(B, mlp->n_nets) > 0, "cvf_ef16_front_transfer_rows: bad argument");
  return ef16_front_transfer_impl(mlp, theta, packed, feat_tiled, pp, x, x_lag, B, y_tiled, saved, w, w_lag, scratch, stream,
                                  "cvf_ef16_front_transfer_rows");
}
