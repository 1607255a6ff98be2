// The backward kernel of the 16-frames-per-wave step (see ef16_front.hip for the decomposition and the hand-off layout).
#include "ef16_common.hpp"

namespace {

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

  {   // (16-byte LDS writes: kRows * kPitch is a multiple of 4)
    static_assert((kRows * kPitch) % 4 == 0, "IMG is zeroed in 16-byte pieces");
    float4* img4 = reinterpret_cast<float4*>(IMG);
    for (int i = tid; i < kRows * kPitch / 4; i += NT) img4[i] = float4{0.0f, 0.0f, 0.0f, 0.0f};
  }
  if (MULTI)
    for (int i = tid; i < gspan; i += NT) GI[i] = 0.0f;
  // (LDS only: __syncthreads() would also wait for the coefficient / last-layer loads requested above - a whole global round trip
  //  at the head of every block before the tile's own requests go out; stamped prologue 4.4 k cycles)
  lds_barrier();
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

}  // namespace

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


extern "C" int cvf_ef16_backward_transfer(const cvf_ef_cfg* cfg, const cvf_mlp_desc* mlp, const float* theta, const float* packed,
                                          int64_t B, const float* w, const float* w_lag, const float* feat_tiled,
                                          const float* y_tiled, const double* coef, float* slab, int32_t* step_count,
                                          const float* saved, void* stream) {
  CVF_REQUIRE(cfg && cfg->lag_idx > 0, "cvf_ef16_backward_transfer: transfer-operator mode (cfg.lag_idx > 0)");
  return ef16_backward_impl(cfg, mlp, theta, packed, B, w, w_lag, feat_tiled, y_tiled, nullptr, coef, slab, step_count, saved, stream);
}