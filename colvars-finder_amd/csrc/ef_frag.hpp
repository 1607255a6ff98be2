// Register-level building blocks of the eigenfunction-net kernels on the matrix cores (shared by ef_mfma.hip and
// ef16_front.hip, ef16_back.hip): the "acc layout" of hidden vectors, weight-fragment loads, layer products, tanh, LDS operand images.
#pragma once
#include "cvf_common.hpp"
#include "cvf_pack.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));


constexpr int kPitch = 68;   // multiple of 4: every image row is 16-byte aligned, MFMA operands are read as ds_read_b128

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <int H>
struct Hid {
  static constexpr int NG = (H + 3) / 4;   // k-groups of 4 features
  static constexpr int RT = (NG + 3) / 4;  // 16-row tiles
};

template <int H, int FT>
struct Vec {
  f32x4 v[Hid<H>::RT][FT];
};

// Activations handed from the forward kernel to the backward kernel (cvf_ef_saved_floats): per (tile, net) the
// vectors h_1..h_NH in the register layout both kernels use - group g
// (features 4g..4g+3 over q), frame-group pair w (the backward kernel's wave), lane, two frame groups:
//   [vector][g < NG][w < 2][lane < 64][2]   ->  every (vector, g, w) is one coalesced 512-byte row.
template <int H>
__host__ __device__ constexpr int saved_per_vec() { return Hid<H>::NG * 2 * 64 * 2; }
template <int H>
__device__ __forceinline__ void save_vec(float* __restrict__ base, const Vec<H, 4>& X, int lane) {
#pragma unroll
  for (int g = 0; g < Hid<H>::NG; ++g)
#pragma unroll
    for (int w = 0; w < 2; ++w)
      reinterpret_cast<float2*>(base + (g * 2 + w) * 128)[lane] = float2{X.v[g >> 2][2 * w][g & 3], X.v[g >> 2][2 * w + 1][g & 3]};
}
// one half (frame groups 2 w, 2 w + 1) of the same layout, written by a forward block that owns half a tile
template <int H>
__device__ __forceinline__ void save_vec_half(float* __restrict__ base, const Vec<H, 2>& X, int w, int lane) {
#pragma unroll
  for (int g = 0; g < Hid<H>::NG; ++g)
    reinterpret_cast<float2*>(base + (g * 2 + w) * 128)[lane] = float2{X.v[g >> 2][0][g & 3], X.v[g >> 2][1][g & 3]};
}
template <int H>
__device__ __forceinline__ void load_vec(const float* __restrict__ base, Vec<H, 2>& X, int w, int lane) {
#pragma unroll
  for (int rt = 0; rt < Hid<H>::RT; ++rt)
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) X.v[rt][ft] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int g = 0; g < Hid<H>::NG; ++g) {
    const float2 t = reinterpret_cast<const float2*>(base + (g * 2 + w) * 128)[lane];
    X.v[g >> 2][0][g & 3] = t.x;
    X.v[g >> 2][1][g & 3] = t.y;
  }
}

// X <- bias (hidden order)
template <int H, int FT>
__device__ __forceinline__ void init_bias(Vec<H, FT>& X, const float* __restrict__ b, int q) {
#pragma unroll
  for (int rt = 0; rt < Hid<H>::RT; ++rt) {
    f32x4 bv;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = hid_feature(rt, r, q);
      bv[r] = 0.0f;
      if (b != nullptr) {
        const float x = b[f < H ? f : H - 1];
        bv[r] = f < H ? x : 0.0f;
      }
    }
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) X.v[rt][ft] = bv;
  }
}

// FT consecutive floats (the lane's frames 4*col + ft0 .. + FT-1 of one feature row)
template <int FT>
__device__ __forceinline__ void load_frames(const float* __restrict__ p, float (&b)[FT]) {
  if constexpr (FT == 4) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    b[0] = v.x; b[1] = v.y; b[2] = v.z; b[3] = v.w;
  } else if constexpr (FT == 2) {
    const float2 v = *reinterpret_cast<const float2*>(p);
    b[0] = v.x; b[1] = v.y;
  } else {
    b[0] = p[0];
  }
}
template <int FT>
__device__ __forceinline__ void store_frames(float* __restrict__ p, const float (&b)[FT]) {
  if constexpr (FT == 4) *reinterpret_cast<float4*>(p) = make_float4(b[0], b[1], b[2], b[3]);
  else if constexpr (FT == 2) *reinterpret_cast<float2*>(p) = make_float2(b[0], b[1]);
  else p[0] = b[0];
}

// ---- explicit prefetch.  At small batch sizes every wave of a launch starts together and each dependent
// round trip to L2 / Infinity Cache costs 500-900 cycles with nothing else resident to cover it, so the
// kernels issue the loads of a later phase (next chunk of k-steps, next layer's fragments) BEFORE the MFMAs
// of the current one and keep them in registers.

// A fragments of one H x H layer: NG k-steps x RT row tiles
template <int H>
struct HFrag {
  float a[Hid<H>::NG][Hid<H>::RT];
};
template <int H>
__device__ __forceinline__ void load_hfrag(HFrag<H>& f, const float* __restrict__ pk, int lane) {
#pragma unroll
  for (int s = 0; s < Hid<H>::NG; ++s)
#pragma unroll
    for (int rt = 0; rt < Hid<H>::RT; ++rt) f.a[s][rt] = pk[(unsigned)((s * Hid<H>::RT + rt) * 64) + (unsigned)lane];   // (unsigned: uniform base + 32-bit lane offset)
}
// Y += op(W) X with the fragments already in registers
template <int H, int FT>
__device__ __forceinline__ void hidden_mul(Vec<H, FT>& Y, const HFrag<H>& f, const Vec<H, FT>& X) {
#pragma unroll
  for (int s = 0; s < Hid<H>::NG; ++s)
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int rt = 0; rt < Hid<H>::RT; ++rt) Y.v[rt][ft] = mfma4(f.a[s][rt], X.v[s >> 2][ft][s & 3], Y.v[rt][ft]);
}
// hidden -> hidden with a just-in-time fragment load (kept for call sites with nothing to overlap)
template <int H, int FT>
__device__ __forceinline__ void hidden_apply(Vec<H, FT>& Y, const float* __restrict__ pk, const Vec<H, FT>& X, int lane) {
  HFrag<H> f;
  load_hfrag<H>(f, pk, lane);
  hidden_mul<H, FT>(Y, f, X);
}

// first layer: X += W0 [H x D] * in.  `in_lane` = tiled global array (rows = features, 64 frames per row)
// + the lane's frame offset 4*col + ft0; pk0 = packed fragments F0 of this net.  k-steps are processed in
// chunks of CH with the next chunk's operands (weights and activations) loading while the current one runs.
template <int H, int FT, int CH>
struct L0Chunk {
  float a[CH][Hid<H>::RT];
  float b[CH][FT];
};
template <int H, int FT, int CH>
__device__ __forceinline__ void load_l0chunk(L0Chunk<H, FT, CH>& c, const float* __restrict__ pk0, int D, int S,
                                             const float* __restrict__ in_lane, int s0, int lane) {
  constexpr int RT = Hid<H>::RT;
  const int q = lane >> 4;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    // no branches around the loads (steps past S read step S-1 and are zeroed by a select): with straight-line
    // code the compiler counts outstanding loads exactly and waits only for the chunk it is about to use
    const int s = s0 + i;
    const bool live = s < S;
    const int se = live ? s : S - 1;
    const int kf = 4 * se + q;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#ifdef CVF_EXP_NOW
      const float v = 0.01f * (float)(se + rt);
#else
      const float v = pk0[(se * RT + rt) * 64 + lane];
#endif
      c.a[i][rt] = live ? v : 0.0f;
    }
#ifdef CVF_EXP_NOFEAT
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) c.b[i][ft] = 0.5f + 0.01f * (float)kf;
#else
    load_frames<FT>(in_lane + (int64_t)(kf < D ? kf : 0) * CVF_TILE, c.b[i]);  // rows past D meet zero weights
#endif
  }
}
template <int H, int FT, int CH>
__device__ __forceinline__ void mul_l0chunk(Vec<H, FT>& X, const L0Chunk<H, FT, CH>& c) {
#pragma unroll
  for (int i = 0; i < CH; ++i)
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int rt = 0; rt < Hid<H>::RT; ++rt) X.v[rt][ft] = mfma4(c.a[i][rt], c.b[i][ft], X.v[rt][ft]);
}
template <int H, int FT, int CH = 6>
__device__ __forceinline__ void layer0_apply(Vec<H, FT>& X, const float* __restrict__ pk0, int D,
                                             const float* __restrict__ in_lane, int lane) {
  const int S = (D + 3) >> 2;
  L0Chunk<H, FT, CH> c0, c1;
  load_l0chunk<H, FT, CH>(c0, pk0, D, S, in_lane, 0, lane);
  for (int s0 = 0; s0 < S; s0 += 2 * CH) {
    load_l0chunk<H, FT, CH>(c1, pk0, D, S, in_lane, s0 + CH, lane);
    mul_l0chunk<H, FT, CH>(X, c0);
    load_l0chunk<H, FT, CH>(c0, pk0, D, S, in_lane, s0 + 2 * CH, lane);
    mul_l0chunk<H, FT, CH>(X, c1);
  }
}

// Wide first layers (hundreds of features; the large-molecule shapes): NB chunks in flight instead of two.  With two, every
// chunk of CH k-steps waited a full memory round trip (its operands were requested one chunk - ~100 cycles of MFMAs - ahead):
// 16 chunks x 2.8 k cycles = 45 k of the forward kernel's 79 k at d0 = 384.  The ring is straight-line code per round, so the
// compiler still counts outstanding loads exactly - as long as they fit the counter: a wave has at most 63 vector loads in
// flight (vmcnt), i.e. NB x CH x (RT + 1) <= 63 (with eight chunks of six the waits degenerated and the product took 76 k).
template <int H, int FT, int CH, int NB>
__device__ __forceinline__ void layer0_apply_deep(Vec<H, FT>& X, const float* __restrict__ pk0, int D,
                                                  const float* __restrict__ in_lane, int lane) {
  const int S = (D + 3) >> 2;
  L0Chunk<H, FT, CH> c[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) load_l0chunk<H, FT, CH>(c[b], pk0, D, S, in_lane, b * CH, lane);
  for (int s0 = 0; s0 < S; s0 += NB * CH) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      mul_l0chunk<H, FT, CH>(X, c[b]);
      load_l0chunk<H, FT, CH>(c[b], pk0, D, S, in_lane, s0 + (NB + b) * CH, lane);   // (past S: clamped loads, zeroed operands)
    }
  }
}
constexpr int kWideD = 128;   // first layers wider than this take the deep ring (and the hand-offs of the wide backward, ef_mfma.hip)

// the same with the first chunk's operands already requested by the caller (ahead of other work)
template <int H, int FT, int CH>
__device__ __forceinline__ void layer0_apply_from(Vec<H, FT>& X, const float* __restrict__ pk0, int D,
                                                  const float* __restrict__ in_lane, int lane, L0Chunk<H, FT, CH>& c0) {
  const int S = (D + 3) >> 2;
  L0Chunk<H, FT, CH> c1;
  for (int s0 = 0; s0 < S; s0 += 2 * CH) {
    load_l0chunk<H, FT, CH>(c1, pk0, D, S, in_lane, s0 + CH, lane);
    mul_l0chunk<H, FT, CH>(X, c0);
    load_l0chunk<H, FT, CH>(c0, pk0, D, S, in_lane, s0 + 2 * CH, lane);
    mul_l0chunk<H, FT, CH>(X, c1);
  }
}

// per-lane bias / last-layer weights in hidden order
template <int H>
struct HConst {
  float c[Hid<H>::RT][4];
};
template <int H>
__device__ __forceinline__ void load_hconst(HConst<H>& o, const float* __restrict__ v, int q) {
#pragma unroll
  for (int rt = 0; rt < Hid<H>::RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = hid_feature(rt, r, q);
      const float x = v[f < H ? f : H - 1];   // unconditional load + select: no branch, no early wait
      o.c[rt][r] = f < H ? x : 0.0f;
    }
}
template <int H, int FT>
__device__ __forceinline__ void set_const(Vec<H, FT>& X, const HConst<H>& b) {
#pragma unroll
  for (int rt = 0; rt < Hid<H>::RT; ++rt)
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int r = 0; r < 4; ++r) X.v[rt][ft][r] = b.c[rt][r];
}

// The hidden layers' activation.  `act` (cvf_mlp_desc.act[0], uniform over the launch) defaults to Tanh - the 16-frame kernels,
// which are instantiated for Tanh only, call without it and the other branch folds away; the 64-frame kernels pass the net's
// code, so that EigenFunctionTask takes the other activations of include/cvf.h on them (one wave-uniform branch per vector).
// (the other activations through real calls: inlined into every element of every vector of every kernel instance they
//  took the compilation of ef_mfma.hip from 2 to 13 minutes; the Tanh path stays inline and pays nothing for them)
__device__ __attribute__((noinline)) float ef_act_slow(int act, float z) { return cvf_act(act, z); }
__device__ __attribute__((noinline)) float ef_act_d1_slow(int act, float h) { return cvf_act_d1(act, h); }
__device__ __attribute__((noinline)) float ef_act_d2_slow(int act, float h) { return cvf_act_d2(act, h); }

template <int H, int FT>
__device__ __forceinline__ void tanh_inplace(Vec<H, FT>& X, int act = CVF_ACT_TANH) {
  if (act == CVF_ACT_TANH) {
#pragma unroll
    for (int rt = 0; rt < Hid<H>::RT; ++rt)
#pragma unroll
      for (int ft = 0; ft < FT; ++ft)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * rt + r < Hid<H>::NG) X.v[rt][ft][r] = cvf_tanh(X.v[rt][ft][r]);  // padding groups stay 0
  } else {
#pragma unroll
    for (int rt = 0; rt < Hid<H>::RT; ++rt)
#pragma unroll
      for (int ft = 0; ft < FT; ++ft)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * rt + r < Hid<H>::NG) X.v[rt][ft][r] = ef_act_slow(act, X.v[rt][ft][r]);
  }
}
// f'(z) of one element through its output h
__device__ __forceinline__ float act_d1(int act, float h) {
  if (act == CVF_ACT_TANH) return 1.0f - h * h;
  return ef_act_d1_slow(act, h);
}

// per-lane copy of a length-H vector in hidden order: c[rt][r] = v[hid_feature(rt,r,q)]
template <int H>
__device__ __forceinline__ void load_hid_const(const float* __restrict__ v, int q, float (&c)[Hid<H>::RT][4]) {
#pragma unroll
  for (int rt = 0; rt < Hid<H>::RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = hid_feature(rt, r, q);
      const float x = v[f < H ? f : H - 1];
      c[rt][r] = f < H ? x : 0.0f;
    }
}

// forward chain of one net: h[l] = tanh(W_l h_{l-1} + b_l).  All biases and the hidden layers' fragments are
// requested before the first layer's MFMAs, so their latency is covered by that layer.
// LEAN (register-starved callers): no cross-layer prefetch, smaller first-layer chunks.
template <int H, int NH, int FT, bool LEAN = false>
__device__ __forceinline__ void chain_forward(const cvf_mlp_desc& mlp, const float* __restrict__ theta,
                                              const float* __restrict__ pk, const PackLayout& L, int net,
                                              const float* __restrict__ in_lane, int lane, Vec<H, FT> (&h)[NH]) {
  const int q = lane >> 4;
  if constexpr (LEAN) {
    init_bias<H, FT>(h[0], theta + mlp.b_off[net][0], q);
    layer0_apply<H, FT, 3>(h[0], pk + L.f0(), mlp.dims[0], in_lane, lane);
    tanh_inplace<H, FT>(h[0], mlp.act[0]);
#pragma unroll
    for (int l = 1; l < NH; ++l) {
      init_bias<H, FT>(h[l], theta + mlp.b_off[net][l], q);
      hidden_apply<H, FT>(h[l], pk + L.fh(l), h[l - 1], lane);
      tanh_inplace<H, FT>(h[l], mlp.act[0]);
    }
    return;
  }
  HConst<H> bias[NH];
#pragma unroll
  for (int l = 0; l < NH; ++l) load_hconst<H>(bias[l], theta + mlp.b_off[net][l], q);
  HFrag<H> hf[NH > 1 ? NH - 1 : 1];
#pragma unroll
  for (int l = 1; l < NH; ++l) load_hfrag<H>(hf[l - 1], pk + L.fh(l), lane);
  CVF_STAMP(1);
  set_const<H, FT>(h[0], bias[0]);
  if (mlp.dims[0] > kWideD) layer0_apply_deep<H, FT, 6, 3>(h[0], pk + L.f0(), mlp.dims[0], in_lane, lane);
  else layer0_apply<H, FT>(h[0], pk + L.f0(), mlp.dims[0], in_lane, lane);
  CVF_STAMP(2);
  tanh_inplace<H, FT>(h[0], mlp.act[0]);
  CVF_STAMP(3);
#pragma unroll
  for (int l = 1; l < NH; ++l) {
    set_const<H, FT>(h[l], bias[l]);
    hidden_mul<H, FT>(h[l], hf[l - 1], h[l - 1]);
    tanh_inplace<H, FT>(h[l], mlp.act[0]);
  }
}

// sum over the 4 lane groups q (same col): after this every lane holds the full sum
__device__ __forceinline__ float sum_over_q(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// ------------------------------------------------------------------------------------------------------------------
// Rows of 64 floats (weight fragments, hand-off vectors) addressed as  uniform base + uniform row offset + lane:  a buffer
// resource in four SGPRs, the lane's byte offset in ONE VGPR, the row offset in an SGPR / the instruction's immediate
// (buffer_load_dword v, v_lane4, s[rsrc], s_off offen).  With plain pointer arithmetic the compiler folds the lane into
// the base first and then keeps one 64-bit VGPR address per row alive (36 registers for the first layer's fragments of
// one net, spilled when they are needed twice).  Reads past `n_floats` return 0 - which also masks a vector's padding.
// ------------------------------------------------------------------------------------------------------------------
struct URows {
  __amdgpu_buffer_rsrc_t r;
  int lane4;
  __device__ __forceinline__ float ld(int float_off) const {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane4, float_off * 4, 0));
  }
  __device__ __forceinline__ void st(int float_off, float v) const {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, lane4, float_off * 4, 0);
  }
};
__device__ __forceinline__ URows urows(const void* base, int n_floats, int lane) {
  return URows{__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, n_floats * 4, 0x00020000), lane * 4};
}
template <int H>
__device__ __forceinline__ void load_hfrag_u(HFrag<H>& f, const URows& u, int off) {
#pragma unroll
  for (int s = 0; s < Hid<H>::NG; ++s)
#pragma unroll
    for (int rt = 0; rt < Hid<H>::RT; ++rt) f.a[s][rt] = u.ld(off + (s * Hid<H>::RT + rt) * 64);
}
// per-lane copy of a length-H vector in hidden order (c[rt][r] = v[4 (4 rt + r) + q], 0 past H): `u` = urows(v, H, q),
// i.e. the lane offset is q and the feature's group a constant of the instruction
template <int H>
__device__ __forceinline__ void load_hid_const_u(const URows& u, float (&c)[Hid<H>::RT][4]) {
#pragma unroll
  for (int rt = 0; rt < Hid<H>::RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) c[rt][r] = u.ld(4 * (4 * rt + r));
}

// One 16x16 tile of a weight gradient over the block's 64 frames:  A (rows 16*rt..) x B (rows 16*ct..),
// operand images [feature][frame] with pitch kPitch.  The order in which the frames are summed is free as long as
// A and B agree: k-slot kq of k-step (j, c) is frame 16 j + 4 kq + c, so a lane's sixteen values of one operand
// are four 16-byte LDS reads (the natural order 4 s + kq needs sixteen 4-byte reads, four-way bank conflicted).
__device__ __forceinline__ f32x4 outer_half(const float* __restrict__ A, const float* __restrict__ B, int rt, int ct,
                                            int lane, f32x4 acc) {
  const int row = lane & 15, kq = lane >> 4;
  const float4* a = reinterpret_cast<const float4*>(A + (16 * rt + row) * kPitch + 4 * kq);
  const float4* b = reinterpret_cast<const float4*>(B + (16 * ct + row) * kPitch + 4 * kq);
  float4 av[4], bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    av[j] = a[4 * j];
    bv[j] = b[4 * j];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    acc = mfma4(av[j].x, bv[j].x, acc);
    acc = mfma4(av[j].y, bv[j].y, acc);
    acc = mfma4(av[j].z, bv[j].z, acc);
    acc = mfma4(av[j].w, bv[j].w, acc);
  }
  return acc;
}
__device__ __forceinline__ f32x4 outer_tile(const float* __restrict__ A1, const float* __restrict__ B1,
                                            const float* __restrict__ A2, const float* __restrict__ B2, int rt, int ct,
                                            bool two, int lane) {
  f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
  acc = outer_half(A1, B1, rt, ct, lane, acc);
  if (two) acc = outer_half(A2, B2, rt, ct, lane, acc);
  return acc;
}

// write a hidden vector (acc layout, this wave's FT frames per lane) into a [feature][frame] LDS image,
// optionally scaled per frame; the lane's frames fo..fo+FT-1 are contiguous
template <int H, int FT, bool SCALE>
__device__ __forceinline__ void store_image(float* S, const Vec<H, FT>& X, const float (&sc)[FT], int lane, int fo) {
  const int q = lane >> 4;
#pragma unroll
  for (int rt = 0; rt < Hid<H>::RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = hid_feature(rt, r, q);
      if (4 * rt + r < Hid<H>::NG && f < H) {
        float v[FT];
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) v[ft] = SCALE ? sc[ft] * X.v[rt][ft][r] : X.v[rt][ft][r];
        float* dst = S + f * kPitch + fo;
        if constexpr (FT == 4) {
          reinterpret_cast<float2*>(dst)[0] = make_float2(v[0], v[1]);   // rows are 8-byte aligned (pitch 66)
          reinterpret_cast<float2*>(dst)[1] = make_float2(v[2], v[3]);
        } else if constexpr (FT == 2) {
          *reinterpret_cast<float2*>(dst) = make_float2(v[0], v[1]);
        } else {
          dst[0] = v[0];
        }
      }
    }
}

// tdot = f'(z) .* t   (Tanh: (1 - h^2) .* t)
template <int H, int FT>
__device__ __forceinline__ void tangent_of(Vec<H, FT>& td, const Vec<H, FT>& h, const Vec<H, FT>& t, int act = CVF_ACT_TANH) {
  if (act == CVF_ACT_TANH) {
#pragma unroll
    for (int rt = 0; rt < Hid<H>::RT; ++rt)
#pragma unroll
      for (int ft = 0; ft < FT; ++ft)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float hv = h.v[rt][ft][r];
          td.v[rt][ft][r] = (1.0f - hv * hv) * t.v[rt][ft][r];
        }
  } else {
#pragma unroll
    for (int rt = 0; rt < Hid<H>::RT; ++rt)
#pragma unroll
      for (int ft = 0; ft < FT; ++ft)
#pragma unroll
        for (int r = 0; r < 4; ++r) td.v[rt][ft][r] = ef_act_d1_slow(act, h.v[rt][ft][r]) * t.v[rt][ft][r];
  }
}

}  // namespace
