// Shared by the derivative kernel (k1_align.hip) and the fused forward + derivative kernel (ef_mfma.hip):
// the centred-coordinate helper, the descriptor of the fused batch sums, and the three passes of
// q = J A J^T g, E = g^T J A J^T g for the fast layout (one lane = one frame).
#pragma once
#include "cvf_kabsch.hpp"

// x - c in fp32 with the centroid as a (hi, lo) pair of floats: two roundings of the result's own ulp, no
// fp64 conversions in the per-atom loops (v_cvt_f64_f32 / v_cvt_f32_f64 are quarter-rate).
struct Centre {
  float hi[3], lo[3];
};
__device__ __forceinline__ Centre centre_of(const double (&c)[3]) {
  Centre ce;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    ce.hi[i] = (float)c[i];
    ce.lo[i] = (float)(c[i] - (double)ce.hi[i]);
  }
  return ce;
}
__device__ __forceinline__ Centre centre_of(float c0, float c1, float c2) { return Centre{{c0, c1, c2}, {0.0f, 0.0f, 0.0f}}; }
__device__ __forceinline__ V3 centred(const float* my, int a, const Centre& c) {
  return V3{(my[3 * a] - c.hi[0]) - c.lo[0], (my[3 * a + 1] - c.hi[1]) - c.lo[1], (my[3 * a + 2] - c.hi[2]) - c.lo[2]};
}


// 3x3 products in packed fp32 (v_pk_fma_f32: two results per instruction; a third of the passes' vector instructions)
struct MatCols {   // y = M v:  (y0, y1) accumulate column pairs of rows 0 and 1, y2 the third row
  f2 c0, c1, c2;
  float r0, r1, r2;
};
__device__ __forceinline__ MatCols mat_cols(const float* M) {
  return MatCols{f2{M[0], M[3]}, f2{M[1], M[4]}, f2{M[2], M[5]}, M[6], M[7], M[8]};
}
__device__ __forceinline__ void mat_times_acc(const MatCols& m, V3 v, f2& yxy, float& yz) {
  yxy = fma2(splat2(v.x), m.c0, fma2(splat2(v.y), m.c1, fma2(splat2(v.z), m.c2, yxy)));
  yz = fmaf(v.x, m.r0, fmaf(v.y, m.r1, fmaf(v.z, m.r2, yz)));
}
struct MatRows {   // y = v^T M (row vector times matrix): (y0, y1) accumulate the row pairs (M[3i], M[3i+1]), y2 column 2
  f2 r0, r1, r2;
  float c0, c1, c2;
};
__device__ __forceinline__ MatRows mat_rows(const float* M) {
  return MatRows{f2{M[0], M[1]}, f2{M[3], M[4]}, f2{M[6], M[7]}, M[2], M[5], M[8]};
}
__device__ __forceinline__ void row_times_acc(const MatRows& m, V3 v, f2& yxy, float& yz) {
  yxy = fma2(splat2(v.x), m.r0, fma2(splat2(v.y), m.r1, fma2(splat2(v.z), m.r2, yxy)));
  yz = fmaf(v.x, m.c0, fmaf(v.y, m.c1, fmaf(v.z, m.c2, yz)));
}
// outer-product accumulation  A[i][0..1] += a_i (b0, b1),  A[i][2] += a_i b2
struct Outer3 {
  f2 xy[3];
  float z[3];
};
__device__ __forceinline__ void outer_acc(Outer3& A, V3 a, V3 b) {
  const f2 b01 = f2{b.x, b.y};
  A.xy[0] = fma2(splat2(a.x), b01, A.xy[0]); A.z[0] = fmaf(a.x, b.z, A.z[0]);
  A.xy[1] = fma2(splat2(a.y), b01, A.xy[1]); A.z[1] = fmaf(a.y, b.z, A.z[1]);
  A.xy[2] = fma2(splat2(a.z), b01, A.xy[2]); A.z[2] = fmaf(a.z, b.z, A.z[2]);
}
__device__ __forceinline__ void outer_to_array(const Outer3& A, float (&M)[9]) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    M[3 * i] = A.xy[i].x;
    M[3 * i + 1] = A.xy[i].y;
    M[3 * i + 2] = A.z[i];
  }
}

constexpr int kGChunk = 8;   // atoms per prefetch chunk of pass 1

// Fused first stage of K5: every block also reduces its tile's contribution to the batch sums of
// EigenFunctionTask.loss_func into one row of `partial` (fp64, fixed DPP order); cvf_ef_stats_finish (stats.hip)
// adds the rows in a fixed order - the separate pass over w, y and E and its launch are gone.
// (A last-block-done epilogue that also finished the sum and the loss tail in this launch was measured and dropped:
// 313 returning atomics on one counter serialise to ~7 us, more than the launch boundary it saves.)
struct MetricFuse {
  int on;
  int ns;
  const float* w;
  const float* y_tiled;
  double* partial;       // [T][ns]
};
constexpr int kFuseMaxTiles = 1024;   // above this the row sum of the finishing kernel would be too serial


// The three passes on one wave's 64 frames (lane = frame).  `my`: the lane's coordinates in LDS; refL / aL: reference
// and diagonal coefficients in LDS; Ul: the wave's [3N][64] image + lane.  G_IN_LDS: g already sits in the image
// (fused kernel); otherwise it is read from `gt` (tiled global, `cur` holding the first prefetched chunk).
template <bool G_IN_LDS>
__device__ __forceinline__ void metric_pure_passes(const cvf_pp_desc& pp, int lane, int net, int k, int64_t tile, int64_t B,
                                                   const float* my, const float* refL, const float* aL, float* Ul,
                                                   const float (&auxv)[CVF_AUX_ROWS], const float* __restrict__ gt,
                                                   float* __restrict__ qt, float* __restrict__ e_tiled, const MetricFuse& fuse,
                                                   float wv, const float (&yv)[CVF_MAX_NETS], float (&cur)[3 * kGChunk]) {
  const int N = pp.n_rec, nal = pp.n_align;
  float R[9], Kinv[6];
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = auxv[i];
  const Centre c = centre_of(auxv[9], auxv[10], auxv[11]);
#pragma unroll
  for (int i = 0; i < 6; ++i) Kinv[i] = auxv[12 + i];
  CVF_STAMP(10);
  // pass 1
  const MatCols Rc = mat_cols(R);
  f2 sump_xy = {0.0f, 0.0f};
  float sump_z = 0.0f;
  Outer3 Mo = {{{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}}, {0.0f, 0.0f, 0.0f}};
  if (G_IN_LDS) {
    // g was left in the image by the forward part of the same wave: plain LDS reads, nothing to prefetch or store
#pragma unroll 4
    for (int at = 0; at < N; ++at) {
      const V3 g = v3(Ul[(3 * at) * CVF_TILE], Ul[(3 * at + 1) * CVF_TILE], Ul[(3 * at + 2) * CVF_TILE]);
      mat_times_acc(Rc, g, sump_xy, sump_z);
      outer_acc(Mo, centred(my, at, c), g);
    }
  } else {
    float nxt[3 * kGChunk];
    const int nchunk = (N + kGChunk - 1) / kGChunk;
    for (int ch = 0; ch < nchunk; ++ch) {
  #pragma unroll
      for (int i = 0; i < kGChunk; ++i) {   // next chunk: clamped, unconditional
        const int an = kGChunk * (ch + 1) + i;
        const int at = an < N ? an : N - 1;
  #pragma unroll
        for (int d = 0; d < 3; ++d) nxt[3 * i + d] = gt[(3 * at + d) * CVF_TILE];
      }
  #pragma unroll
      for (int i = 0; i < kGChunk; ++i) {
        const int an = kGChunk * ch + i;
        const int at = an < N ? an : N - 1;
        const float m = an < N ? 1.0f : 0.0f;
        const V3 g = v3(cur[3 * i], cur[3 * i + 1], cur[3 * i + 2]);
        Ul[(3 * at) * CVF_TILE] = g.x;   // (a clamped atom rewrites atom N-1's own g)
        Ul[(3 * at + 1) * CVF_TILE] = g.y;
        Ul[(3 * at + 2) * CVF_TILE] = g.z;
        const V3 gm = m * g;
        mat_times_acc(Rc, gm, sump_xy, sump_z);
        outer_acc(Mo, centred(my, at, c), gm);
      }
  #pragma unroll
      for (int i = 0; i < 3 * kGChunk; ++i) cur[i] = nxt[i];
    }
  }
  CVF_STAMP(11);
  const V3 sump = v3(sump_xy.x, sump_xy.y, sump_z);
  float M[9];
  outer_to_array(Mo, M);
  float T[9], Z[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) T[3 * i + j] = R[i] * M[j] + R[3 + i] * M[3 + j] + R[6 + i] * M[6 + j];
  const V3 s = sym_times(Kinv, v3(T[7] - T[5], T[2] - T[6], T[3] - T[1]));
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    Z[3 * i + 0] = R[3 * i + 1] * s.z - R[3 * i + 2] * s.y;
    Z[3 * i + 1] = -R[3 * i + 0] * s.z + R[3 * i + 2] * s.x;
    Z[3 * i + 2] = R[3 * i + 0] * s.y - R[3 * i + 1] * s.x;
  }
  const float inv_nal = 1.0f / (float)nal;
  const V3 shift = inv_nal * sump;
  CVF_STAMP(12);
  // pass 2: align atoms first (they carry the rotation's and the centroid's derivative), then the rest
  const MatCols Zc = mat_cols(Z);
  f2 E2 = {0.0f, 0.0f};
  float Ez = 0.0f;
  f2 usum_xy = {0.0f, 0.0f}, rsum_xy = {0.0f, 0.0f};
  float usum_z = 0.0f, rsum_z = 0.0f;
  Outer3 dHo = {{{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}}, {0.0f, 0.0f, 0.0f}};
  const f2 nshift_xy = f2{-shift.x, -shift.y};
#pragma unroll 4
  for (int at = 0; at < nal; ++at) {
    const V3 g = v3(Ul[(3 * at) * CVF_TILE], Ul[(3 * at + 1) * CVF_TILE], Ul[(3 * at + 2) * CVF_TILE]);
    const V3 rf = v3(refL[3 * at], refL[3 * at + 1], refL[3 * at + 2]);
    f2 Gxy = nshift_xy;            // G = R g + (Z rf - shift)
    float Gz = -shift.z;
    mat_times_acc(Rc, g, Gxy, Gz);
    mat_times_acc(Zc, rf, Gxy, Gz);
    const f2 uxy = f2{aL[3 * at], aL[3 * at + 1]} * Gxy;
    const float uz = aL[3 * at + 2] * Gz;
    E2 = fma2(uxy, Gxy, E2);
    Ez = fmaf(uz, Gz, Ez);
    Ul[(3 * at) * CVF_TILE] = uxy.x;
    Ul[(3 * at + 1) * CVF_TILE] = uxy.y;
    Ul[(3 * at + 2) * CVF_TILE] = uz;
    usum_xy += uxy; usum_z += uz;
    rsum_xy += f2{rf.x, rf.y}; rsum_z += rf.z;
    outer_acc(dHo, v3(uxy.x, uxy.y, uz), rf);
  }
#pragma unroll 4
  for (int at = nal; at < N; ++at) {
    const V3 g = v3(Ul[(3 * at) * CVF_TILE], Ul[(3 * at + 1) * CVF_TILE], Ul[(3 * at + 2) * CVF_TILE]);
    f2 Gxy = {0.0f, 0.0f};
    float Gz = 0.0f;
    mat_times_acc(Rc, g, Gxy, Gz);
    const f2 uxy = f2{aL[3 * at], aL[3 * at + 1]} * Gxy;
    const float uz = aL[3 * at + 2] * Gz;
    E2 = fma2(uxy, Gxy, E2);
    Ez = fmaf(uz, Gz, Ez);
    Ul[(3 * at) * CVF_TILE] = uxy.x;
    Ul[(3 * at + 1) * CVF_TILE] = uxy.y;
    Ul[(3 * at + 2) * CVF_TILE] = uz;
  }
  const float E = (E2.x + E2.y) + Ez;
  const V3 usum = v3(usum_xy.x, usum_xy.y, usum_z), rsum = v3(rsum_xy.x, rsum_xy.y, rsum_z);
  float dH[9];
  outer_to_array(dHo, dH);
  CVF_STAMP(13);
  e_tiled[(tile * k + net) * CVF_TILE + lane] = E;
  if (fuse.on) {
    // this wave's slots of the tile's row: [W | S1(k) | S2(i<=j) | E(k)]  (include/cvf.h)
    double* row = fuse.partial + tile * (int64_t)fuse.ns;
    const double wb = (double)wv, yn = (double)yv[net];
    auto put = [&](int slot, double v) {
      const double sum = wave_sum(v);
      if (lane == 0) row[slot] = sum;
    };
    if (net == 0) put(0, wb);
    put(1 + net, wb * yn);
    const int s2o = 1 + k + net * k - (net * (net - 1)) / 2 - net;   // + j : slot of S2[net][j], j >= net
#pragma unroll
    for (int j = 0; j < CVF_MAX_NETS; ++j)
      if (j >= net && j < k) put(s2o + j, wb * yn * (double)yv[j]);
    put(1 + k + CVF_NPAIR(k) + net, wb * (double)E);
  }
  const V3 ubar = inv_nal * usum;
  // dH = sum_b (u_b - ubar) (x) ref_b = dH' - ubar (x) sum_b ref_b
  dH[0] -= ubar.x * rsum.x; dH[1] -= ubar.x * rsum.y; dH[2] -= ubar.x * rsum.z;
  dH[3] -= ubar.y * rsum.x; dH[4] -= ubar.y * rsum.y; dH[5] -= ubar.y * rsum.z;
  dH[6] -= ubar.z * rsum.x; dH[7] -= ubar.z * rsum.y; dH[8] -= ubar.z * rsum.z;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) T[3 * i + j] = R[i] * dH[j] + R[3 + i] * dH[3 + j] + R[6 + i] * dH[6 + j];
  const V3 w = sym_times(Kinv, v3(T[7] - T[5], T[2] - T[6], T[3] - T[1]));
  float dR[9];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    dR[3 * i + 0] = R[3 * i + 1] * w.z - R[3 * i + 2] * w.y;
    dR[3 * i + 1] = -R[3 * i + 0] * w.z + R[3 * i + 2] * w.x;
    dR[3 * i + 2] = R[3 * i + 0] * w.y - R[3 * i + 1] * w.x;
  }
  CVF_STAMP(14);
  // pass 3
  const MatRows Rr = mat_rows(R), dRr = mat_rows(dR);
#pragma unroll 4
  for (int at = 0; at < N; ++at) {
    const V3 u = v3(Ul[(3 * at) * CVF_TILE], Ul[(3 * at + 1) * CVF_TILE], Ul[(3 * at + 2) * CVF_TILE]);
    f2 qxy = {0.0f, 0.0f};
    float qz = 0.0f;
    row_times_acc(Rr, u - ubar, qxy, qz);
    row_times_acc(dRr, centred(my, at, c), qxy, qz);
    qt[(3 * at) * CVF_TILE] = qxy.x;
    qt[(3 * at + 1) * CVF_TILE] = qxy.y;
    qt[(3 * at + 2) * CVF_TILE] = qz;
  }
  CVF_STAMP(15);
}

