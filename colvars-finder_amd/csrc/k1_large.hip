// K1 for large molecules (thousands of atoms).  A frame of N atoms is 12 N bytes (60 KB at N = 5000): too large
// for the lane-per-frame LDS tile, and HBM-bound by construction - every coordinate is needed exactly once, for
// the centroid and the 3x3 covariance.  One WAVE per frame, sixteen waves (= sixteen consecutive frames, a
// quarter of a 64-frame tile) per workgroup:
//   stream    12-byte loads fully coalesced across the wave, 8 atoms per lane in flight, fp64 accumulation of
//             sum x (3), sum x (x) ref (9), sum ref (3); one shuffle reduction; no LDS, no barrier.
//             (H = sum x (x) ref - c (x) sum ref: no second pass for the centred coordinates.)
//   solve     the workgroup's 3x3 problems are solved together, one per lane of ONE wave (cvf_kabsch.hpp).
//   features  the wave's lanes evaluate the feature records of ITS frame right away: the few hundred atoms they
//             touch were streamed microseconds ago and come from L2 / Infinity Cache, not HBM (a separate
//             feature launch re-fetched more sectors from HBM than the frame itself holds).
//   flush     features of the 16 frames are staged in LDS and leave as full 64-byte segments of the tiled layout.
#include "cvf_kabsch.hpp"
#include <stdio.h>
#include <stdlib.h>

namespace {

constexpr int kGroup = 8;  // waves = frames per workgroup
// Workgroups go round the 8 XCDs by blockIdx.  The eight workgroups of a 64-frame tile each write 32 bytes of every row of the tiled
// outputs ([tile][d_r][64] features, [tile][18][64] rotation rows): on eight different XCDs these stay eight partial lines in eight
// L2s, on ONE XCD they meet in its L2 and leave as whole lines.  The map keeps consecutive frame groups on one XCD.
__device__ __forceinline__ int64_t k1_group_of_block(bool same_xcd) {
  if (!same_xcd) return blockIdx.x;
  const int nb = gridDim.x, q8 = nb >> 3, r8 = nb & 7, xcd = blockIdx.x & 7, ix = blockIdx.x >> 3;
  return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + ix;
}

struct Rec {
  int type, a0, a1, a2, a3, out;
};
__device__ __forceinline__ V3 gatom(const float* __restrict__ xf, int a) { return V3{xf[3 * a], xf[3 * a + 1], xf[3 * a + 2]}; }

template <bool CONTIG>
__global__ __launch_bounds__(64 * kGroup) void k1_large_gather_kernel(cvf_pp_desc pp, const float* __restrict__ x, int64_t B,
                                                               float* __restrict__ feat_tiled, float* __restrict__ feat_rows,
                                                               float* __restrict__ aux_tiled) {
  extern __shared__ float featL[];               // [d_r][kGroup] when feat_tiled != nullptr
  __shared__ float bc[kGroup][CVF_AUX_ROWS + 2]; // per frame: R (9), c (3), Kinv (6)
  __shared__ double cD[kGroup][3];
  __shared__ double sums[kGroup][16];
  const int tid = threadIdx.x, lane = tid & 63, fi = tid >> 6;
  const int nc = pp.n_coord, nal = pp.n_align;
  const int64_t f0 = (int64_t)blockIdx.x * kGroup;
  // lanes of the last tile past B replicate frame B-1 (weight 0 downstream): every tile is fully written
  const bool real = f0 + fi < B;
  const int64_t frame = real ? f0 + fi : B - 1;
  const float* __restrict__ xf = x + frame * nc;
  // ---- stream
  double acc[15];
#pragma unroll
  for (int i = 0; i < 15; ++i) acc[i] = 0.0;
#pragma unroll 8
  for (int b = lane; b < nal; b += 64) {
    const int a = CONTIG ? b : pp.align_idx[b];
    const float x0 = xf[3 * a], x1 = xf[3 * a + 1], x2 = xf[3 * a + 2];
    const float r0 = pp.ref_c[3 * b], r1 = pp.ref_c[3 * b + 1], r2 = pp.ref_c[3 * b + 2];
    const double d0 = (double)x0, d1 = (double)x1, d2 = (double)x2;
    const double e0 = (double)r0, e1 = (double)r1, e2 = (double)r2;
    acc[0] += d0; acc[1] += d1; acc[2] += d2;
    acc[3] = fma(d0, e0, acc[3]); acc[4] = fma(d0, e1, acc[4]); acc[5] = fma(d0, e2, acc[5]);
    acc[6] = fma(d1, e0, acc[6]); acc[7] = fma(d1, e1, acc[7]); acc[8] = fma(d1, e2, acc[8]);
    acc[9] = fma(d2, e0, acc[9]); acc[10] = fma(d2, e1, acc[10]); acc[11] = fma(d2, e2, acc[11]);
    acc[12] += e0; acc[13] += e1; acc[14] += e2;
  }
  {
    double mine = 0.0;
#pragma unroll
    for (int i = 0; i < 15; ++i) {
      const double sv = wave_sum(acc[i]);
      if (lane == i) mine = sv;
    }
    if (lane < 15) sums[fi][lane] = mine;
  }
  __syncthreads();
  // ---- solve: the workgroup's kGroup 3x3 problems run on kGroup LANES of one wave (one instruction stream
  //      for all of them).  A solve on lane 0 of every wave costs the SIMD the same issue slots as a full wave
  //      and made the kernel fp64-issue-bound instead of HBM-bound.
  if (tid < kGroup) {
    const double* t = sums[tid];
    const double inv = fast_rcp((double)nal);
    const double c0 = t[0] * inv, c1 = t[1] * inv, c2 = t[2] * inv;
    double H[3][3];
    H[0][0] = t[3] - c0 * t[12]; H[0][1] = t[4] - c0 * t[13]; H[0][2] = t[5] - c0 * t[14];
    H[1][0] = t[6] - c1 * t[12]; H[1][1] = t[7] - c1 * t[13]; H[1][2] = t[8] - c1 * t[14];
    H[2][0] = t[9] - c2 * t[12]; H[2][1] = t[10] - c2 * t[13]; H[2][2] = t[11] - c2 * t[14];
    KabschOut ko;
    if (aux_tiled != nullptr) {   // (uniform) rotation + K^-1 for the derivative kernel
      kabsch_from_H<true>(H, ko);
    } else {                      // features only: the rotation alone, one Newton step less (as k1_stream_kernel)
      kabsch_from_H<false>(H, ko);
#pragma unroll
      for (int i = 0; i < 6; ++i) ko.Kinv[i] = 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) bc[tid][i] = ko.R[i];
    bc[tid][9] = (float)c0; bc[tid][10] = (float)c1; bc[tid][11] = (float)c2;
#pragma unroll
    for (int i = 0; i < 6; ++i) bc[tid][12 + i] = ko.Kinv[i];
    cD[tid][0] = c0; cD[tid][1] = c1; cD[tid][2] = c2;
  }
  __syncthreads();
  const int64_t tile = f0 / CVF_TILE;
  const int l0 = (int)(f0 % CVF_TILE);  // 0, 16, 32 or 48
  if (aux_tiled != nullptr && lane < CVF_AUX_ROWS) aux_tiled[(tile * CVF_AUX_ROWS + lane) * CVF_TILE + l0 + fi] = bc[fi][lane];
  // ---- features of this wave's frame
  float R[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = bc[fi][i];
  const double cc0 = cD[fi][0], cc1 = cD[fi][1], cc2 = cD[fi][2];
  // (rows of padded frames land in the last real frame's row: same values)
  float* fr = feat_rows ? feat_rows + frame * pp.d_r : nullptr;
  const bool tiled = feat_tiled != nullptr;
  float* fl = featL + fi;
  auto emit = [&](int o, float v) {
    if (tiled) fl[o * kGroup] = v;
    if (fr) fr[o] = v;
  };
  for (int r = lane; r < pp.n_rec; r += 64) {
    const int32_t* p = pp.rec + 6 * r;
    const Rec rc{p[0], p[1], p[2], p[3], p[4], p[5]};
    if (rc.type == CVF_FEAT_POSITION) {
      const V3 xa = gatom(xf, rc.a0);
      const V3 xc = v3((float)((double)xa.x - cc0), (float)((double)xa.y - cc1), (float)((double)xa.z - cc2));
      const V3 al = row_times(xc, R);
      emit(rc.out, al.x);
      emit(rc.out + 1, al.y);
      emit(rc.out + 2, al.z);
    } else if (rc.type == CVF_FEAT_BOND) {
      emit(rc.out, bond_eval(gatom(xf, rc.a0), gatom(xf, rc.a1)).val);
    } else if (rc.type == CVF_FEAT_ANGLE) {
      const float cs = angle_eval(gatom(xf, rc.a0), gatom(xf, rc.a1), gatom(xf, rc.a2)).cs;
      emit(rc.out, pp.use_angle_value ? acosf(cs) : cs);
    } else {
      const DihedralG dg = dihedral_eval(gatom(xf, rc.a0), gatom(xf, rc.a1), gatom(xf, rc.a2), gatom(xf, rc.a3));
      if (pp.use_angle_value) {
        emit(rc.out, atan2f(dg.sn, dg.cs));
      } else {
        emit(rc.out, dg.cs);
        emit(rc.out + 1, dg.sn);
      }
    }
  }
  // ---- flush the 16 frames' features as 64-byte segments
  if (feat_tiled != nullptr) {
    __syncthreads();
    for (int idx = tid; idx < pp.d_r * kGroup; idx += 64 * kGroup) {
      const int o = idx / kGroup, f = idx % kGroup;
      feat_tiled[(tile * pp.d_r + o) * CVF_TILE + l0 + f] = featL[idx];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Capture variant (the fast path).  While a wave streams its frame it copies the coordinates of the atoms
// the features need ("slots", a few hundred of the thousands) into LDS; the records are then evaluated from
// LDS.  HBM traffic is the frame once plus the outputs - the gather variant above re-fetches most of the
// frame's cache lines (a random 10 % of the atoms touches > 70 % of the 128-byte lines).
// VEC4: N % 4 == 0 and the align atoms are the first n_align (multiple of 4) atoms: a lane takes 4 atoms =
// 48 contiguous bytes = three 16-byte loads (coordinates and reference alike) instead of twelve 4-byte ones.
// ---------------------------------------------------------------------------------------------------------
struct Acc15 {
  double v[15];
};
__device__ __forceinline__ void acc_atom(Acc15& A, float x0, float x1, float x2, float r0, float r1, float r2) {
  const double d0 = (double)x0, d1 = (double)x1, d2 = (double)x2;
  const double e0 = (double)r0, e1 = (double)r1, e2 = (double)r2;
  A.v[0] += d0; A.v[1] += d1; A.v[2] += d2;
  A.v[3] = fma(d0, e0, A.v[3]); A.v[4] = fma(d0, e1, A.v[4]); A.v[5] = fma(d0, e2, A.v[5]);
  A.v[6] = fma(d1, e0, A.v[6]); A.v[7] = fma(d1, e1, A.v[7]); A.v[8] = fma(d1, e2, A.v[8]);
  A.v[9] = fma(d2, e0, A.v[9]); A.v[10] = fma(d2, e1, A.v[10]); A.v[11] = fma(d2, e2, A.v[11]);
  A.v[12] += e0; A.v[13] += e1; A.v[14] += e2;
}

// Everything after the streaming pass, shared by the two captured-atom kernels: the workgroup's kGroup 3x3
// problems (one per lane of one wave) from sums[frame][0..14] = {sum x (3), sum x (x) ref (9), sum ref (3)}, then
// per frame (wave fi) aux, the compact feature-atom copy, the features from the captured atoms, and the flush.
__device__ __forceinline__ void large_solve_features(const cvf_pp_desc& pp, int64_t B, int64_t f0, int tid, int lane, int fi,
                                                     bool /*real*/, int64_t /*frame*/, const double (*sums)[16],
                                                     float (*bc)[CVF_AUX_ROWS + 2], double (*cD)[3], const float* capL,
                                                     float* featL, float* __restrict__ feat_tiled,
                                                     float* __restrict__ feat_rows, float* __restrict__ aux_tiled,
                                                     float* __restrict__ slot_xyz) {
  const int nal = pp.n_align, nslot = pp.n_slot;
  CVF_STAMP(4);
  constexpr int kRecPre = 5;
  const int nrs = pp.n_rec_slot > 0 ? pp.n_rec_slot : pp.n_rec;   // batched lists carry padding entries (type -1)
  Rec pre[kRecPre];
#pragma unroll
  for (int it = 0; it < kRecPre; ++it) {
    const int r = lane + 64 * it;
    const int32_t* p = pp.rec_slot + 6 * (r < nrs ? r : nrs - 1);
    pre[it] = Rec{p[0], p[1], p[2], p[3], p[4], p[5]};
  }
  // ---- staging of the features (decided here: the invariant features below are emitted before the solve is known)
  // staged == 1: the features go through LDS as [feature][frame] for the tiled output (rows, if also wanted, straight from the lanes);
  // staged == 2 (row-major output alone - what AutoEncoderTask's one-off feature trajectory is, core.py:635): as [frame][feature], the
  // workgroup's eight rows then leave as ONE contiguous run in 16-byte stores (round 4: this flavour used to fall to the gather
  // kernel - 1473 us per 100 000 frames of 5000 atoms against 1060)
  const int staged = feat_tiled != nullptr ? 1 : (featL != nullptr && feat_rows != nullptr ? 2 : 0);
  const float* capBase = capL - (size_t)fi * nslot * 3;      // [kGroup][nslot * 3]: every frame's captured atoms
  // features of frame `fs` of the group (fs = fi: this wave's own; fs = 0: a share of wave 0's, see below)
  auto emit_to = [&](int fs, int o, float v) {
    if (staged == 1) featL[o * kGroup + fs] = v;
    else if (staged == 2) featL[fs * pp.d_r + o] = v;
    if (staged != 2 && feat_rows != nullptr && f0 + fs < B) feat_rows[(f0 + fs) * pp.d_r + o] = v;
  };
  // bonds, angles and dihedrals do not see the alignment: they are evaluated WHILE wave 0 solves the group's eight 3x3 problems
  // (round 4; stamped tail of a workgroup: barrier 2.5 k | solve 6.1 k | features 5.6 k | flush 1.6 k cycles behind 41 k of streaming
  // - solve and features now run side by side).  Wave 0's own frame is shared out: wave w takes its record batches it = w - 1 (mod 7).
  auto invariant = [&](const Rec& rc, int fs) {
    if (rc.type < 0 || rc.type == CVF_FEAT_POSITION) return;
    const float* cp = capBase + (size_t)fs * nslot * 3;
    auto sat = [&](int sl) { return V3{cp[3 * sl], cp[3 * sl + 1], cp[3 * sl + 2]}; };
    if (rc.type == CVF_FEAT_BOND) {
      emit_to(fs, rc.out, bond_eval(sat(rc.a0), sat(rc.a1)).val);
    } else if (rc.type == CVF_FEAT_ANGLE) {
      const float cs = angle_eval(sat(rc.a0), sat(rc.a1), sat(rc.a2)).cs;
      emit_to(fs, rc.out, pp.use_angle_value ? acosf(cs) : cs);
    } else {
      const DihedralG dg = dihedral_eval(sat(rc.a0), sat(rc.a1), sat(rc.a2), sat(rc.a3));
      if (pp.use_angle_value) {
        emit_to(fs, rc.out, atan2f(dg.sn, dg.cs));
      } else {
        emit_to(fs, rc.out, dg.cs);
        emit_to(fs, rc.out + 1, dg.sn);
      }
    }
  };
  if (tid < kGroup) {  // the group's 3x3 problems, one per lane of one wave
    const double* t = sums[tid];
    const double inv = fast_rcp((double)nal);
    const double c0 = t[0] * inv, c1 = t[1] * inv, c2 = t[2] * inv;
    double H[3][3];
    H[0][0] = t[3] - c0 * t[12]; H[0][1] = t[4] - c0 * t[13]; H[0][2] = t[5] - c0 * t[14];
    H[1][0] = t[6] - c1 * t[12]; H[1][1] = t[7] - c1 * t[13]; H[1][2] = t[8] - c1 * t[14];
    H[2][0] = t[9] - c2 * t[12]; H[2][1] = t[10] - c2 * t[13]; H[2][2] = t[11] - c2 * t[14];
    KabschOut ko;
    if (aux_tiled != nullptr) {   // (uniform) rotation + K^-1 for the derivative kernel
      kabsch_from_H<true>(H, ko);
    } else {                      // features only: the rotation alone, one Newton step less (as k1_stream_kernel)
      kabsch_from_H<false>(H, ko);
#pragma unroll
      for (int i = 0; i < 6; ++i) ko.Kinv[i] = 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) bc[tid][i] = ko.R[i];
    bc[tid][9] = (float)c0; bc[tid][10] = (float)c1; bc[tid][11] = (float)c2;
#pragma unroll
    for (int i = 0; i < 6; ++i) bc[tid][12 + i] = ko.Kinv[i];
    cD[tid][0] = c0; cD[tid][1] = c1; cD[tid][2] = c2;
  } else if (tid >= 64) {
    if (slot_xyz != nullptr) {
      // meanwhile the other waves write the compact copy of the feature atoms for the derivative kernel (metric_large.hip), which
      // works with one frame per lane: coordinate rows of this workgroup's kGroup frames, [frame group][n_slot * 3][kGroup] - one
      // contiguous run per workgroup, written in 16-byte pieces.  (As rows of the 64-frame tile - 32 bytes per row and workgroup -
      // the partial lines of eight workgroups cost 1406-1733 us per 100 000 frames against 1231-1321 us.)
      static_assert(kGroup == 8, "two 16-byte pieces per row");
      typedef float nt4 __attribute__((ext_vector_type(4)));
      const float* img = capL - (size_t)fi * nslot * 3;      // [kGroup][nslot * 3]
      const int ns3 = nslot * 3;
      nt4* dst = reinterpret_cast<nt4*>(slot_xyz + (f0 / kGroup) * (int64_t)ns3 * kGroup);
      for (int i = tid - 64; i < 2 * ns3; i += 64 * (kGroup - 1)) {
        const int row = i >> 1, fr = 4 * (i & 1);
        const nt4 v = {img[fr * ns3 + row], img[(fr + 1) * ns3 + row], img[(fr + 2) * ns3 + row], img[(fr + 3) * ns3 + row]};
        __builtin_nontemporal_store(v, dst + i);   // written once, read by a later kernel
      }

    }
    // this wave's frame, then its share of wave 0's frame
#pragma unroll
    for (int it = 0; it < kRecPre; ++it)
      if (lane + 64 * it < nrs) invariant(pre[it], fi);
    for (int r = lane + 64 * kRecPre; r < nrs; r += 64) {
      const int32_t* p = pp.rec_slot + 6 * r;
      invariant(Rec{p[0], p[1], p[2], p[3], p[4], p[5]}, fi);
    }
#pragma unroll
    for (int it = 0; it < kRecPre; ++it)
      if (it % (kGroup - 1) == fi - 1 && lane + 64 * it < nrs) invariant(pre[it], 0);
    for (int it = kRecPre; 64 * it < nrs; ++it)
      if (it % (kGroup - 1) == fi - 1 && lane + 64 * it < nrs) {
        const int32_t* p = pp.rec_slot + 6 * (lane + 64 * it);
        invariant(Rec{p[0], p[1], p[2], p[3], p[4], p[5]}, 0);
      }
  }
  __syncthreads();
  CVF_STAMP(5);
  const int64_t tile = f0 / CVF_TILE;
  const int l0 = (int)(f0 % CVF_TILE);
  if (aux_tiled != nullptr && lane < CVF_AUX_ROWS) aux_tiled[(tile * CVF_AUX_ROWS + lane) * CVF_TILE + l0 + fi] = bc[fi][lane];
  CVF_STAMP(6);
  float R[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = bc[fi][i];
  const double cc0 = cD[fi][0], cc1 = cD[fi][1], cc2 = cD[fi][2];
  // ---- the position features (the only ones that need the rotation), every wave its own frame
  auto position = [&](const Rec& rc) {
    if (rc.type != CVF_FEAT_POSITION) return;
    const V3 xa = V3{capL[3 * rc.a0], capL[3 * rc.a0 + 1], capL[3 * rc.a0 + 2]};
    const V3 xc = v3((float)((double)xa.x - cc0), (float)((double)xa.y - cc1), (float)((double)xa.z - cc2));
    const V3 al = row_times(xc, R);
    emit_to(fi, rc.out, al.x);
    emit_to(fi, rc.out + 1, al.y);
    emit_to(fi, rc.out + 2, al.z);
  };
#pragma unroll
  for (int it = 0; it < kRecPre; ++it)
    if (lane + 64 * it < nrs) position(pre[it]);
  for (int r = lane + 64 * kRecPre; r < nrs; r += 64) {
    const int32_t* p = pp.rec_slot + 6 * r;   // like rec, atom fields hold slots
    const Rec rc{p[0], p[1], p[2], p[3], p[4], p[5]};
    position(rc);
  }
  CVF_STAMP(7);
  if (staged == 1) {
    __syncthreads();
    static_assert(kGroup == 8, "a row's piece of this group: two 16-byte stores");
    for (int idx = tid; idx < pp.d_r * 2; idx += 64 * kGroup) {   // (non-temporal stores here: no difference, 1217 vs 1217 us)
      const int o = idx >> 1, h = 4 * (idx & 1);
      *reinterpret_cast<float4*>(feat_tiled + (tile * pp.d_r + o) * CVF_TILE + l0 + h) = *reinterpret_cast<const float4*>(featL + o * kGroup + h);
    }
  } else if (staged == 2) {
    __syncthreads();
    const int64_t nreal = B - f0 < kGroup ? B - f0 : kGroup;     // frames of this workgroup inside the batch
    float* dst = feat_rows + f0 * pp.d_r;
    const int total = (int)nreal * pp.d_r;
    if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0 && (total & 3) == 0) {
      const float4* s4 = reinterpret_cast<const float4*>(featL);
      float4* d4 = reinterpret_cast<float4*>(dst);
      for (int i = tid; i < total / 4; i += 64 * kGroup) d4[i] = s4[i];
    } else {
      for (int i = tid; i < total; i += 64 * kGroup) dst[i] = featL[i];
    }
  }
  CVF_STAMP(8);
}

// fp32 partial sums of the same fifteen quantities over a few atoms (see the VEC4 loop)
__device__ __forceinline__ void part_first(float (&s)[15], float x0, float x1, float x2, float r0, float r1, float r2) {
  s[0] = x0; s[1] = x1; s[2] = x2;
  s[3] = x0 * r0; s[4] = x0 * r1; s[5] = x0 * r2;
  s[6] = x1 * r0; s[7] = x1 * r1; s[8] = x1 * r2;
  s[9] = x2 * r0; s[10] = x2 * r1; s[11] = x2 * r2;
  s[12] = r0; s[13] = r1; s[14] = r2;
}
__device__ __forceinline__ void part_next(float (&s)[15], float x0, float x1, float x2, float r0, float r1, float r2) {
  s[0] += x0; s[1] += x1; s[2] += x2;
  s[3] = fmaf(x0, r0, s[3]); s[4] = fmaf(x0, r1, s[4]); s[5] = fmaf(x0, r2, s[5]);
  s[6] = fmaf(x1, r0, s[6]); s[7] = fmaf(x1, r1, s[7]); s[8] = fmaf(x1, r2, s[8]);
  s[9] = fmaf(x2, r0, s[9]); s[10] = fmaf(x2, r1, s[10]); s[11] = fmaf(x2, r2, s[11]);
  s[12] += r0; s[13] += r1; s[14] += r2;
}

template <bool VEC4>
__global__ __launch_bounds__(64 * kGroup) void k1_large_capture_kernel(cvf_pp_desc pp, const float* __restrict__ x, int64_t B,
                                                                       float* __restrict__ feat_tiled,
                                                                       float* __restrict__ feat_rows,
                                                                       float* __restrict__ aux_tiled,
                                                                       float* __restrict__ slot_xyz, int same_xcd) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];                  // [kGroup][n_slot][3] captured atoms, then [d_r][kGroup] features
  __shared__ float bc[kGroup][CVF_AUX_ROWS + 2];
  __shared__ double cD[kGroup][3];
  __shared__ double sums[kGroup][16];
  const int tid = threadIdx.x, lane = tid & 63, fi = tid >> 6;
  const int nc = pp.n_coord, nal = pp.n_align, nslot = pp.n_slot, N = nc / 3;
  float* capL = dyn + (size_t)fi * nslot * 3;
  float* featL = dyn + (size_t)kGroup * nslot * 3;
  const int64_t f0 = k1_group_of_block((same_xcd & 1) != 0) * kGroup;
  const bool real = f0 + fi < B;
  const int64_t frame = real ? f0 + fi : B - 1;
  const float* __restrict__ xf = x + frame * nc;
  Acc15 A;
#pragma unroll
  for (int i = 0; i < 15; ++i) A.v[i] = 0.0;
  if (VEC4) {
    const float4* __restrict__ x4 = reinterpret_cast<const float4*>(xf);
    const float4* __restrict__ r4 = reinterpret_cast<const float4*>(pp.ref_c);
    const int4* __restrict__ s4 = reinterpret_cast<const int4*>(pp.atom_slot);
    const int nq = N >> 2, nqa = nal >> 2;
#pragma unroll 4
    for (int g = lane; g < nq; g += 64) {
      const float4 a = x4[3 * g], b = x4[3 * g + 1], c = x4[3 * g + 2];  // atoms 4g..4g+3
      const int4 sl = s4[g];
      if (g < nqa) {
        // The four atoms' sums are formed in fp32 and only those partial sums enter the fp64 accumulators: 30 instead
        // of 84 double-rate instructions per 48 bytes (with all of it in fp64 this loop was VALU-bound at ~4.4 TB/s,
        // tools/stream_probe.hip reads 6.3 TB/s with the same load pattern).  A partial sum of four products of
        // O(1e3) magnitude carries ~4e-4 absolute rounding; over 1250 of them that is 1e-8 of the covariance entries.
        const float4 p = r4[3 * g], q = r4[3 * g + 1], r = r4[3 * g + 2];
        float s[15];
        part_first(s, a.x, a.y, a.z, p.x, p.y, p.z);
        part_next(s, a.w, b.x, b.y, p.w, q.x, q.y);
        part_next(s, b.z, b.w, c.x, q.z, q.w, r.x);
        part_next(s, c.y, c.z, c.w, r.y, r.z, r.w);
#pragma unroll
        for (int i = 0; i < 15; ++i) A.v[i] += (double)s[i];
      }
      if (sl.x >= 0) { capL[3 * sl.x] = a.x; capL[3 * sl.x + 1] = a.y; capL[3 * sl.x + 2] = a.z; }
      if (sl.y >= 0) { capL[3 * sl.y] = a.w; capL[3 * sl.y + 1] = b.x; capL[3 * sl.y + 2] = b.y; }
      if (sl.z >= 0) { capL[3 * sl.z] = b.z; capL[3 * sl.z + 1] = b.w; capL[3 * sl.z + 2] = c.x; }
      if (sl.w >= 0) { capL[3 * sl.w] = c.y; capL[3 * sl.w + 1] = c.z; capL[3 * sl.w + 2] = c.w; }
    }
  } else {
#pragma unroll 4
    for (int a = lane; a < N; a += 64) {
      const float x0 = xf[3 * a], x1 = xf[3 * a + 1], x2 = xf[3 * a + 2];
      const int b = pp.atom_align[a];
      const int sl = pp.atom_slot[a];
      if (b >= 0) acc_atom(A, x0, x1, x2, pp.ref_c[3 * b], pp.ref_c[3 * b + 1], pp.ref_c[3 * b + 2]);
      if (sl >= 0) { capL[3 * sl] = x0; capL[3 * sl + 1] = x1; capL[3 * sl + 2] = x2; }
    }
  }
  {
    double mine = 0.0;
#pragma unroll
    for (int i = 0; i < 15; ++i) {
      const double sv = wave_sum(A.v[i]);
      if (lane == i) mine = sv;
    }
    if (lane < 15) sums[fi][lane] = mine;
  }
  __syncthreads();
  const bool staging = feat_tiled != nullptr || feat_rows != nullptr;   // (the launch reserves the staging area for either output)
  large_solve_features(pp, B, f0, tid, lane, fi, real, frame, sums, bc, cD, capL, staging ? featL : nullptr, feat_tiled, feat_rows, aux_tiled,
                       slot_xyz);
}

// ---------------------------------------------------------------------------------------------------------
// Slice variant of the streaming pass (VEC4 layouts, N <= 8192): the eight waves of a workgroup split the ATOMS of
// a frame (wave w, step i owns the 4-atom groups (8 i + w) 64 + lane) and walk through the workgroup's eight frames
// together.  A lane then meets the same atoms in every frame, so their reference coordinates and capture slots are
// loaded ONCE into registers: the streaming loop issues coordinate loads only.  (With one frame per wave each wave
// re-read the 12 N-byte reference and the slot table from L2 for every frame - twice the vector-memory
// instructions per byte of coordinates - and tools/stream_probe.hip shows exactly that costing 6.2 -> 3.9 TB/s.)
// Per frame a wave reduces its twelve partial sums through a small wave-private LDS transpose (4 values a round:
// 4 writes, one 16-byte read, four DPP steps) instead of twelve full-wave butterflies.
// ---------------------------------------------------------------------------------------------------------
// (min. 4 waves per SIMD = 128 registers = TWO workgroups per CU, which this kernel's loop without prefetch relies on.  Round 4's first edits of
//  the tail took it to 130 registers - 3 waves per SIMD, ONE 8-wave workgroup per CU - unnoticed for most of the round: 1131-1221 us per 100 000
//  frames of 5000 atoms instead of 1078-1153.  NI = 4 (6145..8192 atoms) would spill 17 registers at 128 and stays at one workgroup per CU.)
template <int NI, bool NT>
__global__ __launch_bounds__(64 * kGroup, NI <= 3 ? 4 : 2) void k1_large_slice_kernel(cvf_pp_desc pp, const float* __restrict__ x, int64_t B,
                                                                     float* __restrict__ feat_tiled,
                                                                     float* __restrict__ feat_rows,
                                                                     float* __restrict__ aux_tiled,
                                                                     float* __restrict__ slot_xyz, int same_xcd) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];                  // [kGroup][n_slot][3] captured atoms | [d_r][kGroup] features (first: reduction scratch)
  __shared__ float bc[kGroup][CVF_AUX_ROWS + 2];
  __shared__ double cD[kGroup][3];
  __shared__ double sums[kGroup][16];
  __shared__ double part[kGroup][kGroup][12];     // [frame][wave][value]
  __shared__ double rpart[kGroup][3];             // per-wave sums of the reference rows it owns
  constexpr int kRedPitch = 68;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int nc = pp.n_coord, nal = pp.n_align, nslot = pp.n_slot, N = nc / 3;
  const int nq = N >> 2, nqa = nal >> 2;
  float* featL = dyn + (size_t)kGroup * nslot * 3;
  float* red = featL + w * (4 * kRedPitch);       // wave-private, reused by the feature staging later
  const int64_t f0 = k1_group_of_block((same_xcd & 1) != 0) * kGroup;
  CVF_STAMP(0);
  // ---- this lane's atoms: reference rows, capture slots (registers for the whole batch)
  const float4* __restrict__ r4 = reinterpret_cast<const float4*>(pp.ref_c);
  const int4* __restrict__ s4 = reinterpret_cast<const int4*>(pp.atom_slot);
  float4 rp[NI], rq[NI], rr[NI];
  int4 sl[NI];
  int gi[NI];
  float mk[NI];
  float rs0 = 0.0f, rs1 = 0.0f, rs2 = 0.0f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int g = (i * kGroup + w) * 64 + lane;
    const bool valid = g < nq, al = g < nqa;
    gi[i] = valid ? g : nq - 1;
    const int ga = al ? g : 0;
    const float4 z = {0.0f, 0.0f, 0.0f, 0.0f};
    const float4 p = r4[3 * ga], q = r4[3 * ga + 1], r = r4[3 * ga + 2];
    rp[i] = al ? p : z;
    rq[i] = al ? q : z;
    rr[i] = al ? r : z;
    mk[i] = al ? 1.0f : 0.0f;
    const int4 sv = s4[gi[i]];
    sl[i] = valid ? sv : int4{-1, -1, -1, -1};
    rs0 += rp[i].x + rp[i].w + rq[i].z + rr[i].y;
    rs1 += rp[i].y + rq[i].x + rq[i].w + rr[i].z;
    rs2 += rp[i].z + rq[i].y + rr[i].x + rr[i].w;
  }
  {
    const double t0 = wave_sum((double)rs0), t1 = wave_sum((double)rs1), t2 = wave_sum((double)rs2);
    if (lane == 0) { rpart[w][0] = t0; rpart[w][1] = t1; rpart[w][2] = t2; }
  }
  CVF_STAMP(1);
  // ---- the batch's frames, one after the other
  // (issuing the next frame's loads before this frame's arithmetic was tried: 36 more live registers push the
  // kernel past 128 VGPRs, i.e. to one workgroup per CU or into spills - 2.8 TB/s instead of 4.5)
#pragma unroll 1
  for (int j = 0; j < kGroup; ++j) {
    const int64_t frame = f0 + j < B ? f0 + j : B - 1;
    typedef float nt4 __attribute__((ext_vector_type(4)));
    const nt4* __restrict__ x4 = reinterpret_cast<const nt4*>(x + frame * nc);
    float* capL = dyn + (size_t)j * nslot * 3;
    nt4 a[NI], b[NI], c[NI];
    // NT (off by default, see the launch code): non-temporal loads for the once-read coordinate stream
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      a[i] = NT ? __builtin_nontemporal_load(x4 + 3 * gi[i]) : x4[3 * gi[i]];
      b[i] = NT ? __builtin_nontemporal_load(x4 + 3 * gi[i] + 1) : x4[3 * gi[i] + 1];
      c[i] = NT ? __builtin_nontemporal_load(x4 + 3 * gi[i] + 2) : x4[3 * gi[i] + 2];
    }
    float s[12];
#pragma unroll
    for (int v = 0; v < 12; ++v) s[v] = 0.0f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      // (rows of atoms outside the align set are zero, their coordinates are masked out of the centroid)
      auto atom4 = [&](float x0, float x1, float x2, float r0, float r1, float r2) {
        s[0] = fmaf(mk[i], x0, s[0]); s[1] = fmaf(mk[i], x1, s[1]); s[2] = fmaf(mk[i], x2, s[2]);
        s[3] = fmaf(x0, r0, s[3]); s[4] = fmaf(x0, r1, s[4]); s[5] = fmaf(x0, r2, s[5]);
        s[6] = fmaf(x1, r0, s[6]); s[7] = fmaf(x1, r1, s[7]); s[8] = fmaf(x1, r2, s[8]);
        s[9] = fmaf(x2, r0, s[9]); s[10] = fmaf(x2, r1, s[10]); s[11] = fmaf(x2, r2, s[11]);
      };
      atom4(a[i].x, a[i].y, a[i].z, rp[i].x, rp[i].y, rp[i].z);
      atom4(a[i].w, b[i].x, b[i].y, rp[i].w, rq[i].x, rq[i].y);
      atom4(b[i].z, b[i].w, c[i].x, rq[i].z, rq[i].w, rr[i].x);
      atom4(c[i].y, c[i].z, c[i].w, rr[i].y, rr[i].z, rr[i].w);
      if (sl[i].x >= 0) { capL[3 * sl[i].x] = a[i].x; capL[3 * sl[i].x + 1] = a[i].y; capL[3 * sl[i].x + 2] = a[i].z; }
      if (sl[i].y >= 0) { capL[3 * sl[i].y] = a[i].w; capL[3 * sl[i].y + 1] = b[i].x; capL[3 * sl[i].y + 2] = b[i].y; }
      if (sl[i].z >= 0) { capL[3 * sl[i].z] = b[i].z; capL[3 * sl[i].z + 1] = b[i].w; capL[3 * sl[i].z + 2] = c[i].x; }
      if (sl[i].w >= 0) { capL[3 * sl[i].w] = c[i].y; capL[3 * sl[i].w + 1] = c[i].z; capL[3 * sl[i].w + 2] = c[i].w; }
    }
    // wave reduction of the 12 sums, 4 per round (the per-lane sums cover <= 4 NI atoms: fp32; from here on fp64)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int v = 0; v < 4; ++v) red[v * kRedPitch + lane] = s[4 * r + v];
      const float4 t = *reinterpret_cast<const float4*>(red + (lane >> 4) * kRedPitch + 4 * (lane & 15));
      double d = ((double)t.x + (double)t.y) + ((double)t.z + (double)t.w);
      d += dpp_movd<0x111, 0xf>(d);
      d += dpp_movd<0x112, 0xf>(d);
      d += dpp_movd<0x114, 0xf>(d);
      d += dpp_movd<0x118, 0xf>(d);
      if ((lane & 15) == 15) part[j][w][4 * r + (lane >> 4)] = d;
    }
  }
  CVF_STAMP(2);
  if (same_xcd & 2) return;   // developer probe (CVF_K1_XCD=2 / 3): the streaming loop alone - wrong results, its time is the point
  __syncthreads();
  CVF_STAMP(3);
  if (tid < kGroup * 15) {   // sums[frame][value] over the eight waves, fixed order
    const int j = tid / 15, v = tid - 15 * j;
    double t = 0.0;
    if (v < 12) {
#pragma unroll
      for (int ww = 0; ww < kGroup; ++ww) t += part[j][ww][v];
    } else {
#pragma unroll
      for (int ww = 0; ww < kGroup; ++ww) t += rpart[ww][v - 12];
    }
    sums[j][v] = t;
  }
  __syncthreads();
  const int fi = w;
  const bool real = f0 + fi < B;
  const int64_t frame = real ? f0 + fi : B - 1;
  const float* capL = dyn + (size_t)fi * nslot * 3;
  large_solve_features(pp, B, f0, tid, lane, fi, real, frame, sums, bc, cD, capL, featL, (same_xcd & 4) ? nullptr : feat_tiled, feat_rows, aux_tiled,
                       slot_xyz);   // (same_xcd & 4: developer probe - the launch without its tiled feature stores)
}


// ---------------------------------------------------------------------------------------------------------
// Pipelined variant of the slice kernel (round 4; large batches, tiled outputs).  In the slice kernel a workgroup
// streams its eight frames and then spends a fifth of its life in the tail (sums -> 3x3 solves -> features -> flush) with no load in
// flight; two workgroups per CU overlap that only by chance, and the streaming loop alone (CVF_K1_XCD=2) runs at the read sweep's rate
// while the whole kernel is 10-15 % below it.  Here ONE workgroup per CU stays for the whole launch and is split by role:
//   waves 0..7   stream: exactly the slice kernel's loop (same atoms per lane, same partial sums, same order: bit-identical sums),
//                but over group after group, and with the NEXT frame's loads requested before the current frame's arithmetic
//                (one workgroup per CU leaves 168 registers per lane: room for the second set of 36) - also across the group
//                boundary, so the memory pipe never drains;
//   wave 8       tail: the group's eight 3x3 solves (a lane each), the rotation rows, the features that need the rotation;
//   waves 9..11  tail: the other features, the slot copy - and the feature stores: the features of FOUR consecutive groups wait in
//                their registers and leave as whole 128-byte lines (see THE FLUSH in the kernel) - while the streaming waves are already
//                capturing the next group into the other LDS buffer.
// The roles meet at counters in LDS (in the kernel), never at the workgroup barrier.  Measured against the slice kernel AT TWO WORKGROUPS PER CU
// (tools/k1_ab.sh / k1_ab2.sh, 100 000 frames of 5000 atoms, both kernels in one lease, two kinds of box): tiled features 1065 / 1118-1122 us
// against 1078 / 1153-1156 (0.72-0.735 and 0.685-0.69 of 8 TB/s); with the generator-mode extras 1254 / 1327-1331 against 1271 / 1343-1350;
// row-major output 1067 / 1123-1129 against 1032 / 1110-1115 (the slice kernel's 12-KB runs are whole lines already) - so it runs where it pays
// (tiled features alone, large batches: see the launch code), a 1-4 % matter.  (For most of round 4 the comparison read "-9 %": the slice kernel had grown to 130 registers - one
// workgroup per CU - in the same round.)  What the probes of this kernel established holds for both: without their feature stores the slice
// kernel runs at 949 us = the read sweep of the same bytes (950-960), this one at 983-989 (streaming waves alone 968-975); the stores - 2.5 % of
// the bytes - are the whole distance to the sweep, 200 us as 32-byte pieces, 135 as whole lines.  Every output is bit for bit the slice kernel's
// (tests/test_gpu_parity.py::test_k1_pipelined_kernel_equals_the_slice_kernel_and_the_oracle, tools/k1_pipe_fuzz.py).
// ---------------------------------------------------------------------------------------------------------
constexpr int kStream = 8, kTail = 4;
// developer aid (tools/k1_large_probe.hip, -DCVF_STAMPS): cycles a wave spends between two marks, summed over its groups
#ifdef CVF_STAMPS
#define PIPE_T(i)                                                            \
  do {                                                                       \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();            \
    acc_[i] += now_ - last_;                                                 \
    last_ = now_;                                                            \
  } while (0)
#define PIPE_T_OUT()                                                                                                   \
  do {                                                                                                                 \
    if (lane == 0)                                                                                                     \
      for (int i_ = 0; i_ < 9; ++i_) g_stamps[((blockIdx.x * (kStream + kTail) + w) % 4096) * 64 + i_] = acc_[i_];     \
  } while (0)
#else
#define PIPE_T(i) do {} while (0)
#define PIPE_T_OUT() do {} while (0)
#endif
static_assert(kStream == kGroup && kTail == 4, "slice mapping: eight streaming waves; one solving + three holding tail waves");
constexpr int kRedPitchP = 68;

template <int NI>
__global__ __launch_bounds__(64 * (kStream + kTail)) void k1_large_pipe_kernel(cvf_pp_desc pp, const float* __restrict__ x, int64_t B,
                                                                              int64_t nquads, int64_t groups, int rounds,
                                                                              float* __restrict__ feat_tiled,
                                                                              float* __restrict__ feat_rows,
                                                                              float* __restrict__ aux_tiled,
                                                                              float* __restrict__ slot_xyz, int probe) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];   // [2][kGroup][n_slot * 3] captured atoms | [d_r * kGroup] feature staging | [kStream][4 * 68] reduction scratch | [n_rec_slot][6] record table
  __shared__ float bc[kGroup][CVF_AUX_ROWS + 2];
  __shared__ double cD[kGroup][3];
  __shared__ double sums[kGroup][16];
  __shared__ double part[2][kGroup][kStream][12];   // [buffer][frame][streaming wave][value]
  __shared__ double rpart[kStream][3];
  __shared__ int ctr[5];   // LDS counters the roles meet at (below)
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nc = pp.n_coord, nal = pp.n_align, nslot = pp.n_slot, N = nc / 3;
  const int nq = N >> 2, nqa = nal >> 2;
  const int cap_floats = kGroup * nslot * 3;
  float* featL = dyn + 2 * (size_t)cap_floats;
  // a workgroup takes QUADS of four consecutive frame groups (half a 64-frame tile: blockIdx.x, + gridDim.x, ...; see the flush below).  With
  // `rounds` >= 0 every workgroup takes exactly `rounds` quads and the groups behind them go round the workgroups ONE AT A TIME (100 000 frames
  // are 12.2 quads per workgroup: 13 rounds of quads would idle 6 % of the chip-time, 12 rounds + one single group 0.3 %).
  const int nq_mine = rounds >= 0 ? rounds : (int)((nquads - blockIdx.x + gridDim.x - 1) / gridDim.x);
  const int64_t single0 = 4 * (int64_t)gridDim.x * (rounds >= 0 ? rounds : 0) + blockIdx.x;   // this workgroup's first single group
  const int nx_mine = rounds >= 0 && single0 < groups ? (int)((groups - single0 + gridDim.x - 1) / gridDim.x) : 0;
  const int nit = 4 * nq_mine + nx_mine;   // its groups, numbered n = 0 .. nit - 1
  auto group_of = [&](int n) __attribute__((always_inline)) {
    return n < 4 * nq_mine ? ((int64_t)blockIdx.x + (int64_t)(n >> 2) * gridDim.x) * 4 + (n & 3) : single0 + (int64_t)(n - 4 * nq_mine) * gridDim.x;
  };
  // The roles meet at COUNTERS in LDS, not at the workgroup barrier (a barrier per group made every streaming wave wait for the slowest one
  // with one frame of loads in flight - stamped: waves 0..3 stream a group in 33 k cycles and then waited 16 k for waves 4..7):
  //   kDone  += 1 by a streaming wave that has captured its share of group n       -> the tail starts group n at 8 (n / 2 + 1); one counter per
  //             parity of n: a streaming wave may be a group ahead of the slowest one, and its count must not complete the older group's
  //   kFree  += 1 by a tail wave that is through with group n's capture buffer     -> streaming may overwrite it (group n + 2) at 4 (n + 1)
  //   kMet   += 1 by a tail wave whose features of group n are staged              -> waves 1..3 gather them at 4 (n + 1)
  //   kTaken += 1 by tail waves 1..3 when group n's staged features are in their registers -> the staging area is free at 3 (n + 1)
  // (LDS operations of a wave complete in order: data written before the increment is visible to whoever has seen the count.)
  enum { kDone = 0, kFree = 2, kMet = 3, kTaken = 4 };   // (kDone, kDone + 1: even and odd groups)
  if (tid < 5) ctr[tid] = 0;
  auto post = [&](int c) __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(&ctr[c], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto await = [&](int c, int target) __attribute__((always_inline)) {
    while (__hip_atomic_load(&ctr[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
  };
  // the feature records, once per launch into LDS: while the streaming waves keep the memory pipe full a global load takes ~6 k cycles
  // to come back, and the tail waves' record reads (one dependent round trip per 64 records) made THEM the slower role (stamped: features
  // 47 k cycles per group next to 32 k of streaming)
  const int nrs = pp.n_rec_slot > 0 ? pp.n_rec_slot : pp.n_rec;
  int32_t* recL = reinterpret_cast<int32_t*>(featL + (size_t)pp.d_r * kGroup + kStream * (4 * kRedPitchP));
  for (int i = tid; i < 6 * nrs; i += 64 * (kStream + kTail)) recL[i] = pp.rec_slot[i];
  __syncthreads();   // (the only barrier)
#ifdef CVF_STAMPS
  unsigned long long last_ = __builtin_amdgcn_s_memtime(), acc_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  typedef float nt4 __attribute__((ext_vector_type(4)));
  if (w < kStream) {
    // ================================================= streaming waves
    float* red = featL + (size_t)pp.d_r * kGroup + w * (4 * kRedPitchP);
    const float4* __restrict__ r4 = reinterpret_cast<const float4*>(pp.ref_c);
    const int4* __restrict__ s4 = reinterpret_cast<const int4*>(pp.atom_slot);
    float4 rp[NI], rq[NI], rr[NI];
    uint2 sl[NI];
    int vo[NI];        // byte offset of the lane's group of four atoms inside a frame (past the frame for lanes without one: such loads return 0 and fetch nothing)
    float mk[NI];
    float rs0 = 0.0f, rs1 = 0.0f, rs2 = 0.0f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int g = (i * kStream + w) * 64 + lane;
      const bool valid = g < nq, al = g < nqa;
      vo[i] = valid ? g * 48 : nc * 4;
      const int ga = al ? g : 0;
      const float4 z = {0.0f, 0.0f, 0.0f, 0.0f};
      const float4 p = r4[3 * ga], q = r4[3 * ga + 1], r = r4[3 * ga + 2];
      rp[i] = al ? p : z;
      rq[i] = al ? q : z;
      rr[i] = al ? r : z;
      mk[i] = al ? 1.0f : 0.0f;
      const int4 sv = s4[valid ? g : nq - 1];
      // 3 x slot = the atom's place in a frame's capture image, 0xffff: not a feature atom; two atoms to a register (the loop below is at
      // the register limit: unpacked, the twelve values were spilled - and a spill's reload waits for every load in flight)
      auto place = [&](int sv_) { return valid && sv_ >= 0 ? (unsigned)(3 * sv_) : 0xffffu; };
      sl[i] = uint2{place(sv.x) | (place(sv.y) << 16), place(sv.z) | (place(sv.w) << 16)};
      rs0 += rp[i].x + rp[i].w + rq[i].z + rr[i].y;
      rs1 += rp[i].y + rq[i].x + rq[i].w + rr[i].z;
      rs2 += rp[i].z + rq[i].y + rr[i].x + rr[i].w;
    }
    {
      const double t0 = wave_sum((double)rs0), t1 = wave_sum((double)rs1), t2 = wave_sum((double)rs2);
      if (lane == 0) { rpart[w][0] = t0; rpart[w][1] = t1; rpart[w][2] = t2; }
    }
    // frame number t of this workgroup (8 per group) -> a buffer descriptor of its coordinates (scalar registers; the lane's offset is one
    // 32-bit register per group of atoms); past the end: the last frame again (harmless, cached)
    auto frame_of = [&](int t) __attribute__((always_inline)) {
      int n = t >> 3;
      n = n < nit ? n : nit - 1;
      int64_t f = group_of(n) * kGroup + (t & 7);
      f = f < B ? f : B - 1;
      return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + f * nc), 0, nc * 4, 0x00020000);
    };
    auto ld16 = [&](__amdgpu_buffer_rsrc_t r, int voff, int imm) __attribute__((always_inline)) {
      return __builtin_bit_cast(nt4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, imm, 0));
    };
    auto request = [&](nt4 (&a)[NI], nt4 (&b)[NI], nt4 (&c)[NI], __amdgpu_buffer_rsrc_t r) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        a[i] = ld16(r, vo[i], 0);
        b[i] = ld16(r, vo[i], 16);
        c[i] = ld16(r, vo[i], 32);
      }
    };
    // one frame: its sums and captures
    auto step = [&](nt4 (&a)[NI], nt4 (&b)[NI], nt4 (&c)[NI], int capoff, double* pw) __attribute__((always_inline)) {
      capoff = __builtin_amdgcn_readfirstlane(capoff);
      asm volatile("" : "+s"(capoff));   // (keeps `image + slot` sums out of the loop preheader: 24 hoisted addresses spilled, and a spill's reload waits for every load in flight)
      float* capL = dyn + capoff;
      float s[12];
#pragma unroll
      for (int v = 0; v < 12; ++v) s[v] = 0.0f;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        auto atom4 = [&](float x0, float x1, float x2, float r0, float r1, float r2) {
          s[0] = fmaf(mk[i], x0, s[0]); s[1] = fmaf(mk[i], x1, s[1]); s[2] = fmaf(mk[i], x2, s[2]);
          s[3] = fmaf(x0, r0, s[3]); s[4] = fmaf(x0, r1, s[4]); s[5] = fmaf(x0, r2, s[5]);
          s[6] = fmaf(x1, r0, s[6]); s[7] = fmaf(x1, r1, s[7]); s[8] = fmaf(x1, r2, s[8]);
          s[9] = fmaf(x2, r0, s[9]); s[10] = fmaf(x2, r1, s[10]); s[11] = fmaf(x2, r2, s[11]);
        };
        atom4(a[i].x, a[i].y, a[i].z, rp[i].x, rp[i].y, rp[i].z);
        atom4(a[i].w, b[i].x, b[i].y, rp[i].w, rq[i].x, rq[i].y);
        atom4(b[i].z, b[i].w, c[i].x, rq[i].z, rq[i].w, rr[i].x);
        atom4(c[i].y, c[i].z, c[i].w, rr[i].y, rr[i].z, rr[i].w);
        {
          unsigned p0 = sl[i].x, p1 = sl[i].y;
          asm volatile("" : "+v"(p0), "+v"(p1));   // (unpacked at the use: not in the loop preheader)
          const unsigned sx = p0 & 0xffffu, sy = p0 >> 16, sz = p1 & 0xffffu, sw = p1 >> 16;
          if (sx != 0xffffu) { capL[sx] = a[i].x; capL[sx + 1] = a[i].y; capL[sx + 2] = a[i].z; }
          if (sy != 0xffffu) { capL[sy] = a[i].w; capL[sy + 1] = b[i].x; capL[sy + 2] = b[i].y; }
          if (sz != 0xffffu) { capL[sz] = b[i].z; capL[sz + 1] = b[i].w; capL[sz + 2] = c[i].x; }
          if (sw != 0xffffu) { capL[sw] = c[i].y; capL[sw + 1] = c[i].z; capL[sw + 2] = c[i].w; }
        }
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int v = 0; v < 4; ++v) red[v * kRedPitchP + lane] = s[4 * r + v];
        const float4 t = *reinterpret_cast<const float4*>(red + (lane >> 4) * kRedPitchP + 4 * (lane & 15));
        double d = ((double)t.x + (double)t.y) + ((double)t.z + (double)t.w);
        d += dpp_movd<0x111, 0xf>(d);
        d += dpp_movd<0x112, 0xf>(d);
        d += dpp_movd<0x114, 0xf>(d);
        d += dpp_movd<0x118, 0xf>(d);
        if ((lane & 15) == 15) pw[4 * r + (lane >> 4)] = d;
      }
    };
    nt4 a0[NI], b0[NI], c0[NI], a1[NI], b1[NI], c1[NI];
    request(a0, b0, c0, frame_of(0));
    __builtin_amdgcn_sched_barrier(0);
    // (the buffer of group n as integers that flip between 0 and the buffer's size: written as `n & 1` the compiler carried a lane mask through
    //  the loop and rebuilt the offsets from it in vector registers - which it then spilled.  Requesting a lane's group of atoms again the moment
    //  it is used up - instead of a whole frame at a time - was tried: the register allocator answers with a second copy of the buffer and a
    //  wait for every load at the loop header.)
    int capG = 0, partG = 0;
    double* part0 = &part[0][0][0][0];
#pragma unroll 1
    for (int n = 0; n < nit; ++n) {
      if (n >= 2) await(kFree, kTail * (n - 1));   // (group n - 2 used this capture buffer; the next frame's loads stay in flight meanwhile)
      PIPE_T(2);
#pragma unroll 1
      for (int j = 0; j < kGroup; j += 2) {
        request(a1, b1, c1, frame_of(8 * n + j + 1));
        __builtin_amdgcn_sched_barrier(0);
        step(a0, b0, c0, capG + j * nslot * 3, part0 + partG + (j * kStream + w) * 12);
        __builtin_amdgcn_sched_barrier(0);
        request(a0, b0, c0, frame_of(8 * n + j + 2));   // (j + 2 == 8: the next group's first frame)
        __builtin_amdgcn_sched_barrier(0);
        step(a1, b1, c1, capG + (j + 1) * nslot * 3, part0 + partG + ((j + 1) * kStream + w) * 12);
        __builtin_amdgcn_sched_barrier(0);
      }
      post(kDone + (n & 1));
      PIPE_T(1);
      capG = cap_floats - capG;
      partG = kGroup * kStream * 12 - partG;
    }
    PIPE_T_OUT();
    return;
  }
  // =================================================== tail waves
  // Tail wave 0 solves the group's eight 3x3 problems (one per lane) and evaluates the features that need the rotation (positions); tail waves
  // 1..3 evaluate the others (bonds, angles, dihedrals do not see the alignment), write the slot copy, and hold the quad's features in registers:
  //
  // THE FLUSH.  Written group by group the staged features are 32-byte pieces of the tiled rows (or 12 KB runs of the row-major output)
  // trickling into a pure read stream, and that costs far more than their 2.5 % of the bytes: 1222 us per 100 000 frames of 5000 atoms with
  // the pieces, 969 us without any store, 1005 us when the same stores stay in L2 - and 1050 us when the same bytes leave as whole 128-byte
  // lines once per four groups (developer probes of this kernel, one lease).  So the features of four consecutive groups = 32 frames = one
  // 128-byte line per feature row wait in REGISTERS of tail waves 1..3 (a lane: kHold 16-byte items; item (row o, piece p) is filled when group
  // p / 2 of the quad is staged) and are stored after the fourth group, eight lanes to a line.  (Held by all four tail waves next to the solve's
  // fp64 state they were spilled: 189 registers.)
  const int tw = w - kStream;
  const int staged = feat_tiled != nullptr ? 1 : 2;   // (the launch takes this kernel only with a staged output; 2: row-major alone)
  const int ns3 = nslot * 3;
  const int nb = (nrs + 63) >> 6;                     // batches of 64 records
  // one feature record of frame j of the group
  auto feature = [&](const float* capG, int64_t f0, int j, int r, bool positions) __attribute__((always_inline)) {
    const int32_t* p = recL + 6 * r;   // like rec, atom fields hold slots
    const Rec rc{p[0], p[1], p[2], p[3], p[4], p[5]};
    if (rc.type < 0 || (rc.type == CVF_FEAT_POSITION) != positions) return;   // (type < 0: padding entries of batched lists)
    const float* cp = capG + (size_t)j * ns3;
    auto sat = [&](int s_) { return V3{cp[3 * s_], cp[3 * s_ + 1], cp[3 * s_ + 2]}; };
    auto emit = [&](int o, float v) {
      if (staged == 1) featL[o * kGroup + j] = v;
      else featL[j * pp.d_r + o] = v;
      if (staged == 1 && feat_rows != nullptr && f0 + j < B) feat_rows[(f0 + j) * pp.d_r + o] = v;
    };
    if (rc.type == CVF_FEAT_POSITION) {
      const V3 xa = sat(rc.a0);
      const V3 xc = v3((float)((double)xa.x - cD[j][0]), (float)((double)xa.y - cD[j][1]), (float)((double)xa.z - cD[j][2]));
      float R[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) R[q] = bc[j][q];
      const V3 al = row_times(xc, R);
      emit(rc.out, al.x);
      emit(rc.out + 1, al.y);
      emit(rc.out + 2, al.z);
    } else if (rc.type == CVF_FEAT_BOND) {
      emit(rc.out, bond_eval(sat(rc.a0), sat(rc.a1)).val);
    } else if (rc.type == CVF_FEAT_ANGLE) {
      const float cs = angle_eval(sat(rc.a0), sat(rc.a1), sat(rc.a2)).cs;
      emit(rc.out, pp.use_angle_value ? acosf(cs) : cs);
    } else {
      const DihedralG dg = dihedral_eval(sat(rc.a0), sat(rc.a1), sat(rc.a2), sat(rc.a3));
      if (pp.use_angle_value) {
        emit(rc.out, atan2f(dg.sn, dg.cs));
      } else {
        emit(rc.out, dg.cs);
        emit(rc.out + 1, dg.sn);
      }
    }
  };
  if (tw == 0) {
    // ------------------------------------------------- tail wave 0: sums, solve, aux rows, position features
    constexpr int kAuxHold = CVF_AUX_ROWS * 32 / 64;   // 18 rows x 32 frames over the 64 lanes
    static_assert(CVF_AUX_ROWS * 32 % 64 == 0, "whole items per lane");
    float auxh[kAuxHold];
#pragma unroll
    for (int k = 0; k < kAuxHold; ++k) auxh[k] = 0.0f;
#pragma unroll 1
    for (int n = 0; n < nit; ++n) {
      PIPE_T(8);
      await(kDone + (n & 1), kStream * ((n >> 1) + 1));
      PIPE_T(1);
      if (probe & 1) {   // developer probe (CVF_K1_PIPE_PROBE=1): the streaming waves alone - wrong results, the time is the point
        post(kFree);
        continue;
      }
      const int64_t f0 = group_of(n) * kGroup;
      const float* capG = dyn + (size_t)(n & 1) * cap_floats;
      for (int i = lane; i < kGroup * 15; i += 64) {   // sums[frame][value] over the eight streaming waves, fixed order
        const int j = i / 15, v = i - 15 * j;
        double t = 0.0;
        if (v < 12) {
#pragma unroll
          for (int ww = 0; ww < kStream; ++ww) t += part[n & 1][j][ww][v];
        } else {
#pragma unroll
          for (int ww = 0; ww < kStream; ++ww) t += rpart[ww][v - 12];
        }
        sums[j][v] = t;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PIPE_T(2);
      if (lane < kGroup) {
        const int j = lane;
        const double* t = sums[j];
        const double inv = fast_rcp((double)nal);
        const double c0 = t[0] * inv, c1 = t[1] * inv, c2 = t[2] * inv;
        double H[3][3];
        H[0][0] = t[3] - c0 * t[12]; H[0][1] = t[4] - c0 * t[13]; H[0][2] = t[5] - c0 * t[14];
        H[1][0] = t[6] - c1 * t[12]; H[1][1] = t[7] - c1 * t[13]; H[1][2] = t[8] - c1 * t[14];
        H[2][0] = t[9] - c2 * t[12]; H[2][1] = t[10] - c2 * t[13]; H[2][2] = t[11] - c2 * t[14];
        KabschOut ko;
        if (aux_tiled != nullptr) {
          kabsch_from_H<true>(H, ko);
        } else {
          kabsch_from_H<false>(H, ko);
#pragma unroll
          for (int i = 0; i < 6; ++i) ko.Kinv[i] = 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) bc[j][i] = ko.R[i];
        bc[j][9] = (float)c0; bc[j][10] = (float)c1; bc[j][11] = (float)c2;
#pragma unroll
        for (int i = 0; i < 6; ++i) bc[j][12 + i] = ko.Kinv[i];
        cD[j][0] = c0; cD[j][1] = c1; cD[j][2] = c2;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PIPE_T(3);
      if (aux_tiled != nullptr) {
        if (n < 4 * nq_mine) {   // like the features (below): a quad's rotation rows leave as whole 128-byte lines, 32 frames of a row
          const int gq = n & 3;
#pragma unroll
          for (int k = 0; k < kAuxHold; ++k) {
            const int i = lane + 64 * k, row = i >> 5, fr = i & 31;
            if ((fr >> 3) == gq) auxh[k] = bc[fr & 7][row];
          }
          if (gq == 3) {
            const int64_t f0q = f0 - 3 * kGroup;
            float* a0p = aux_tiled + (f0q / CVF_TILE) * CVF_AUX_ROWS * CVF_TILE + (int)(f0q % CVF_TILE);
#pragma unroll
            for (int k = 0; k < kAuxHold; ++k) {
              int i = lane + 64 * k;
              asm volatile("" : "+v"(i));
              a0p[(i >> 5) * CVF_TILE + (i & 31)] = auxh[k];
            }
          }
        } else {
          const int64_t tile = f0 / CVF_TILE;
          const int l0 = (int)(f0 % CVF_TILE);
          for (int i = lane; i < kGroup * CVF_AUX_ROWS; i += 64) {
            const int row = i / kGroup, j = i - row * kGroup;
            aux_tiled[(tile * CVF_AUX_ROWS + row) * CVF_TILE + l0 + j] = bc[j][row];
          }
        }
      }
      if (n >= 1) await(kTaken, (kTail - 1) * n);   // the staging area still held group n - 1
      for (int m = 0; m < kGroup * nb; ++m) {
        const int j = m / nb, r = (m - j * nb) * 64 + lane;
        if (r < nrs) feature(capG, f0, j, r, true);
      }
      PIPE_T(4);
      post(kFree);
      post(kMet);
    }
    PIPE_T_OUT();
    return;
  }
  // --------------------------------------------------- tail waves 1..3
  constexpr int kHold = 16;   // x 192 lanes x 16 bytes = 32 frames x 384 features (the launch checks d_r)
  float4 hold[kHold];
  const int hl = (tw - 1) * 64 + lane;   // this lane among the 192 that hold
#pragma unroll
  for (int k = 0; k < kHold; ++k) hold[k] = float4{0.0f, 0.0f, 0.0f, 0.0f};
  const int d8 = pp.d_r * 8;   // items of a quad: tiled - (row o, 16-byte piece p8 of its 128-byte line); row-major - float4 i of its 4 x 8 d_r floats
#pragma unroll 1
  for (int n = 0; n < nit; ++n) {
    PIPE_T(8);
    await(kDone + (n & 1), kStream * ((n >> 1) + 1));
    PIPE_T(1);
    if (probe & 1) {
      post(kFree);
      continue;
    }
    const int gq = n & 3;
    const int64_t f0 = group_of(n) * kGroup;
    const float* capG = dyn + (size_t)(n & 1) * cap_floats;
    if (slot_xyz != nullptr) {   // the compact copy of the feature atoms for the derivative kernel: [frame group][n_slot * 3][kGroup]
      nt4* dst = reinterpret_cast<nt4*>(slot_xyz + (f0 / kGroup) * (int64_t)ns3 * kGroup);
      for (int i = hl; i < 2 * ns3; i += 64 * (kTail - 1)) {
        const int row = i >> 1, fr = 4 * (i & 1);
        const nt4 v = {capG[fr * ns3 + row], capG[(fr + 1) * ns3 + row], capG[(fr + 2) * ns3 + row], capG[(fr + 3) * ns3 + row]};
        __builtin_nontemporal_store(v, dst + i);   // (plain stores: the same time; the copy is 11 % of the launch's bytes and 16 % of its time)
      }
    }
    PIPE_T(2);
    if (n >= 1) await(kTaken, (kTail - 1) * n);
    for (int m = tw - 1; m < kGroup * nb; m += kTail - 1) {   // (frame, batch of 64 records) pairs dealt to the three waves
      const int j = m / nb, r = (m - j * nb) * 64 + lane;
      if (r < nrs) feature(capG, f0, j, r, false);
    }
    PIPE_T(3);
    post(kFree);
    post(kMet);
    await(kMet, kTail * (n + 1));
    PIPE_T(5);
    if (probe & 8) {   // developer probe: no feature stores
      post(kTaken);
      continue;
    }
    if (n >= 4 * nq_mine) {   // a single group behind the quads: its staged features leave at once (32-byte pieces of the tiled rows / one 8 d_r run)
      if (staged == 1) {
        float* row0 = feat_tiled + (f0 / CVF_TILE) * pp.d_r * CVF_TILE + (int)(f0 % CVF_TILE);
        for (int idx = hl; idx < pp.d_r * 2; idx += 192)
          *reinterpret_cast<float4*>(row0 + (idx >> 1) * CVF_TILE + 4 * (idx & 1)) = *reinterpret_cast<const float4*>(featL + (idx >> 1) * kGroup + 4 * (idx & 1));
      } else {
        const int64_t base = f0 * pp.d_r, lim = B * pp.d_r;
        for (int idx = hl; idx < 2 * pp.d_r; idx += 192) {
          const int64_t e = base + 4 * (int64_t)idx;
          const float4 v = *reinterpret_cast<const float4*>(featL + 4 * idx);
          if (e + 3 < lim) {
            *reinterpret_cast<float4*>(feat_rows + e) = v;
          } else if (e < lim) {
            feat_rows[e] = v.x;
            if (e + 1 < lim) feat_rows[e + 1] = v.y;
            if (e + 2 < lim) feat_rows[e + 2] = v.z;
          }
        }
      }
      post(kTaken);
      PIPE_T(6);
      continue;
    }
#pragma unroll
    for (int k = 0; k < kHold; ++k) {
      const int idx = hl + 192 * k;
      const int gi = staged == 1 ? (idx & 7) >> 1 : idx / (2 * pp.d_r);   // the group of the quad that fills this item
      if (idx < d8 && gi == gq) {
        const int src = staged == 1 ? (idx & ~7) + 4 * (idx & 1) : 4 * (idx - gq * 2 * pp.d_r);
        hold[k] = *reinterpret_cast<const float4*>(featL + src);
      }
    }
    post(kTaken);
    if (gq == 3) {
      const int64_t f0q = f0 - 3 * kGroup;   // first frame of the quad: a multiple of 32
      if (staged == 1) {
        float* row0 = feat_tiled + (f0q / CVF_TILE) * pp.d_r * CVF_TILE + (int)(f0q % CVF_TILE);
#pragma unroll
        for (int k = 0; k < kHold; ++k) {
          int idx = hl + 192 * k;
          asm volatile("" : "+v"(idx));   // (the address is formed here: hoisted out of the group loop the sixteen 64-bit addresses were spilled)
          // (cache-policy bits on these stores - nt, sc1, sc0 sc1 - change nothing: 1042 / 1043 / 1047 / 1043 us; neither does putting every
          //  workgroup's flush on a chip-wide time grid (s_memrealtime), with or without the bits: 1037 / 1040 / 1047 us for no grid / 10 / 20 us)
          if (idx < d8) *reinterpret_cast<float4*>(row0 + (idx >> 3) * CVF_TILE + 4 * (idx & 7)) = hold[k];
        }
      } else {
        const int64_t base = f0q * pp.d_r, lim = B * pp.d_r;   // (the launch checked that feat_rows is 16-byte aligned)
#pragma unroll
        for (int k = 0; k < kHold; ++k) {
          int idx = hl + 192 * k;
          asm volatile("" : "+v"(idx));
          const int64_t e = base + 4 * (int64_t)idx;
          if (idx < d8 && e + 3 < lim) {
            *reinterpret_cast<float4*>(feat_rows + e) = hold[k];
          } else if (idx < d8 && e < lim) {   // the output's last, partial float4
            feat_rows[e] = hold[k].x;
            if (e + 1 < lim) feat_rows[e + 1] = hold[k].y;
            if (e + 2 < lim) feat_rows[e + 2] = hold[k].z;
          }
        }
      }
    }
    PIPE_T(6);
  }
  PIPE_T_OUT();
}

}  // namespace

static bool capture_ok(const cvf_pp_desc* pp, bool tiled) {
  if (!(pp->atom_slot && pp->rec_slot && pp->atom_align && pp->n_slot > 0)) return false;
  const size_t ldsc = ((size_t)kGroup * pp->n_slot * 3 + (tiled ? (size_t)pp->d_r * kGroup : 0)) * sizeof(float);
  return ldsc <= 150 * 1024;
}
// bytes of the compact feature-atom copy [frame groups of 8][n_slot * 3][8] written when `scratch` is given
size_t cvf_k1_large_scratch_bytes(const cvf_pp_desc* pp, int64_t B) {
  return capture_ok(pp, true) ? (size_t)cvf_ntiles(B) * CVF_TILE * pp->n_slot * 3 * sizeof(float) : 0;
}

// called from cvf_align_feature_fwd (k1_align.hip) when the frame does not fit the lane-per-frame tile
int cvf_k1_large_launch(const cvf_pp_desc* pp, const float* x, int64_t B, float* feat_tiled, float* feat_rows,
                        float* aux_tiled, float* slot_xyz, hipStream_t s) {
  const bool contig = (pp->flags & CVF_PP_ALIGN_CONTIG) != 0;
  // CVF_K1_XCD=1: the frame groups of a 64-frame tile on ONE XCD (their 32-byte pieces of the tiled rows meet in one L2); 0 / unset:
  // blockIdx order (developer switch: bench.py's A/B of the two)
  const int xcd_env = getenv("CVF_K1_XCD") ? atoi(getenv("CVF_K1_XCD")) : -1;
  // (round 4, A/B in one lease, both flavours: blockIdx order 1160 / 1351 us per 100 000 frames of 5000 atoms against 1240 / 1371 us
  //  with the tile on one XCD on one box, equal on another - the slot copy has been one contiguous run per workgroup since round 3,
  //  which is what the placement was for.  Default: blockIdx order.)
  const int same_xcd = xcd_env >= 0 ? xcd_env : 0;
  if (same_xcd & 6) {
    static bool told = false;
    if (!told) fprintf(stderr, "[cvf] CVF_K1_XCD=%d: developer timing probe - the outputs of cvf_align_feature_fwd are WRONG\n", same_xcd);
    told = true;
  }
  const int64_t groups = feat_tiled || aux_tiled || slot_xyz ? cvf_ntiles(B) * (CVF_TILE / kGroup) : (B + kGroup - 1) / kGroup;
  const bool staged = feat_tiled != nullptr || feat_rows != nullptr;   // features leave through an LDS staging area
  if (capture_ok(pp, staged)) {
    const size_t ldsc = ((size_t)kGroup * pp->n_slot * 3 + (staged ? (size_t)pp->d_r * kGroup : 0)) * sizeof(float);
    {
      const int N = pp->n_coord / 3;
      const bool vec4 = contig && (N % 4 == 0) && (pp->n_align % 4 == 0) && (((uintptr_t)x & 15) == 0) &&
                        (((uintptr_t)pp->ref_c & 15) == 0) && (((uintptr_t)pp->atom_slot & 15) == 0);
      if (ldsc > 48 * 1024) {
        (void)hipFuncSetAttribute((const void*)k1_large_capture_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsc);
        (void)hipFuncSetAttribute((const void*)k1_large_capture_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsc);
      }
      const int ni = (N / 4 + 64 * kGroup - 1) / (64 * kGroup);
      // (the slice kernel stages its reductions in the feature staging area: needs the tiled output's buffer)
      const bool slice = vec4 && ni <= 4 && staged && (size_t)pp->d_r * kGroup >= (size_t)kGroup * 4 * 68 &&
                         getenv("CVF_K1_NOSLICE") == nullptr;
      // large batches: the pipelined kernel (one resident workgroup per CU; streaming waves + tail waves).  It needs two capture
      // buffers in LDS and enough groups per workgroup for the overlap to matter; CVF_K1_NOPIPE=1: developer switch (A/B).
      const int ncu = cvf_cu_count();
      const size_t lds_pipe = (2 * (size_t)kGroup * pp->n_slot * 3 + (size_t)pp->d_r * kGroup + (size_t)kStream * 4 * kRedPitchP +
                               6 * (size_t)(pp->n_rec_slot > 0 ? pp->n_rec_slot : pp->n_rec)) * sizeof(float);   // (+ 14.4 KB of static arrays: 160 KB in all)
      // Where it pays against the one-group-per-workgroup kernel at two workgroups per CU (tools/k1_ab2.sh, both in one lease, two kinds of box,
      // us per launch pipelined / other): tiled features only - 16 000 frames 194 / 192, 24 000: 279 / 284, 32 768: 372 / 383, 100 000: 1065-1118 /
      // 1078-1153; with the generator-mode outputs (slot copy: 11 % of the bytes, written whole by either kernel) - 16 000: 225 / 218, 50 000: 675 / 667,
      // 100 000: 1254-1336 / 1240-1343 (by box: from 1.3 % faster to 7 % slower); row-major output alone (one 12-KB run per group either way):
      // 1067-1123 / 1032-1110.  So: tiled features alone from 24 576 frames on; never with the generator-mode outputs or for the row-major output.
      // (developer switches: CVF_K1_PIPE_MIN_GROUPS - with single groups only below four groups per compute unit: 2000 frames 41 us / 34, 8000: 120 / 113;
      //  CVF_K1_NOPIPE)
      const int64_t pipe_min = getenv("CVF_K1_PIPE_MIN_GROUPS") ? atoll(getenv("CVF_K1_PIPE_MIN_GROUPS")) : -1;
      const int64_t pipe_from = pipe_min >= 0 ? pipe_min : (feat_tiled == nullptr || aux_tiled || slot_xyz ? INT64_MAX : 12 * (int64_t)ncu);
      if (vec4 && ni <= 3 && staged && groups >= pipe_from && lds_pipe <= 145 * 1024 && 3 * pp->n_slot < 0xffff && pp->d_r <= 384 &&
          (feat_tiled != nullptr || ((uintptr_t)feat_rows & 15) == 0) && getenv("CVF_K1_NOPIPE") == nullptr) {
        const int64_t nquads = (groups + 3) / 4;   // (tiled outputs: groups is a multiple of 8)
        // quads only, or whole rounds of quads and the groups behind them one at a time - whichever gives the busiest workgroup fewer groups;
        // fewer than four groups per compute unit: single groups only
        int64_t grid = nquads < ncu ? nquads : ncu;
        const int64_t r_all = (nquads + grid - 1) / grid, r_full = nquads / grid;
        const int64_t left = groups - 4 * grid * r_full, with_singles = 4 * r_full + (left + grid - 1) / grid;
        int rounds = r_full >= 1 && with_singles < 4 * r_all ? (int)r_full : -1;
        if (groups < 4 * (int64_t)ncu) {
          grid = groups < ncu ? groups : ncu;
          rounds = 0;
        }
        auto go = [&](auto kernel) {
          (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pipe);
          const int probe = getenv("CVF_K1_PIPE_PROBE") ? atoi(getenv("CVF_K1_PIPE_PROBE")) : 0;
          if (probe != 0) {
            static bool told = false;
            if (!told) fprintf(stderr, "[cvf] CVF_K1_PIPE_PROBE=%d: developer timing probe - the outputs of cvf_align_feature_fwd are WRONG\n", probe);
            told = true;
          }
          hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(64 * (kStream + kTail)), lds_pipe, s, *pp, x, B, nquads, groups, rounds,
                             feat_tiled, feat_rows, aux_tiled, slot_xyz, probe);
        };
        if (ni == 1) go(k1_large_pipe_kernel<1>);
        else if (ni == 2) go(k1_large_pipe_kernel<2>);
        else go(k1_large_pipe_kernel<3>);
        return cvf_check_launch("k1_large_pipe_kernel");
      }
      if (slice) {
        auto go = [&](auto kernel) {
          if (ldsc > 48 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsc);
          hipLaunchKernelGGL(kernel, dim3((unsigned)groups), dim3(64 * kGroup), ldsc, s, *pp, x, B, feat_tiled, feat_rows, aux_tiled,
                             slot_xyz, same_xcd);
        };
        // non-temporal coordinate loads, a developer switch: measured 1592 us against 1065 us (0.48 vs 0.72 of 8 TB/s) for
        // 100 000 frames of 5000 atoms (tools/bench_k1_c5.py) - the plain loads stay the default
        static const bool nt = getenv("CVF_K1_NT") != nullptr && atoi(getenv("CVF_K1_NT")) != 0;
        if (nt) {
          if (ni == 1) go(k1_large_slice_kernel<1, true>);
          else if (ni == 2) go(k1_large_slice_kernel<2, true>);
          else if (ni == 3) go(k1_large_slice_kernel<3, true>);
          else go(k1_large_slice_kernel<4, true>);
        } else {
          if (ni == 1) go(k1_large_slice_kernel<1, false>);
          else if (ni == 2) go(k1_large_slice_kernel<2, false>);
          else if (ni == 3) go(k1_large_slice_kernel<3, false>);
          else go(k1_large_slice_kernel<4, false>);
        }
        return cvf_check_launch("k1_large_slice_kernel");
      }
      if (vec4)
        hipLaunchKernelGGL(k1_large_capture_kernel<true>, dim3((unsigned)groups), dim3(64 * kGroup), ldsc, s, *pp, x, B,
                           feat_tiled, feat_rows, aux_tiled, slot_xyz, same_xcd);
      else
        hipLaunchKernelGGL(k1_large_capture_kernel<false>, dim3((unsigned)groups), dim3(64 * kGroup), ldsc, s, *pp, x, B,
                           feat_tiled, feat_rows, aux_tiled, slot_xyz, same_xcd);
      return cvf_check_launch("k1_large_capture_kernel");
    }
  }
  const size_t lds = feat_tiled ? (size_t)pp->d_r * kGroup * sizeof(float) : 0;
  CVF_REQUIRE(lds <= 96 * 1024, "cvf_align_feature_fwd: %d features do not fit the staging buffer", pp->d_r);
  if (lds > 48 * 1024) {
    (void)hipFuncSetAttribute((const void*)k1_large_gather_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)k1_large_gather_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  if (contig)
    hipLaunchKernelGGL(k1_large_gather_kernel<true>, dim3((unsigned)groups), dim3(64 * kGroup), lds, s, *pp, x, B, feat_tiled, feat_rows,
                       aux_tiled);
  else
    hipLaunchKernelGGL(k1_large_gather_kernel<false>, dim3((unsigned)groups), dim3(64 * kGroup), lds, s, *pp, x, B, feat_tiled, feat_rows,
                       aux_tiled);
  return cvf_check_launch("k1_large_gather_kernel");
}


// ---------------------------------------------------------------------------------------------------------------------------
// Measurement aid (bench.py roofline_align_feature): what plain streaming reaches on THIS box at THIS moment, timed in the same
// loop as the alignment kernel so that box-to-box and minute-to-minute spread can be told from a change of the kernel.
//   mode 0: dst[i] = src[i], one float4 per thread (1:1 read/write traffic: the dipeptide shape's mix)
//   mode 1: read-only sweep, a workgroup adds a contiguous 32 KB run (eight 16-byte loads per thread in flight), one float written
//           (the large-molecule shape: 60 KB read, 1.5 KB written per frame)
namespace {
__global__ __launch_bounds__(256) void probe_copy_kernel(float4* __restrict__ dst, const float4* __restrict__ src, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void probe_read_kernel(float* __restrict__ dst, const float4* __restrict__ src, int64_t n4) {
  const int64_t base = (int64_t)blockIdx.x * 2048 + threadIdx.x;   // the block's 32 KB run, 4 KB (one piece per lane) per load
  float4 v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = src[base + 256 * j < n4 ? base + 256 * j : n4 - 1];
  float acc = 0.0f;
#pragma unroll
  for (int j = 0; j < 8; ++j) acc += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  acc = wave_sumf(acc);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) dst[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}
}  // namespace

extern "C" int64_t cvf_probe_stream_out_floats(int mode, int64_t n_float4) {
  return mode == 0 ? 4 * n_float4 : (n_float4 + 2047) / 2048;
}
extern "C" int cvf_probe_stream(int mode, float* dst, const float* src, int64_t n_float4, void* stream) {
  CVF_REQUIRE(dst && src && n_float4 > 0 && (mode == 0 || mode == 1), "cvf_probe_stream: bad argument");
  CVF_REQUIRE((((uintptr_t)dst | (uintptr_t)src) & 15) == 0, "cvf_probe_stream: buffers must be 16-byte aligned");
  if (mode == 0)
    hipLaunchKernelGGL(probe_copy_kernel, dim3((unsigned)((n_float4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<float4*>(dst), reinterpret_cast<const float4*>(src), n_float4);
  else
    hipLaunchKernelGGL(probe_read_kernel, dim3((unsigned)((n_float4 + 2047) / 2048)), dim3(256), 0, (hipStream_t)stream, dst,
                       reinterpret_cast<const float4*>(src), n_float4);
  return cvf_check_launch("probe_stream_kernel");
}
