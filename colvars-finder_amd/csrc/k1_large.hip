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

namespace {

constexpr int kGroup = 8;  // waves = frames per workgroup

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
    kabsch_from_H(H, ko);
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
  float* fr = (feat_rows && real) ? feat_rows + frame * pp.d_r : nullptr;
  auto emit = [&](int o, float v) {
    if (feat_tiled) featL[o * kGroup + fi] = v;
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

template <bool VEC4>
__global__ __launch_bounds__(64 * kGroup) void k1_large_capture_kernel(cvf_pp_desc pp, const float* __restrict__ x, int64_t B,
                                                                       float* __restrict__ feat_tiled,
                                                                       float* __restrict__ feat_rows,
                                                                       float* __restrict__ aux_tiled,
                                                                       float* __restrict__ slot_xyz) {
  extern __shared__ float dyn[];                  // [kGroup][n_slot][3] captured atoms, then [d_r][kGroup] features
  __shared__ float bc[kGroup][CVF_AUX_ROWS + 2];
  __shared__ double cD[kGroup][3];
  __shared__ double sums[kGroup][16];
  const int tid = threadIdx.x, lane = tid & 63, fi = tid >> 6;
  const int nc = pp.n_coord, nal = pp.n_align, nslot = pp.n_slot, N = nc / 3;
  float* capL = dyn + (size_t)fi * nslot * 3;
  float* featL = dyn + (size_t)kGroup * nslot * 3;
  const int64_t f0 = (int64_t)blockIdx.x * kGroup;
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
        const float4 p = r4[3 * g], q = r4[3 * g + 1], r = r4[3 * g + 2];
        acc_atom(A, a.x, a.y, a.z, p.x, p.y, p.z);
        acc_atom(A, a.w, b.x, b.y, p.w, q.x, q.y);
        acc_atom(A, b.z, b.w, c.x, q.z, q.w, r.x);
        acc_atom(A, c.y, c.z, c.w, r.y, r.z, r.w);
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
  if (tid < kGroup) {  // the group's 3x3 problems, one per lane of one wave
    const double* t = sums[tid];
    const double inv = fast_rcp((double)nal);
    const double c0 = t[0] * inv, c1 = t[1] * inv, c2 = t[2] * inv;
    double H[3][3];
    H[0][0] = t[3] - c0 * t[12]; H[0][1] = t[4] - c0 * t[13]; H[0][2] = t[5] - c0 * t[14];
    H[1][0] = t[6] - c1 * t[12]; H[1][1] = t[7] - c1 * t[13]; H[1][2] = t[8] - c1 * t[14];
    H[2][0] = t[9] - c2 * t[12]; H[2][1] = t[10] - c2 * t[13]; H[2][2] = t[11] - c2 * t[14];
    KabschOut ko;
    kabsch_from_H(H, ko);
#pragma unroll
    for (int i = 0; i < 9; ++i) bc[tid][i] = ko.R[i];
    bc[tid][9] = (float)c0; bc[tid][10] = (float)c1; bc[tid][11] = (float)c2;
#pragma unroll
    for (int i = 0; i < 6; ++i) bc[tid][12 + i] = ko.Kinv[i];
    cD[tid][0] = c0; cD[tid][1] = c1; cD[tid][2] = c2;
  }
  __syncthreads();
  const int64_t tile = f0 / CVF_TILE;
  const int l0 = (int)(f0 % CVF_TILE);
  if (aux_tiled != nullptr && lane < CVF_AUX_ROWS) aux_tiled[(tile * CVF_AUX_ROWS + lane) * CVF_TILE + l0 + fi] = bc[fi][lane];
  if (slot_xyz != nullptr) {  // compact copy of the feature atoms for the derivative kernels (padded frame index)
    float* dst = slot_xyz + (f0 + fi) * (int64_t)nslot * 3;
    for (int i = lane; i < nslot * 3; i += 64) dst[i] = capL[i];
  }
  float R[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = bc[fi][i];
  const double cc0 = cD[fi][0], cc1 = cD[fi][1], cc2 = cD[fi][2];
  float* fr = (feat_rows && real) ? feat_rows + frame * pp.d_r : nullptr;
  auto emit = [&](int o, float v) {
    if (feat_tiled) featL[o * kGroup + fi] = v;
    if (fr) fr[o] = v;
  };
  auto satom = [&](int sl) { return V3{capL[3 * sl], capL[3 * sl + 1], capL[3 * sl + 2]}; };
  for (int r = lane; r < pp.n_rec; r += 64) {
    const int32_t* p = pp.rec_slot + 6 * r;   // like rec, atom fields hold slots
    const Rec rc{p[0], p[1], p[2], p[3], p[4], p[5]};
    if (rc.type == CVF_FEAT_POSITION) {
      const V3 xa = satom(rc.a0);
      const V3 xc = v3((float)((double)xa.x - cc0), (float)((double)xa.y - cc1), (float)((double)xa.z - cc2));
      const V3 al = row_times(xc, R);
      emit(rc.out, al.x);
      emit(rc.out + 1, al.y);
      emit(rc.out + 2, al.z);
    } else if (rc.type == CVF_FEAT_BOND) {
      emit(rc.out, bond_eval(satom(rc.a0), satom(rc.a1)).val);
    } else if (rc.type == CVF_FEAT_ANGLE) {
      const float cs = angle_eval(satom(rc.a0), satom(rc.a1), satom(rc.a2)).cs;
      emit(rc.out, pp.use_angle_value ? acosf(cs) : cs);
    } else {
      const DihedralG dg = dihedral_eval(satom(rc.a0), satom(rc.a1), satom(rc.a2), satom(rc.a3));
      if (pp.use_angle_value) {
        emit(rc.out, atan2f(dg.sn, dg.cs));
      } else {
        emit(rc.out, dg.cs);
        emit(rc.out + 1, dg.sn);
      }
    }
  }
  if (feat_tiled != nullptr) {
    __syncthreads();
    for (int idx = tid; idx < pp.d_r * kGroup; idx += 64 * kGroup) {
      const int o = idx / kGroup, f = idx % kGroup;
      feat_tiled[(tile * pp.d_r + o) * CVF_TILE + l0 + f] = featL[idx];
    }
  }
}

}  // namespace

static bool capture_ok(const cvf_pp_desc* pp, bool tiled) {
  if (!(pp->atom_slot && pp->rec_slot && pp->atom_align && pp->n_slot > 0)) return false;
  const size_t ldsc = ((size_t)kGroup * pp->n_slot * 3 + (tiled ? (size_t)pp->d_r * kGroup : 0)) * sizeof(float);
  return ldsc <= 150 * 1024;
}
// bytes of the compact feature-atom copy [padded frames][n_slot][3] written when `scratch` is given
size_t cvf_k1_large_scratch_bytes(const cvf_pp_desc* pp, int64_t B) {
  return capture_ok(pp, true) ? (size_t)cvf_ntiles(B) * CVF_TILE * pp->n_slot * 3 * sizeof(float) : 0;
}

// called from cvf_align_feature_fwd (k1_align.hip) when the frame does not fit the lane-per-frame tile
int cvf_k1_large_launch(const cvf_pp_desc* pp, const float* x, int64_t B, float* feat_tiled, float* feat_rows,
                        float* aux_tiled, float* slot_xyz, hipStream_t s) {
  const bool contig = (pp->flags & CVF_PP_ALIGN_CONTIG) != 0;
  const int64_t groups = feat_tiled || aux_tiled || slot_xyz ? cvf_ntiles(B) * (CVF_TILE / kGroup) : (B + kGroup - 1) / kGroup;
  if (capture_ok(pp, feat_tiled != nullptr)) {
    const size_t ldsc = ((size_t)kGroup * pp->n_slot * 3 + (feat_tiled ? (size_t)pp->d_r * kGroup : 0)) * sizeof(float);
    {
      const int N = pp->n_coord / 3;
      const bool vec4 = contig && (N % 4 == 0) && (pp->n_align % 4 == 0) && (((uintptr_t)x & 15) == 0) &&
                        (((uintptr_t)pp->ref_c & 15) == 0) && (((uintptr_t)pp->atom_slot & 15) == 0);
      if (ldsc > 48 * 1024) {
        (void)hipFuncSetAttribute((const void*)k1_large_capture_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsc);
        (void)hipFuncSetAttribute((const void*)k1_large_capture_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsc);
      }
      if (vec4)
        hipLaunchKernelGGL(k1_large_capture_kernel<true>, dim3((unsigned)groups), dim3(64 * kGroup), ldsc, s, *pp, x, B,
                           feat_tiled, feat_rows, aux_tiled, slot_xyz);
      else
        hipLaunchKernelGGL(k1_large_capture_kernel<false>, dim3((unsigned)groups), dim3(64 * kGroup), ldsc, s, *pp, x, B,
                           feat_tiled, feat_rows, aux_tiled, slot_xyz);
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
