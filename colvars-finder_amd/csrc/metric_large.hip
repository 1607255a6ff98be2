// K2+K3 for large molecules: q = J A J^T g and E = g^T J A J^T g per (frame, net) without ever touching the
// N-atom frame again.
//
// J^T g is sparse - only the atoms some feature reads ("slots", captured by K1 into `slot_xyz`) receive a direct
// contribution s_t - plus, when position features exist, the rigid-body correction of the alignment on EVERY
// align atom:  G_b = s_b + d_b,  d_b = Z ref_b - shift  (Z = R [Kinv ax(R^T M)]x, shift = sum_p / n_align).
// Everything the loss needs from the dense part is a moment of (a, ref) over the align atoms and is precomputed
// once (cvf_metric_dense_tensors):
//     T0[c] = sum_b a_bc      T1[c][j] = sum_b a_bc ref_bj      T2[c][j][k] = sum_b a_bc ref_bj ref_bk      R1[j] = sum_b ref_bj
//     E_dense   = sum_c ( Z_c. T2[c] Z_c.^T - 2 shift_c Z_c. T1[c] + shift_c^2 T0[c] )
//     usum_dense[c] = Z_c. T1[c] - shift_c T0[c]          dH_dense[c][j] = Z_c. T2[c][.][j] - shift_c T1[c][j]
// so a frame costs O(n_slot + n_rec) work instead of O(N).
// One wave per frame, 8 frames per workgroup; per net the g / q columns of the 8 frames move through LDS so that
// the tiled tensors are read and written in 32-byte segments.
#include "cvf_kabsch.hpp"

namespace {

constexpr int kGroup = 8;

struct Rec {
  int type, a0, a1, a2, a3, out;
};

__device__ __forceinline__ V3 wave_sum3(V3 v) { return V3{wave_sumf(v.x), wave_sumf(v.y), wave_sumf(v.z)}; }

__global__ __launch_bounds__(64 * kGroup) void metric_large_kernel(cvf_pp_desc pp, int64_t B, const float* __restrict__ aux_tiled,
                                                                   const float* __restrict__ a, int k,
                                                                   const float* __restrict__ slot_xyz,
                                                                   const double* __restrict__ dense,
                                                                   const float* __restrict__ g_tiled,
                                                                   float* __restrict__ q_tiled, float* __restrict__ e_tiled) {
  extern __shared__ float dyn[];
  const int tid = threadIdx.x, lane = tid & 63, fi = tid >> 6;
  const int ns = pp.n_slot, d_r = pp.d_r, nal = pp.n_align;
  float* gL = dyn;                                  // [d_r][kGroup]
  float* qL = gL + d_r * kGroup;                    // [d_r][kGroup]
  float* xsL = qL + d_r * kGroup + (size_t)fi * ns * 3;          // this wave's slot coordinates
  float* GsL = qL + d_r * kGroup + (size_t)kGroup * ns * 3 + (size_t)fi * ns * 3;  // this wave's slot accumulators
  const int64_t f0 = (int64_t)blockIdx.x * kGroup;
  const int64_t tile = f0 / CVF_TILE;
  const int l0 = (int)(f0 % CVF_TILE);
  const int64_t fpad = f0 + fi;                     // index into the padded (tile-complete) frame range
  // per-frame constants
  const float* ax = aux_tiled + tile * CVF_AUX_ROWS * CVF_TILE + l0 + fi;
  float R[9], Kinv[6], c[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = ax[i * CVF_TILE];
#pragma unroll
  for (int i = 0; i < 3; ++i) c[i] = ax[(9 + i) * CVF_TILE];
#pragma unroll
  for (int i = 0; i < 6; ++i) Kinv[i] = ax[(12 + i) * CVF_TILE];
  const float* xs = slot_xyz + fpad * (int64_t)ns * 3;
  for (int i = lane; i < ns * 3; i += 64) xsL[i] = xs[i];
  // dense moments (wave-uniform scalars)
  float T0[3], T1[9], T2[27], R1[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) T0[i] = (float)dense[i];
#pragma unroll
  for (int i = 0; i < 9; ++i) T1[i] = (float)dense[3 + i];
#pragma unroll
  for (int i = 0; i < 27; ++i) T2[i] = (float)dense[12 + i];
#pragma unroll
  for (int i = 0; i < 3; ++i) R1[i] = (float)dense[39 + i];
  const float inv_nal = 1.0f / (float)nal;
  auto xat = [&](int sl) { return V3{xsL[3 * sl], xsL[3 * sl + 1], xsL[3 * sl + 2]}; };
  auto uat = [&](int sl) { return V3{GsL[3 * sl], GsL[3 * sl + 1], GsL[3 * sl + 2]}; };
  auto addG = [&](int sl, V3 v) {
    atomicAdd(&GsL[3 * sl], v.x);
    atomicAdd(&GsL[3 * sl + 1], v.y);
    atomicAdd(&GsL[3 * sl + 2], v.z);
  };

  for (int net = 0; net < k; ++net) {
    const int64_t base = (tile * k + net) * (int64_t)d_r * CVF_TILE + l0;
    __syncthreads();  // previous net's qL flushed, gL free
    for (int idx = tid; idx < d_r * kGroup; idx += 64 * kGroup) {
      const int o = idx / kGroup, f = idx % kGroup;
      gL[idx] = g_tiled[base + (int64_t)o * CVF_TILE + f];
    }
    for (int i = lane; i < ns * 3; i += 64) GsL[i] = 0.0f;
    __syncthreads();
    // ---- sparse VJP
    V3 sump = v3(0, 0, 0);
    float M[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = lane; r < pp.n_rec; r += 64) {
      const int32_t* p = pp.rec_slot + 6 * r;
      const Rec rc{p[0], p[1], p[2], p[3], p[4], p[5]};
      if (rc.type == CVF_FEAT_POSITION) {
        const V3 g = v3(gL[rc.out * kGroup + fi], gL[(rc.out + 1) * kGroup + fi], gL[(rc.out + 2) * kGroup + fi]);
        const V3 pv = mat_times(R, g);
        addG(rc.a0, pv);
        sump = sump + pv;
        const V3 xa = xat(rc.a0);
        const V3 xc = v3(xa.x - c[0], xa.y - c[1], xa.z - c[2]);
        M[0] += xc.x * g.x; M[1] += xc.x * g.y; M[2] += xc.x * g.z;
        M[3] += xc.y * g.x; M[4] += xc.y * g.y; M[5] += xc.y * g.z;
        M[6] += xc.z * g.x; M[7] += xc.z * g.y; M[8] += xc.z * g.z;
      } else if (rc.type == CVF_FEAT_BOND) {
        const BondG e = bond_eval(xat(rc.a0), xat(rc.a1));
        const float gs = gL[rc.out * kGroup + fi];
        addG(rc.a0, gs * e.ga);
        addG(rc.a1, gs * e.gb);
      } else if (rc.type == CVF_FEAT_ANGLE) {
        const AngleG e = angle_eval(xat(rc.a0), xat(rc.a1), xat(rc.a2));
        float gs = gL[rc.out * kGroup + fi];
        if (pp.use_angle_value) gs = -gs / sqrtf(fmaxf(1.0f - e.cs * e.cs, 1e-30f));
        addG(rc.a0, gs * e.ga);
        addG(rc.a1, gs * e.gb);
        addG(rc.a2, gs * e.gc);
      } else {
        const DihedralG e = dihedral_eval(xat(rc.a0), xat(rc.a1), xat(rc.a2), xat(rc.a3));
        const float gs = pp.use_angle_value ? gL[rc.out * kGroup + fi]
                                            : (gL[(rc.out + 1) * kGroup + fi] * e.cs - gL[rc.out * kGroup + fi] * e.sn);
        addG(rc.a0, gs * e.g1);
        addG(rc.a1, gs * e.g2);
        addG(rc.a2, gs * e.g3);
        addG(rc.a3, gs * e.g4);
      }
    }
    sump = wave_sum3(sump);
#pragma unroll
    for (int i = 0; i < 9; ++i) M[i] = wave_sumf(M[i]);
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
    const float sh[3] = {inv_nal * sump.x, inv_nal * sump.y, inv_nal * sump.z};
    // ---- dense part from the moments
    float E = 0.0f, usd[3], dH[9];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      float zt1 = 0.0f, quad = 0.0f;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        zt1 += Z[3 * cc + j] * T1[3 * cc + j];
        float row = 0.0f;
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) row += Z[3 * cc + kk] * T2[9 * cc + 3 * kk + j];
        dH[3 * cc + j] = row - sh[cc] * T1[3 * cc + j];
        quad += row * Z[3 * cc + j];
      }
      usd[cc] = zt1 - sh[cc] * T0[cc];
      E += quad - 2.0f * sh[cc] * zt1 + sh[cc] * sh[cc] * T0[cc];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // ---- touched atoms: u_t = a_t .* (s_t + d_t); corrections to E, usum, dH
    float Ep = 0.0f;
    V3 usp = v3(0, 0, 0);
    float dHp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int sl = lane; sl < ns; sl += 64) {
      const int atom = pp.slot_atom[sl];
      const int b = pp.atom_align[atom];
      const V3 st = uat(sl);
      const V3 at = v3(a[3 * atom], a[3 * atom + 1], a[3 * atom + 2]);
      V3 dt = v3(0, 0, 0), rf = v3(0, 0, 0);
      if (b >= 0) {
        rf = v3(pp.ref_c[3 * b], pp.ref_c[3 * b + 1], pp.ref_c[3 * b + 2]);
        const V3 zr = mat_times(Z, rf);
        dt = v3(zr.x - sh[0], zr.y - sh[1], zr.z - sh[2]);
      }
      Ep += at.x * (2.0f * st.x * dt.x + st.x * st.x) + at.y * (2.0f * st.y * dt.y + st.y * st.y) +
            at.z * (2.0f * st.z * dt.z + st.z * st.z);
      const V3 as = v3(at.x * st.x, at.y * st.y, at.z * st.z);
      if (b >= 0) {
        usp = usp + as;
        dHp[0] += as.x * rf.x; dHp[1] += as.x * rf.y; dHp[2] += as.x * rf.z;
        dHp[3] += as.y * rf.x; dHp[4] += as.y * rf.y; dHp[5] += as.y * rf.z;
        dHp[6] += as.z * rf.x; dHp[7] += as.z * rf.y; dHp[8] += as.z * rf.z;
      }
      GsL[3 * sl] = at.x * (st.x + dt.x);
      GsL[3 * sl + 1] = at.y * (st.y + dt.y);
      GsL[3 * sl + 2] = at.z * (st.z + dt.z);
    }
    E += wave_sumf(Ep);
    usp = wave_sum3(usp);
#pragma unroll
    for (int i = 0; i < 9; ++i) dH[i] += wave_sumf(dHp[i]);
    if (lane == 0) e_tiled[(tile * k + net) * CVF_TILE + l0 + fi] = E;
    const float ub[3] = {inv_nal * (usd[0] + usp.x), inv_nal * (usd[1] + usp.y), inv_nal * (usd[2] + usp.z)};
#pragma unroll
    for (int cc = 0; cc < 3; ++cc)
#pragma unroll
      for (int j = 0; j < 3; ++j) dH[3 * cc + j] -= ub[cc] * R1[j];
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
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // ---- JVP: q = J u
    for (int r = lane; r < pp.n_rec; r += 64) {
      const int32_t* p = pp.rec_slot + 6 * r;
      const Rec rc{p[0], p[1], p[2], p[3], p[4], p[5]};
      if (rc.type == CVF_FEAT_POSITION) {
        const V3 u = uat(rc.a0);
        const V3 xa = xat(rc.a0);
        const V3 qa = row_times(v3(u.x - ub[0], u.y - ub[1], u.z - ub[2]), R) +
                      row_times(v3(xa.x - c[0], xa.y - c[1], xa.z - c[2]), dR);
        qL[rc.out * kGroup + fi] = qa.x;
        qL[(rc.out + 1) * kGroup + fi] = qa.y;
        qL[(rc.out + 2) * kGroup + fi] = qa.z;
      } else if (rc.type == CVF_FEAT_BOND) {
        const BondG e = bond_eval(xat(rc.a0), xat(rc.a1));
        qL[rc.out * kGroup + fi] = dot(e.ga, uat(rc.a0)) + dot(e.gb, uat(rc.a1));
      } else if (rc.type == CVF_FEAT_ANGLE) {
        const AngleG e = angle_eval(xat(rc.a0), xat(rc.a1), xat(rc.a2));
        float dv = dot(e.ga, uat(rc.a0)) + dot(e.gb, uat(rc.a1)) + dot(e.gc, uat(rc.a2));
        if (pp.use_angle_value) dv = -dv / sqrtf(fmaxf(1.0f - e.cs * e.cs, 1e-30f));
        qL[rc.out * kGroup + fi] = dv;
      } else {
        const DihedralG e = dihedral_eval(xat(rc.a0), xat(rc.a1), xat(rc.a2), xat(rc.a3));
        const float dphi = dot(e.g1, uat(rc.a0)) + dot(e.g2, uat(rc.a1)) + dot(e.g3, uat(rc.a2)) + dot(e.g4, uat(rc.a3));
        if (pp.use_angle_value) {
          qL[rc.out * kGroup + fi] = dphi;
        } else {
          qL[rc.out * kGroup + fi] = -e.sn * dphi;
          qL[(rc.out + 1) * kGroup + fi] = e.cs * dphi;
        }
      }
    }
    __syncthreads();
    for (int idx = tid; idx < d_r * kGroup; idx += 64 * kGroup) {
      const int o = idx / kGroup, f = idx % kGroup;
      q_tiled[base + (int64_t)o * CVF_TILE + f] = qL[idx];
    }
  }
}

// dense[42] = T0[3], T1[3][3], T2[3][3][3], R1[3] in fp64; one block
__global__ void metric_dense_kernel(cvf_pp_desc pp, const float* __restrict__ a, double* __restrict__ dense) {
  __shared__ double red[42];
  const int tid = threadIdx.x;
  if (tid < 42) red[tid] = 0.0;
  __syncthreads();
  double acc[42];
#pragma unroll
  for (int i = 0; i < 42; ++i) acc[i] = 0.0;
  for (int b = tid; b < pp.n_align; b += blockDim.x) {
    const int atom = pp.align_idx[b];
    double r[3], aw[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      r[j] = (double)pp.ref_c[3 * b + j];
      aw[j] = (double)a[3 * atom + j];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      acc[c] += aw[c];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        acc[3 + 3 * c + j] += aw[c] * r[j];
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) acc[12 + 9 * c + 3 * j + kk] += aw[c] * r[j] * r[kk];
      }
      acc[39 + c] += r[c];
    }
  }
#pragma unroll
  for (int i = 0; i < 42; ++i) {
    const double sv = wave_sum(acc[i]);
    if ((tid & 63) == 0) atomicAdd(&red[i], sv);
  }
  __syncthreads();
  if (tid < 42) dense[tid] = red[tid];
}

}  // namespace

size_t cvf_metric_large_lds(const cvf_pp_desc* pp) {
  return ((size_t)2 * pp->d_r * kGroup + (size_t)2 * kGroup * pp->n_slot * 3) * sizeof(float);
}

int cvf_metric_large_launch(const cvf_pp_desc* pp, int64_t B, const float* aux_tiled, const float* a, int k,
                            const float* slot_xyz, const double* dense, const float* g_tiled, float* q_tiled, float* e_tiled,
                            hipStream_t s) {
  const size_t lds = cvf_metric_large_lds(pp);
  CVF_REQUIRE(lds <= 158 * 1024, "cvf_metric_apply: %d feature atoms x %d features need %zu B of LDS (> 158 KiB)", pp->n_slot,
              pp->d_r, lds);
  (void)hipFuncSetAttribute((const void*)metric_large_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int64_t groups = cvf_ntiles(B) * (CVF_TILE / kGroup);
  hipLaunchKernelGGL(metric_large_kernel, dim3((unsigned)groups), dim3(64 * kGroup), lds, s, *pp, B, aux_tiled, a, k, slot_xyz,
                     dense, g_tiled, q_tiled, e_tiled);
  return cvf_check_launch("metric_large_kernel");
}

extern "C" int cvf_metric_dense_tensors(const cvf_pp_desc* pp, const float* a, double* dense, void* stream) {
  CVF_REQUIRE(pp && a && dense && pp->mode == CVF_PP_ALIGN && pp->align_idx && pp->ref_c, "cvf_metric_dense_tensors: bad argument");
  hipLaunchKernelGGL(metric_dense_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, *pp, a, dense);
  return cvf_check_launch("metric_dense_kernel");
}
