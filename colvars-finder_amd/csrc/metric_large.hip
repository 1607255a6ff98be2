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

// Everything about a feature record that does not depend on the net: its slots, output offset and the gradient
// vectors of the feature with respect to its atoms (for a position record: the centred coordinates).  The records
// a lane owns are evaluated ONCE per frame and kept in registers across the loop over the k nets - the dihedral
// geometry (cross products, two reciprocal square roots, divisions) used to be recomputed twice per net.
struct Geo {
  int to;      // (type + 1) | out << 3   (type -1 = padding entry)
  int s01, s23;   // slots, 16 bits each (n_slot < 65536: the LDS budget caps it far lower)
  V3 v0, v1;           // position: v0 = x - c;  bond: v0 = ga (gb = -ga);  angle: ga, gc (gb = -(ga+gc));  dihedral: g1, g4
  float p, q;          // dihedral: g2 = (-1 - p) g1 + q g4,  g3 = p g1 + (-1 - q) g4
  float cs, sn;        // angle: cs (and sn = -1/sqrt(1-cs^2) for angle-value mode);  dihedral: cos, sin
};
__device__ __forceinline__ int geo_type(const Geo& g) { return (g.to & 7) - 1; }
__device__ __forceinline__ int geo_out(const Geo& g) { return g.to >> 3; }
__device__ __forceinline__ int geo_s0(const Geo& g) { return g.s01 & 0xffff; }
__device__ __forceinline__ int geo_s1(const Geo& g) { return (unsigned)g.s01 >> 16; }
__device__ __forceinline__ int geo_s2(const Geo& g) { return g.s23 & 0xffff; }
__device__ __forceinline__ int geo_s3(const Geo& g) { return (unsigned)g.s23 >> 16; }
constexpr int kGeoPre = 5;   // records per lane held in registers (<= 320 entries); the rest is evaluated on the fly

__global__ __launch_bounds__(64 * kGroup) void metric_large_kernel(cvf_pp_desc pp, int64_t B, const float* __restrict__ aux_tiled,
                                                                   const float* __restrict__ a, int k,
                                                                   const float* __restrict__ slot_xyz,
                                                                   const double* __restrict__ dense,
                                                                   const float* __restrict__ g_tiled,
                                                                   float* __restrict__ q_tiled, float* __restrict__ e_tiled) {
  extern __shared__ float dyn[];
  const int tid = threadIdx.x, lane = tid & 63, fi = tid >> 6;
  CVF_STAMP(20);
  const int ns = pp.n_slot, d_r = pp.d_r, nal = pp.n_align;
  float* gL = dyn;                                  // [d_r][kGroup]
  float* qL = gL + d_r * kGroup;                    // [d_r][kGroup]
  float* xsL = qL + d_r * kGroup + (size_t)fi * ns * 3;          // this wave's slot coordinates
  float* GsL = qL + d_r * kGroup + (size_t)kGroup * ns * 3 + (size_t)fi * ns * 3;  // this wave's slot accumulators
  float* slotC = qL + d_r * kGroup + (size_t)2 * kGroup * ns * 3;   // [ns][8]: a (3), ref (3, zero off the align set), align flag, -
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
  // per-slot constants, shared by the workgroup's frames (the three dependent table look-ups happen once here
  // instead of once per slot, frame and net)
  for (int sl = tid; sl < ns; sl += 64 * kGroup) {
    const int atom = pp.slot_atom[sl];
    const int b = pp.atom_align[atom];
    const int bc = b >= 0 ? b : 0;
    float* sc = slotC + 8 * sl;
    sc[0] = a[3 * atom]; sc[1] = a[3 * atom + 1]; sc[2] = a[3 * atom + 2];
    const float r0 = pp.ref_c[3 * bc], r1 = pp.ref_c[3 * bc + 1], r2 = pp.ref_c[3 * bc + 2];
    sc[3] = b >= 0 ? r0 : 0.0f; sc[4] = b >= 0 ? r1 : 0.0f; sc[5] = b >= 0 ? r2 : 0.0f;
    sc[6] = b >= 0 ? 1.0f : 0.0f;
    sc[7] = 0.0f;
  }
  // dense moments: 42 wave-uniform numbers, kept in LDS (as registers they would be 42 VGPRs: v_cvt_f32_f64 has no
  // scalar form) and read by broadcast where the closed forms need them
  __shared__ float dn[42];
  if (tid < 42) dn[tid] = (float)dense[tid];
  const float* T0 = dn;
  const float* T1 = dn + 3;
  const float* T2 = dn + 12;
  const float* R1 = dn + 39;
  const float inv_nal = 1.0f / (float)nal;
  auto xat = [&](int sl) { return V3{xsL[3 * sl], xsL[3 * sl + 1], xsL[3 * sl + 2]}; };
  auto uat = [&](int sl) { return V3{GsL[3 * sl], GsL[3 * sl + 1], GsL[3 * sl + 2]}; };
  // Scatter into the slot accumulators.  LDS float atomics cost ~850 cycles per wave instruction here (8 waves share
  // the unit): with a batched record list (CVF_PP_SLOT_BATCHED) no two lanes of one instruction name the same slot,
  // and instructions of one wave execute in order, so a plain read-modify-write is exact.
  const bool batched = (pp.flags & CVF_PP_SLOT_BATCHED) != 0;
  const int nrs = pp.n_rec_slot > 0 ? pp.n_rec_slot : pp.n_rec;
  auto addG = [&](int sl, V3 v) {
    float* g3 = GsL + 3 * sl;
    if (batched) {
      g3[0] += v.x;
      g3[1] += v.y;
      g3[2] += v.z;
    } else {
      atomicAdd(g3, v.x);
      atomicAdd(g3 + 1, v.y);
      atomicAdd(g3 + 2, v.z);
    }
  };
  // CVF_PP_SLOT_DISJOINT: no slot occurs twice in a batch, so all accumulators of ONE record are read first and written
  // afterwards - one LDS round trip per record instead of one per component (without the flag a lane's slot may be another
  // lane's slot in a different atom position, and only the component-by-component order is exact)
  const bool disjoint = batched && (pp.flags & CVF_PP_SLOT_DISJOINT) != 0;
  auto addG2 = [&](int sa, V3 va, int sb, V3 vb) {
    if (!disjoint) { addG(sa, va); addG(sb, vb); return; }
    float* ga = GsL + 3 * sa; float* gb = GsL + 3 * sb;
    const float a0 = ga[0], a1 = ga[1], a2 = ga[2], b0 = gb[0], b1 = gb[1], b2 = gb[2];
    ga[0] = a0 + va.x; ga[1] = a1 + va.y; ga[2] = a2 + va.z;
    gb[0] = b0 + vb.x; gb[1] = b1 + vb.y; gb[2] = b2 + vb.z;
  };
  auto addG3 = [&](int sa, V3 va, int sb, V3 vb, int sc, V3 vc) {
    if (!disjoint) { addG(sa, va); addG(sb, vb); addG(sc, vc); return; }
    float* ga = GsL + 3 * sa; float* gb = GsL + 3 * sb; float* gc = GsL + 3 * sc;
    const float a0 = ga[0], a1 = ga[1], a2 = ga[2], b0 = gb[0], b1 = gb[1], b2 = gb[2], c0 = gc[0], c1 = gc[1], c2 = gc[2];
    ga[0] = a0 + va.x; ga[1] = a1 + va.y; ga[2] = a2 + va.z;
    gb[0] = b0 + vb.x; gb[1] = b1 + vb.y; gb[2] = b2 + vb.z;
    gc[0] = c0 + vc.x; gc[1] = c1 + vc.y; gc[2] = c2 + vc.z;
  };
  auto addG4 = [&](int sa, V3 va, int sb, V3 vb, int sc, V3 vc, int sd, V3 vd) {
    if (!disjoint) { addG(sa, va); addG(sb, vb); addG(sc, vc); addG(sd, vd); return; }
    float* ga = GsL + 3 * sa; float* gb = GsL + 3 * sb; float* gc = GsL + 3 * sc; float* gd = GsL + 3 * sd;
    const float a0 = ga[0], a1 = ga[1], a2 = ga[2], b0 = gb[0], b1 = gb[1], b2 = gb[2];
    const float c0 = gc[0], c1 = gc[1], c2 = gc[2], d0 = gd[0], d1 = gd[1], d2 = gd[2];
    ga[0] = a0 + va.x; ga[1] = a1 + va.y; ga[2] = a2 + va.z;
    gb[0] = b0 + vb.x; gb[1] = b1 + vb.y; gb[2] = b2 + vb.z;
    gc[0] = c0 + vc.x; gc[1] = c1 + vc.y; gc[2] = c2 + vc.z;
    gd[0] = d0 + vd.x; gd[1] = d1 + vd.y; gd[2] = d2 + vd.z;
  };
  auto make_geo = [&](int r) {
    const int32_t* p = pp.rec_slot + 6 * r;
    Geo ge;
    ge.to = (p[0] + 1) | (p[5] << 3);
    ge.s01 = p[1] | (p[2] << 16);
    ge.s23 = p[3] | (p[4] << 16);
    ge.v0 = ge.v1 = v3(0, 0, 0);
    ge.p = ge.q = ge.cs = ge.sn = 0.0f;
    if (geo_type(ge) < 0) {
      // padding entry
    } else if (geo_type(ge) == CVF_FEAT_POSITION) {
      const V3 xa = xat(geo_s0(ge));
      ge.v0 = v3(xa.x - c[0], xa.y - c[1], xa.z - c[2]);
    } else if (geo_type(ge) == CVF_FEAT_BOND) {
      ge.v0 = bond_eval(xat(geo_s0(ge)), xat(geo_s1(ge))).ga;
    } else if (geo_type(ge) == CVF_FEAT_ANGLE) {
      const AngleG e = angle_eval(xat(geo_s0(ge)), xat(geo_s1(ge)), xat(geo_s2(ge)));
      ge.v0 = e.ga; ge.v1 = e.gc;
      ge.cs = e.cs;
      ge.sn = -1.0f / sqrtf(fmaxf(1.0f - e.cs * e.cs, 1e-30f));
    } else {
      const DihedralG e = dihedral_eval(xat(geo_s0(ge)), xat(geo_s1(ge)), xat(geo_s2(ge)), xat(geo_s3(ge)));
      ge.v0 = e.g1; ge.v1 = e.g4;
      ge.p = e.p; ge.q = e.q;
      ge.cs = e.cs; ge.sn = e.sn;
    }
    return ge;
  };
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();   // this wave's xsL is in place (wave-private region)
  Geo pre[kGeoPre];
#pragma unroll
  for (int it = 0; it < kGeoPre; ++it) pre[it] = make_geo(lane + 64 * it < nrs ? lane + 64 * it : nrs - 1);

  CVF_STAMP(21);
  for (int net = 0; net < k; ++net) {
    const int64_t base = (tile * k + net) * (int64_t)d_r * CVF_TILE + l0;
    __syncthreads();  // previous net's qL flushed, gL free (first pass: slotC complete)
    // (eight loads in flight per thread, then the LDS writes: one load at a time, each followed by its LDS write, was six
    //  dependent round trips to memory per net - 26 k of the 48 k cycles a net took)
    for (int i0 = tid; i0 < d_r * kGroup; i0 += 64 * kGroup * 8) {
      float gv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = i0 + 64 * kGroup * u;
        const int ic = idx < d_r * kGroup ? idx : d_r * kGroup - 1;
        gv[u] = g_tiled[base + (int64_t)(ic / kGroup) * CVF_TILE + ic % kGroup];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = i0 + 64 * kGroup * u;
        if (idx < d_r * kGroup) gL[idx] = gv[u];
      }
    }
    for (int i = lane; i < ns * 3; i += 64) GsL[i] = 0.0f;
    __syncthreads();
    if (net == 0) CVF_STAMP(22);
    // ---- sparse VJP
    V3 sump = v3(0, 0, 0);
    float M[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto vjp = [&](const Geo& ge) {
      const int ty = geo_type(ge);
      if (ty < 0) return;
      const float* gp = gL + geo_out(ge) * kGroup + fi;
      if (ty == CVF_FEAT_POSITION) {
        const V3 g = v3(gp[0], gp[kGroup], gp[2 * kGroup]);
        const V3 pv = mat_times(R, g);
        addG(geo_s0(ge), pv);
        sump = sump + pv;
        const V3 xc = ge.v0;
        M[0] += xc.x * g.x; M[1] += xc.x * g.y; M[2] += xc.x * g.z;
        M[3] += xc.y * g.x; M[4] += xc.y * g.y; M[5] += xc.y * g.z;
        M[6] += xc.z * g.x; M[7] += xc.z * g.y; M[8] += xc.z * g.z;
      } else if (ty == CVF_FEAT_BOND) {
        const V3 ga = gp[0] * ge.v0;
        addG2(geo_s0(ge), ga, geo_s1(ge), v3(-ga.x, -ga.y, -ga.z));
      } else if (ty == CVF_FEAT_ANGLE) {
        float gs = gp[0];
        if (pp.use_angle_value) gs *= ge.sn;
        const V3 ga = gs * ge.v0, gc = gs * ge.v1;
        addG3(geo_s0(ge), ga, geo_s1(ge), v3(-ga.x - gc.x, -ga.y - gc.y, -ga.z - gc.z), geo_s2(ge), gc);
      } else {
        const float gs = pp.use_angle_value ? gp[0] : (gp[kGroup] * ge.cs - gp[0] * ge.sn);
        const V3 g1 = gs * ge.v0, g4 = gs * ge.v1;
        addG4(geo_s0(ge), g1, geo_s1(ge), (-1.0f - ge.p) * g1 + ge.q * g4, geo_s2(ge), ge.p * g1 + (-1.0f - ge.q) * g4, geo_s3(ge), g4);
      }
    };
#pragma unroll
    for (int it = 0; it < kGeoPre; ++it)
      if (lane + 64 * it < nrs) vjp(pre[it]);
    for (int r = lane + 64 * kGeoPre; r < nrs; r += 64) vjp(make_geo(r));
    if (net == 0) CVF_STAMP(23);
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
    if (net == 0) CVF_STAMP(24);
    // ---- touched atoms: u_t = a_t .* (s_t + d_t); corrections to E, usum, dH
    float Ep = 0.0f;
    V3 usp = v3(0, 0, 0);
    float dHp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 5
    for (int sl = lane; sl < ns; sl += 64) {
      const float4 c0 = *reinterpret_cast<const float4*>(slotC + 8 * sl);
      const float4 c1 = *reinterpret_cast<const float4*>(slotC + 8 * sl + 4);
      const V3 at = v3(c0.x, c0.y, c0.z), rf = v3(c0.w, c1.x, c1.y);
      const float al = c1.z;   // 1 on the align set, else 0 (then rf = 0 as well)
      const V3 st = uat(sl);
      const V3 zr = mat_times(Z, rf);
      const V3 dt = v3(al * (zr.x - sh[0]), al * (zr.y - sh[1]), al * (zr.z - sh[2]));
      Ep += at.x * (2.0f * st.x * dt.x + st.x * st.x) + at.y * (2.0f * st.y * dt.y + st.y * st.y) +
            at.z * (2.0f * st.z * dt.z + st.z * st.z);
      const V3 as = v3(al * at.x * st.x, al * at.y * st.y, al * at.z * st.z);
      usp = usp + as;
      dHp[0] += as.x * rf.x; dHp[1] += as.x * rf.y; dHp[2] += as.x * rf.z;
      dHp[3] += as.y * rf.x; dHp[4] += as.y * rf.y; dHp[5] += as.y * rf.z;
      dHp[6] += as.z * rf.x; dHp[7] += as.z * rf.y; dHp[8] += as.z * rf.z;
      GsL[3 * sl] = at.x * (st.x + dt.x);
      GsL[3 * sl + 1] = at.y * (st.y + dt.y);
      GsL[3 * sl + 2] = at.z * (st.z + dt.z);
    }
    if (net == 0) CVF_STAMP(25);
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
    if (net == 0) CVF_STAMP(26);
    // ---- JVP: q = J u
    auto jvp = [&](const Geo& ge) {
      const int ty = geo_type(ge);
      if (ty < 0) return;
      float* qp = qL + geo_out(ge) * kGroup + fi;
      if (ty == CVF_FEAT_POSITION) {
        const V3 u = uat(geo_s0(ge));
        const V3 qa = row_times(v3(u.x - ub[0], u.y - ub[1], u.z - ub[2]), R) + row_times(ge.v0, dR);
        qp[0] = qa.x;
        qp[kGroup] = qa.y;
        qp[2 * kGroup] = qa.z;
      } else if (ty == CVF_FEAT_BOND) {
        qp[0] = dot(ge.v0, uat(geo_s0(ge)) - uat(geo_s1(ge)));
      } else if (ty == CVF_FEAT_ANGLE) {
        const V3 ub_ = uat(geo_s1(ge));
        float dv = dot(ge.v0, uat(geo_s0(ge)) - ub_) + dot(ge.v1, uat(geo_s2(ge)) - ub_);
        if (pp.use_angle_value) dv *= ge.sn;
        qp[0] = dv;
      } else {
        const V3 u1 = uat(geo_s0(ge)), u2 = uat(geo_s1(ge)), u3 = uat(geo_s2(ge)), u4 = uat(geo_s3(ge));
        // g1.u1 + g2.u2 + g3.u3 + g4.u4 with g2, g3 expressed through g1, g4
        const V3 w1 = u1 + (-1.0f - ge.p) * u2 + ge.p * u3;
        const V3 w4 = u4 + ge.q * u2 + (-1.0f - ge.q) * u3;
        const float dphi = dot(ge.v0, w1) + dot(ge.v1, w4);
        if (pp.use_angle_value) {
          qp[0] = dphi;
        } else {
          qp[0] = -ge.sn * dphi;
          qp[kGroup] = ge.cs * dphi;
        }
      }
    };
#pragma unroll
    for (int it = 0; it < kGeoPre; ++it)
      if (lane + 64 * it < nrs) jvp(pre[it]);
    for (int r = lane + 64 * kGeoPre; r < nrs; r += 64) jvp(make_geo(r));
    if (net == 0) CVF_STAMP(27);
    __syncthreads();
    for (int idx = tid; idx < d_r * kGroup; idx += 64 * kGroup) {
      const int o = idx / kGroup, f = idx % kGroup;
      q_tiled[base + (int64_t)o * CVF_TILE + f] = qL[idx];
    }
    if (net == 0) CVF_STAMP(28);
  }
  CVF_STAMP(29);
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
  return ((size_t)2 * pp->d_r * kGroup + (size_t)2 * kGroup * pp->n_slot * 3 + (size_t)8 * pp->n_slot) * sizeof(float);
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
