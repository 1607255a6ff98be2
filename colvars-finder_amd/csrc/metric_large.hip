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
//
// Work decomposition (round 3; rounds 1-2 ran one wave per frame with the records over the lanes and scattered into per-frame
// slot accumulators in LDS - 8 waves per CU, dependent read-modify-write chains, 117 us for 2000 frames x 6 nets):
//   * one workgroup = F frames (16 when the table below fits the LDS) x ONE net; a lane is (frame f = lane % F, group = lane / F),
//     the 64/F groups of the 16 waves ("lane groups") share out the RECORDS (phases A, C) and the SLOTS (phase B).  Every table
//     index is uniform over a group's F lanes, every global access is F consecutive floats of a [row][64 frames] tile
//     (g, q, aux; the slot coordinates in K1's groups of eight frames), and sums over records / slots are plain per-lane sums
//     followed by ONE fixed-order reduction over the lane groups per phase: no cross-lane work inside the loops.
//   * the scatter J^T g -> slots is a table of contribution rows in LDS, [n_ref][3][F]: atom p of record r owns row
//     mrec[r].row[p] (nobody else writes it), and the rows of one slot are contiguous (slot_row[t] .. slot_row[t+1]) - phase A
//     writes rows, phase B sums each slot's rows, adds the rigid-body term, multiplies by a and leaves u_t in the slot's first
//     row, phase C reads the u rows of its record's atoms.  No atomics, no read-modify-write, no batching constraint on
//     the record list.
//   * the records a lane group owns keep their geometry (gradient vectors of the feature) in registers from phase A to C.
#include "cvf_metric.hpp"

namespace {

constexpr int kNRed = 13;        // values per block reduction (phase A: 12, phase B: 13)
constexpr int kMrecInts = 8;     // mrec entry (include/cvf.h): (type + 1) | out << 3, slots (2 x 16 bits) x 2, u rows x 2, row offsets (4 x 8 bits), 0, 0

// Everything about a feature record that does not depend on the net: output offset, the u rows of its atoms and the gradient
// vectors of the feature with respect to its atoms (for a position record: the centred coordinates).
struct Geo {
  int to;              // (type + 1) | out << 3   (type -1 = no record)
  int u01, u23;        // u rows of the atoms (= first rows of their slots), 16 bits each (n_ref < 65536)
  int off;             // contribution row of atom p = its u row + byte p of off
  V3 v0, v1;           // position: v0 = x - c;  bond: v0 = ga (gb = -ga);  angle: ga, gc (gb = -(ga+gc));  dihedral: g1, g4
  float p, q;          // dihedral: g2 = (-1 - p) g1 + q g4,  g3 = p g1 + (-1 - q) g4
  float cs, sn;        // angle: cs (and sn = -1/sqrt(1-cs^2) for angle-value mode);  dihedral: cos, sin
};
__device__ __forceinline__ int geo_type(const Geo& g) { return (g.to & 7) - 1; }
__device__ __forceinline__ int geo_out(const Geo& g) { return g.to >> 3; }

__device__ __forceinline__ float group_xor(float v, int off) {   // lane ^ off, off a multiple of F
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((int)(threadIdx.x & 63) ^ off) << 2, __builtin_bit_cast(int, v)));
}

// Sum of v[i] over all lane groups of the workgroup, per frame, in a fixed order; result in every lane.  Two barriers.
template <int F, int kMWaves, int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* red, float* fin, int wave, int f, int grp) {
  constexpr int G = 64 / F;
#pragma unroll
  for (int off = F; off < 64; off <<= 1) {   // (butterfly: the same bits in every lane; the NV exchanges of a level in flight together)
    float o[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) o[i] = group_xor(v[i], off);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] += o[i];
  }
  if (grp == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) red[(i * kMWaves + wave) * F + f] = v[i];
  }
  __syncthreads();
  for (int i = wave; i < NV; i += kMWaves) {   // wave i adds value i over the waves
    float acc = 0.0f;
    for (int j = grp; j < kMWaves; j += G) acc += red[(i * kMWaves + j) * F + f];
#pragma unroll
    for (int off = F; off < 64; off <<= 1) acc += group_xor(acc, off);
    if (grp == 0) fin[i * F + f] = acc;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = fin[i * F + f];
}

// F frames per workgroup, kMWaves waves, kGeoPre records per lane group held in registers (the rest is evaluated twice per net);
// kMulti: the workgroup walks npb nets (else one: the loop below folds away)
template <int F, int kMWaves, int kGeoPre, bool kMulti>
__global__ __launch_bounds__(64 * kMWaves) void metric_rows_kernel(cvf_pp_desc pp, int64_t B, const float* __restrict__ aux_tiled,
                                                                   const float* __restrict__ a, int k,
                                                                   const float* __restrict__ slot_xyz,
                                                                   const double* __restrict__ dense,
                                                                   const float* __restrict__ g_tiled,
                                                                   float* __restrict__ q_tiled, float* __restrict__ e_tiled, int npb,
                                                                   MetricFuse fuse) {
  extern __shared__ float dyn[];
  constexpr int G = 64 / F, LG = kMWaves * G;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int f = lane & (F - 1), grp = lane / F, lg = wave * G + grp;
  CVF_STAMP(20);
  const int ns = pp.n_slot, d_r = pp.d_r, nal = pp.n_align, nref = pp.n_ref, nrec = pp.n_mrec;
  float* rows = dyn;                                   // [nref * 3][F]
  // [8][ns]: a (3), ref (3, zero off the align set), align flag, rows (start | count << 20) of every slot, copied from the table
  // cvf_metric_dense_tensors prepared behind the 42 moments.  One dword per read: the F lanes of a group read the SAME address,
  // which the LDS serves as a broadcast for 32-bit reads - as [ns][8] read by two ds_read_b128 the 16 equal addresses were
  // serialised (3.5 k cycles per three slots instead of 1.6 k; straight from global memory with the next iteration's constants
  // requested one iteration ahead it is 1.6 k as well, for 24 more registers)
  float* slotC = rows + (size_t)nref * 3 * F;
  float* red = slotC + (size_t)ns * 8;                 // [kNRed][kMWaves][F]
  float* fin = red + kNRed * kMWaves * F;              // [kNRed][F]
  // blockIdx -> (frame group, net) such that the k nets of a frame group follow each other on ONE XCD (workgroups go round
  // the 8 XCDs by blockIdx): they read the same slot coordinates and aux rows, which then come from that XCD's L2
  const int nb = gridDim.x, q8 = nb >> 3, r8 = nb & 7, xcd = blockIdx.x & 7, ix = blockIdx.x >> 3;
  const int work = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + ix;
  // npb nets per workgroup, one after the other (k % npb == 0): the records' geometry is evaluated once and kept in registers
  // across them - a third of the kernel's instructions otherwise repeated per net; taken when the batch fills the chip anyway
  if (!kMulti) npb = 1;
  const int kb = k / npb, net0 = (work % kb) * npb;
  const int64_t f0 = (int64_t)(work / kb) * F;
  const int64_t tile = f0 / CVF_TILE;
  const int l0 = (int)(f0 % CVF_TILE) + f;             // this lane's frame inside its tile
  // per-frame constants
  const float* ax = aux_tiled + tile * CVF_AUX_ROWS * CVF_TILE + l0;
  float R[9], Kinv[6], c[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = ax[i * CVF_TILE];
#pragma unroll
  for (int i = 0; i < 3; ++i) c[i] = ax[(9 + i) * CVF_TILE];
#pragma unroll
  for (int i = 0; i < 6; ++i) Kinv[i] = ax[(12 + i) * CVF_TILE];
  const float* xs = slot_xyz + ((f0 + f) >> 3) * (int64_t)ns * 3 * 8 + ((f0 + f) & 7);   // [frame group of 8][ns * 3][8] (K1's copy)
  {
    const float* slotG = reinterpret_cast<const float*>(dense + 42);   // [ns][8]
    for (int i = tid; i < ns * 8; i += 64 * kMWaves) slotC[(i & 7) * ns + (i >> 3)] = slotG[i];
  }
  // dense moments: 42 uniform numbers, kept in LDS and read by broadcast where the closed forms need them
  __shared__ float dn[42];
  if (tid < 42) dn[tid] = (float)dense[tid];
  const float* T0 = dn;
  const float* T1 = dn + 3;
  const float* T2 = dn + 12;
  const float* R1 = dn + 39;
  const float inv_nal = 1.0f / (float)nal;
  auto rowat = [&](int r) { return V3{rows[(3 * r) * F + f], rows[(3 * r + 1) * F + f], rows[(3 * r + 2) * F + f]}; };
  auto put_row = [&](int r, V3 v) {
    rows[(3 * r) * F + f] = v.x;
    rows[(3 * r + 1) * F + f] = v.y;
    rows[(3 * r + 2) * F + f] = v.z;
  };
  // A record is handled in three steps whose loads are unconditional - table entry; coordinates of its (up to) four atoms and
  // its (up to) three g values, unused positions naming slot 0 / a clamped row; geometry - so that the loads of SEVERAL
  // records are in flight together: as one dependent chain per record (entry -> coordinates -> g) phase A was 21 memory
  // round trips long for the 7 records of a lane group, 22 k of the workgroup's 61 k cycles.
  struct RecI { int to, s01, s23, u01, u23, off; };
  struct RecX { float x[12]; };
  auto load_rec = [&](int r) {
    const int4* p = reinterpret_cast<const int4*>(pp.mrec + kMrecInts * (r < nrec ? r : nrec - 1));
    const int4 lo = p[0], hi = p[1];
    return RecI{r < nrec ? lo.x : 0, lo.y, lo.z, lo.w, hi.x, hi.y};
  };
  auto load_x = [&](const RecI& ri) {
    RecX o;
    const int sl[4] = {ri.s01 & 0xffff, (int)((unsigned)ri.s01 >> 16), ri.s23 & 0xffff, (int)((unsigned)ri.s23 >> 16)};
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) o.x[3 * p + cc] = xs[(3 * sl[p] + cc) * 8];
    return o;
  };
  struct RecG { float g[3]; };
  auto load_g = [&](const float* gt, int to) {   // the (up to) three g rows of a record's outputs, clamped
    RecG o;
    const int out = to >> 3;
#pragma unroll
    for (int j = 0; j < 3; ++j) o.g[j] = gt[(int64_t)(out + j < d_r ? out + j : d_r - 1) * CVF_TILE];
    return o;
  };
  auto eval = [&](const RecI& ri, const RecX& rx) {
    Geo ge;
    ge.to = ri.to;
    ge.u01 = ri.u01; ge.u23 = ri.u23;
    ge.off = kMulti ? ri.off : 0;   // (one net: only the first pass needs it, straight from the table entry)
    ge.v0 = ge.v1 = v3(0, 0, 0);
    ge.p = ge.q = ge.cs = ge.sn = 0.0f;
    const int ty = geo_type(ge);
    const V3 x0 = v3(rx.x[0], rx.x[1], rx.x[2]), x1 = v3(rx.x[3], rx.x[4], rx.x[5]);
    const V3 x2 = v3(rx.x[6], rx.x[7], rx.x[8]), x3 = v3(rx.x[9], rx.x[10], rx.x[11]);
    if (ty == CVF_FEAT_POSITION) {
      ge.v0 = v3(x0.x - c[0], x0.y - c[1], x0.z - c[2]);
    } else if (ty == CVF_FEAT_BOND) {
      ge.v0 = bond_eval(x0, x1).ga;
    } else if (ty == CVF_FEAT_ANGLE) {
      const AngleG e = angle_eval(x0, x1, x2);
      ge.v0 = e.ga; ge.v1 = e.gc;
      ge.cs = e.cs;
      ge.sn = -1.0f / sqrtf(fmaxf(1.0f - e.cs * e.cs, 1e-30f));
    } else if (ty == CVF_FEAT_DIHEDRAL) {
      const DihedralG e = dihedral_eval(x0, x1, x2, x3);
      ge.v0 = e.g1; ge.v1 = e.g4;
      ge.p = e.p; ge.q = e.q;
      ge.cs = e.cs; ge.sn = e.sn;
    }
    return ge;
  };
  // ---- phase A: contribution rows of J^T g (record r, atom p -> its own row), position sums
  float sa[kNRed];
  auto vjp = [&](const Geo& ge, int off, const RecG& rx) {
    const int ty = geo_type(ge);
    if (ty < 0) return;
    const int r0 = (ge.u01 & 0xffff) + (off & 255), r1 = (int)((unsigned)ge.u01 >> 16) + ((off >> 8) & 255);
    const int r2 = (ge.u23 & 0xffff) + ((off >> 16) & 255), r3 = (int)((unsigned)ge.u23 >> 16) + (int)((unsigned)off >> 24);
    if (ty == CVF_FEAT_POSITION) {
      const V3 g = v3(rx.g[0], rx.g[1], rx.g[2]);
      const V3 pv = mat_times(R, g);
      put_row(r0, pv);
      sa[0] += pv.x; sa[1] += pv.y; sa[2] += pv.z;
      const V3 xc = ge.v0;
      sa[3] += xc.x * g.x; sa[4] += xc.x * g.y; sa[5] += xc.x * g.z;
      sa[6] += xc.y * g.x; sa[7] += xc.y * g.y; sa[8] += xc.y * g.z;
      sa[9] += xc.z * g.x; sa[10] += xc.z * g.y; sa[11] += xc.z * g.z;
    } else if (ty == CVF_FEAT_BOND) {
      const V3 ga = rx.g[0] * ge.v0;
      put_row(r0, ga);
      put_row(r1, v3(-ga.x, -ga.y, -ga.z));
    } else if (ty == CVF_FEAT_ANGLE) {
      float gs = rx.g[0];
      if (pp.use_angle_value) gs *= ge.sn;
      const V3 ga = gs * ge.v0, gc = gs * ge.v1;
      put_row(r0, ga);
      put_row(r1, v3(-ga.x - gc.x, -ga.y - gc.y, -ga.z - gc.z));
      put_row(r2, gc);
    } else {
      const float gs = pp.use_angle_value ? rx.g[0] : (rx.g[1] * ge.cs - rx.g[0] * ge.sn);
      const V3 g1 = gs * ge.v0, g4 = gs * ge.v1;
      put_row(r0, g1);
      put_row(r1, (-1.0f - ge.p) * g1 + ge.q * g4);
      put_row(r2, ge.p * g1 + (-1.0f - ge.q) * g4);
      put_row(r3, g4);
    }
  };
  Geo pre[kGeoPre];
  constexpr int kChunk = 4;   // records whose coordinates are requested together
  for (int nn = 0; nn < npb; ++nn) {
  const int net = net0 + nn;
  const float* gt = g_tiled + (tile * k + net) * (int64_t)d_r * CVF_TILE + l0;
  float* qt = q_tiled + (tile * k + net) * (int64_t)d_r * CVF_TILE + l0;
#pragma unroll
  for (int i = 0; i < kNRed; ++i) sa[i] = 0.0f;   // sump (3), M (9), -
  if (nn == 0) {   // first net: table entries, coordinates, geometry (kept in `pre` for the other nets)
    RecI ri[kGeoPre];
#pragma unroll
    for (int it = 0; it < kGeoPre; ++it) ri[it] = load_rec(lg + LG * it);
    CVF_STAMP(21);
#pragma unroll
    for (int c0 = 0; c0 < kGeoPre; c0 += kChunk) {
      RecX rx[kChunk];
      RecG rg[kChunk];
#pragma unroll
      for (int j = 0; j < kChunk; ++j)
        if (c0 + j < kGeoPre) {
          rx[j] = load_x(ri[c0 + j]);
          rg[j] = load_g(gt, ri[c0 + j].to);
        }
#pragma unroll
      for (int j = 0; j < kChunk; ++j)
        if (c0 + j < kGeoPre) {
          pre[c0 + j] = eval(ri[c0 + j], rx[j]);
          vjp(pre[c0 + j], ri[c0 + j].off, rg[j]);
        }
    }
  } else {
    RecG rg[kGeoPre];
#pragma unroll
    for (int it = 0; it < kGeoPre; ++it) rg[it] = load_g(gt, pre[it].to);
#pragma unroll
    for (int it = 0; it < kGeoPre; ++it) vjp(pre[it], pre[it].off, rg[it]);
  }
  for (int r = lg + LG * kGeoPre; r < nrec; r += LG) {
    const RecI ri = load_rec(r);
    const RecX rx = load_x(ri);
    vjp(eval(ri, rx), ri.off, load_g(gt, ri.to));
  }
  CVF_STAMP(22);
  block_sum<F, kMWaves, kNRed>(sa, red, fin, wave, f, grp);     // (its barriers also publish the rows and slotC)
  CVF_STAMP(23);
  const V3 sump = v3(sa[0], sa[1], sa[2]);
  const float* M = sa + 3;
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
  // ---- phase B: slots.  s_t = sum of the slot's rows; u_t = a_t .* (s_t + d_t), d_t = al_t (Z ref_t - shift), into its first
  // row.  Accumulated: Es = sum a s^2, usp = sum al a s, dHp = sum (al a s) (x) ref - the cross term 2 sum a s d of E follows
  // from the last two (2 sum_c (Z_c. dHp_c - shift_c usp_c)) after the reduction.  Three slots per iteration, their LDS reads
  // requested together.
  const MatCols Zc = mat_cols(Z);
  float Es = 0.0f;
  f2 usp_xy = {0.0f, 0.0f};
  float usp_z = 0.0f;
  Outer3 dHo = {{{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}}, {0.0f, 0.0f, 0.0f}};
  constexpr int kSlots = 3;
  CVF_STAMP(30);
#ifdef CVF_STAMPS
  int itb = 0;
#endif
  for (int sl0 = lg; sl0 < ns; sl0 += LG * kSlots) {
    float4 c0[kSlots], c1[kSlots];
#pragma unroll
    for (int j = 0; j < kSlots; ++j) {
      const float* sc = slotC + (sl0 + LG * j < ns ? sl0 + LG * j : ns - 1);
      c0[j] = float4{sc[0], sc[ns], sc[2 * ns], sc[3 * ns]};
      c1[j] = float4{sc[4 * ns], sc[5 * ns], sc[6 * ns], sc[7 * ns]};
    }
    V3 st[kSlots];
    int r0[kSlots], cnt[kSlots], mx = 1;
#pragma unroll
    for (int j = 0; j < kSlots; ++j) {
      const int rc = __builtin_bit_cast(int, c1[j].w);
      r0[j] = rc & 0xfffff;
      cnt[j] = (int)((unsigned)rc >> 20);
      mx = cnt[j] > mx ? cnt[j] : mx;
      st[j] = rowat(r0[j]);
    }
    for (int e = 1; e < mx; ++e) {   // atoms shared by several records
#pragma unroll
      for (int j = 0; j < kSlots; ++j) {
        const V3 x = rowat(r0[j] + e < nref ? r0[j] + e : nref - 1);
        const float m = e < cnt[j] ? 1.0f : 0.0f;
        st[j] = st[j] + m * x;
      }
    }
#pragma unroll
    for (int j = 0; j < kSlots; ++j) {
      const bool valid = sl0 + LG * j < ns;
      const float mv = valid ? 1.0f : 0.0f;
      const V3 at = v3(mv * c0[j].x, mv * c0[j].y, mv * c0[j].z), rf = v3(c0[j].w, c1[j].x, c1[j].y);
      const float al = c1[j].z;   // 1 on the align set, else 0 (then rf = 0 as well)
      const V3 as = v3(at.x * st[j].x, at.y * st[j].y, at.z * st[j].z);
      Es += as.x * st[j].x + as.y * st[j].y + as.z * st[j].z;
      const V3 asl = al * as;
      usp_xy += f2{asl.x, asl.y};
      usp_z += asl.z;
      outer_acc(dHo, asl, rf);
      f2 dxy = f2{-al * sh[0], -al * sh[1]};
      float dz = -al * sh[2];
      mat_times_acc(Zc, rf, dxy, dz);
      if (valid) put_row(r0[j], v3(as.x + at.x * dxy.x, as.y + at.y * dxy.y, as.z + at.z * dz));
    }
#ifdef CVF_STAMPS
    if (itb < 8) CVF_STAMP(31 + itb);
    ++itb;
#endif
  }
  float sb[kNRed];
  sb[0] = Es;
  sb[1] = usp_xy.x; sb[2] = usp_xy.y; sb[3] = usp_z;
  outer_to_array(dHo, *reinterpret_cast<float (*)[9]>(sb + 4));
  CVF_STAMP(24);
  block_sum<F, kMWaves, kNRed>(sb, red, fin, wave, f, grp);     // (publishes the u rows)
  CVF_STAMP(25);
  E += sb[0];
#pragma unroll
  for (int cc = 0; cc < 3; ++cc)
    E += 2.0f * (Z[3 * cc] * sb[4 + 3 * cc] + Z[3 * cc + 1] * sb[5 + 3 * cc] + Z[3 * cc + 2] * sb[6 + 3 * cc] - sh[cc] * sb[1 + cc]);
#pragma unroll
  for (int i = 0; i < 9; ++i) dH[i] += sb[4 + i];
  if (lg == 0) e_tiled[(tile * k + net) * CVF_TILE + l0] = E;
  if (fuse.on && wave == 0) {
    // first stage of the batch sums of EigenFunctionTask.loss_func (K5; cvf_metric.hpp: MetricFuse): this workgroup's F frames of the
    // slots its net owns, one row per frame group, [statistic][row]; cvf_ef_stats_finish adds the rows.  fp64, fixed order
    // (row_shr steps inside the 16-lane row that holds group 0's frames; the other groups add zeros).
    const int64_t frame = f0 + f;
    const bool valid = frame < B && grp == 0;
    const double wb = valid ? (double)fuse.w[frame < B ? frame : B - 1] : 0.0;
    const float* yb = fuse.y_tiled + tile * k * CVF_TILE + l0;
    const double yn = valid ? (double)yb[net * CVF_TILE] : 0.0;
    const int64_t n_rows = (int64_t)(gridDim.x / kb), row = work / kb;
    auto put = [&](int slot, double v) {
      v += dpp_movd<0x111, 0xf>(v);
      v += dpp_movd<0x112, 0xf>(v);
      v += dpp_movd<0x114, 0xf>(v);
      v += dpp_movd<0x118, 0xf>(v);   // lane 15 holds the sum of lanes 0..15
      if (lane == 15) fuse.partial[slot * n_rows + row] = v;
    };
    if (net == 0) put(0, wb);
    put(1 + net, wb * yn);
    const int s2o = 1 + k + net * k - (net * (net - 1)) / 2 - net;   // + j : slot of S2[net][j], j >= net
    for (int j = net; j < k; ++j) put(s2o + j, wb * yn * (valid ? (double)yb[j * CVF_TILE] : 0.0));
    put(1 + k + CVF_NPAIR(k) + net, wb * (double)E);
  }
  const float ub[3] = {inv_nal * (usd[0] + sb[1]), inv_nal * (usd[1] + sb[2]), inv_nal * (usd[2] + sb[3])};
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
  // ---- phase C: q = J u, by the records' owners
  auto jvp = [&](const Geo& ge) {
    const int ty = geo_type(ge);
    if (ty < 0) return;
    float* qp = qt + (int64_t)geo_out(ge) * CVF_TILE;
    const int u0 = ge.u01 & 0xffff, u1 = (unsigned)ge.u01 >> 16, u2 = ge.u23 & 0xffff, u3 = (unsigned)ge.u23 >> 16;
    if (ty == CVF_FEAT_POSITION) {
      const V3 u = rowat(u0);
      const V3 qa = row_times(v3(u.x - ub[0], u.y - ub[1], u.z - ub[2]), R) + row_times(ge.v0, dR);
      qp[0] = qa.x;
      qp[CVF_TILE] = qa.y;
      qp[2 * CVF_TILE] = qa.z;
    } else if (ty == CVF_FEAT_BOND) {
      qp[0] = dot(ge.v0, rowat(u0) - rowat(u1));
    } else if (ty == CVF_FEAT_ANGLE) {
      const V3 ub_ = rowat(u1);
      float dv = dot(ge.v0, rowat(u0) - ub_) + dot(ge.v1, rowat(u2) - ub_);
      if (pp.use_angle_value) dv *= ge.sn;
      qp[0] = dv;
    } else {
      const V3 x1 = rowat(u0), x2 = rowat(u1), x3 = rowat(u2), x4 = rowat(u3);
      // g1.u1 + g2.u2 + g3.u3 + g4.u4 with g2, g3 expressed through g1, g4
      const V3 w1 = x1 + (-1.0f - ge.p) * x2 + ge.p * x3;
      const V3 w4 = x4 + ge.q * x2 + (-1.0f - ge.q) * x3;
      const float dphi = dot(ge.v0, w1) + dot(ge.v1, w4);
      if (pp.use_angle_value) {
        qp[0] = dphi;
      } else {
        qp[0] = -ge.sn * dphi;
        qp[CVF_TILE] = ge.cs * dphi;
      }
    }
  };
#pragma unroll
  for (int it = 0; it < kGeoPre; ++it) jvp(pre[it]);
  for (int r = lg + LG * kGeoPre; r < nrec; r += LG) {
    const RecI ri = load_rec(r);
    jvp(eval(ri, load_x(ri)));
  }
  CVF_STAMP(26);
  if (nn + 1 < npb) __syncthreads();   // the u rows are read; the next net writes its contributions over them
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
  // the derivative kernel's per-slot constants behind the moments: [n_slot][8] floats (see metric_rows_kernel)
  if (pp.slot_row != nullptr) {
    float* sc0 = reinterpret_cast<float*>(dense + 42);
    for (int sl = tid; sl < pp.n_slot; sl += blockDim.x) {
      const int atom = pp.slot_atom[sl];
      const int b = pp.atom_align[atom];
      const int bc = b >= 0 ? b : 0;
      const int r0 = pp.slot_row[sl], r1 = pp.slot_row[sl + 1];
      float* sc = sc0 + 8 * sl;
      sc[0] = a[3 * atom]; sc[1] = a[3 * atom + 1]; sc[2] = a[3 * atom + 2];
      sc[3] = b >= 0 ? pp.ref_c[3 * bc] : 0.0f; sc[4] = b >= 0 ? pp.ref_c[3 * bc + 1] : 0.0f; sc[5] = b >= 0 ? pp.ref_c[3 * bc + 2] : 0.0f;
      sc[6] = b >= 0 ? 1.0f : 0.0f;
      sc[7] = __builtin_bit_cast(float, r0 | ((r1 - r0) << 20));
    }
  }
}

}  // namespace

static size_t metric_rows_lds(const cvf_pp_desc* pp, int F, int waves) {
  return ((size_t)pp->n_ref * 3 * F + (size_t)8 * pp->n_slot + (size_t)kNRed * waves * F + (size_t)kNRed * F) * sizeof(float);
}
size_t cvf_metric_large_lds(const cvf_pp_desc* pp) { return metric_rows_lds(pp, 4, 8); }   // the smallest layout

// fuse / fused_rows: see MetricFuse (cvf_metric.hpp); *fused_rows = rows of [statistic][row] partial sums written (0: none)
int cvf_metric_large_launch(const cvf_pp_desc* pp, int64_t B, const float* aux_tiled, const float* a, int k,
                            const float* slot_xyz, const double* dense, const float* g_tiled, float* q_tiled, float* e_tiled,
                            const MetricFuse* fuse, int* fused_rows, hipStream_t s) {
  if (fused_rows) *fused_rows = 0;
  CVF_REQUIRE(pp->mrec && pp->slot_row && pp->n_mrec > 0 && pp->n_ref >= pp->n_slot && pp->n_ref < 65536,
              "cvf_metric_apply: large molecules need the record / row tables (mrec, slot_row, n_mrec, n_ref) of cvf_pp_desc");
  constexpr size_t kBudget = 158 * 1024;
  static const int waves = getenv("CVF_METRIC_WAVES") ? atoi(getenv("CVF_METRIC_WAVES")) : 8;   // developer switch: 8 | 16
  const int W = waves == 16 ? 16 : 8;
  const int F = metric_rows_lds(pp, 16, W) <= kBudget ? 16 : metric_rows_lds(pp, 8, W) <= kBudget ? 8 : 4;
  const size_t lds = metric_rows_lds(pp, F, W);
  CVF_REQUIRE(lds <= kBudget, "cvf_metric_apply: %d feature-atom references need %zu B of LDS (> 158 KiB)", pp->n_ref, lds);
  const int64_t groups = cvf_ntiles(B) * (CVF_TILE / F) * k;
  CVF_REQUIRE(groups < (int64_t)1 << 31, "cvf_metric_apply: batch too large");
  // nets per workgroup (a divisor of k): a workgroup that walks several nets evaluates the records' geometry once (~7 us of a round at
  // the config-5 shape, ~17.5 us per net behind it; one workgroup per CU), but fewer, longer workgroups fill the chip in coarser rounds.
  // Measured at 5000 atoms, k = 6 (round 4, CVF_METRIC_NPB sweep, us per launch for 1 / 2 / 3 / 6 nets): 2000 frames 73 / 84 / 64 / 100,
  // 4000: 132 / 129 / 119 / 107, 8000: 238 / 231 / 212 / 193 (until then: all k nets from 8192 frames on, else one).  The count with the
  // smallest rounds x (geometry + nets) wins; a chip filled more than twice over counts fractional rounds.
  int npb = 1;
  {
    const int ncu = cvf_cu_count();
    double best = 0.0;
    for (int c = 1; c <= k; ++c) {
      if (k % c != 0) continue;
      const double wgs = (double)(groups / k) * (k / c);
      const double rounds = wgs <= 2.0 * ncu ? (double)(((int64_t)wgs + ncu - 1) / ncu) : wgs / ncu;
      const double cost = rounds * (7.0 + 17.5 * c);
      if (c == 1 || cost < best) {
        best = cost;
        npb = c;
      }
    }
  }
  if (getenv("CVF_METRIC_NPB") && k % atoi(getenv("CVF_METRIC_NPB")) == 0 && atoi(getenv("CVF_METRIC_NPB")) > 0) npb = atoi(getenv("CVF_METRIC_NPB"));   // developer switch
  MetricFuse mf = {};
  if (fuse != nullptr && fuse->on && groups / k <= 384) {   // one row per frame group, while the finishing launch reads them in one trip
    mf = *fuse;
    if (fused_rows) *fused_rows = (int)(groups / k);
  }
  auto go = [&](auto kernel) {
    (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kernel, dim3((unsigned)(groups / npb)), dim3(64 * W), lds, s, *pp, B, aux_tiled, a, k, slot_xyz, dense,
                       g_tiled, q_tiled, e_tiled, npb, mf);
  };
  static const int pre_multi = getenv("CVF_METRIC_PRE") ? atoi(getenv("CVF_METRIC_PRE")) : 7;   // developer switch: 5 (no spills) | 7 (17 spilled registers, but 363 against 416 us at 16 000 frames x 6 nets)
  if (W == 16) {
    if (F == 16) go(metric_rows_kernel<16, 16, 2, true>);
    else if (F == 8) go(metric_rows_kernel<8, 16, 2, true>);
    else go(metric_rows_kernel<4, 16, 2, true>);
  } else if (npb > 1) {
    if (F == 16 && pre_multi == 7) go(metric_rows_kernel<16, 8, 7, true>);
    else if (F == 16) go(metric_rows_kernel<16, 8, 5, true>);
    else if (F == 8) go(metric_rows_kernel<8, 8, 4, true>);
    else go(metric_rows_kernel<4, 8, 2, true>);
  } else {
    if (F == 16) go(metric_rows_kernel<16, 8, 7, false>);
    else if (F == 8) go(metric_rows_kernel<8, 8, 4, false>);
    else go(metric_rows_kernel<4, 8, 2, false>);
  }
  return cvf_check_launch("metric_rows_kernel");
}

extern "C" int64_t cvf_metric_dense_doubles(const cvf_pp_desc* pp) {
  return 42 + (pp && pp->slot_row ? 4 * (int64_t)pp->n_slot : 0);
}

extern "C" int cvf_metric_dense_tensors(const cvf_pp_desc* pp, const float* a, double* dense, void* stream) {
  CVF_REQUIRE(pp && a && dense && pp->mode == CVF_PP_ALIGN && pp->align_idx && pp->ref_c, "cvf_metric_dense_tensors: bad argument");
  // (the packed fields of the record / row tables: 16-bit slots and rows; the host validates the 8-bit row offsets and the 12-bit
  //  per-slot row counts when it builds the tables - colvarsfinder/pp.py derivative_table_limits)
  CVF_REQUIRE(pp->slot_row == nullptr || (pp->n_slot < 65536 && pp->n_ref < 65536 && pp->n_ref >= pp->n_slot),
              "cvf_metric_dense_tensors: %d slots / %d contribution rows do not fit the tables' 16-bit fields", pp->n_slot, pp->n_ref);
  hipLaunchKernelGGL(metric_dense_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, *pp, a, dense);
  return cvf_check_launch("metric_dense_kernel");
}
