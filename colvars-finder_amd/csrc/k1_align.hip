// K1: Kabsch alignment + feature map, forward; K2+K3: q = J A J^T g, E = g^T J A J^T g.
// One lane = one frame, one wave = one 64-frame tile.  The tile's coordinates are staged
// once into LDS with coalesced 16-byte global loads (HBM traffic = 12 B/atom/frame in,
// 4 B/feature/frame out); the 3x3 covariance, its eigen-decomposition and the small
// solves run per lane in fp64, so all 64 lanes stay busy for small molecules.
#include "cvf_kabsch.hpp"
#include "cvf_metric.hpp"

namespace {

struct Rec {
  int type, a0, a1, a2, a3, out;
};
__device__ __forceinline__ Rec load_rec(const int32_t* rec, int r) {
  const int32_t* p = rec + 6 * r;
  return Rec{p[0], p[1], p[2], p[3], p[4], p[5]};
}

// The layer's small tables (feature records, align indices, reference coordinates) are copied to LDS once
// per block: with one wave per SIMD a scalar-cache miss per loop iteration (~200+ cycles, nothing to overlap
// with) was a third of the kernels' time; an LDS broadcast read is ~64 cycles and pipelines.
struct Tables {
  const int32_t* rec;
  const int32_t* align_idx;
  const float* ref_c;
};
__host__ __device__ inline int tables_dwords(const cvf_pp_desc& pp) { return 6 * pp.n_rec + 4 * pp.n_align; }
// Word i of the concatenated tables [rec | align_idx | ref_c], read from global memory by one unconditional load.
__device__ __forceinline__ int32_t table_word(const cvf_pp_desc& pp, int i) {
  const int n1 = 6 * pp.n_rec, n2 = n1 + pp.n_align;
  const int32_t* p = i < n1 ? pp.rec + i
                            : (i < n2 ? pp.align_idx + (i - n1) : reinterpret_cast<const int32_t*>(pp.ref_c) + (i - n2));
  return *p;
}
// The first kTablePre words per lane are fetched into registers *before* the coordinate tile is staged, so their
// round trip overlaps the tile's instead of following it; tables_commit writes them (and any remainder) to LDS.
constexpr int kTablePre = 4;
struct TablePrefetch {
  int32_t w[kTablePre];
};
__device__ __forceinline__ TablePrefetch tables_prefetch(const cvf_pp_desc& pp, int lane) {
  const int n = tables_dwords(pp);
  TablePrefetch tp;
#pragma unroll
  for (int i = 0; i < kTablePre; ++i) {
    const int j = lane + CVF_WAVE * i;
    tp.w[i] = table_word(pp, j < n ? j : n - 1);
  }
  return tp;
}
__device__ __forceinline__ Tables tables_commit(const cvf_pp_desc& pp, const TablePrefetch& tp, float* lds, int lane) {
  const int n = tables_dwords(pp);
  int32_t* L = reinterpret_cast<int32_t*>(lds);
#pragma unroll
  for (int i = 0; i < kTablePre; ++i) {
    const int j = lane + CVF_WAVE * i;
    if (j < n) L[j] = tp.w[i];
  }
  for (int j = lane + CVF_WAVE * kTablePre; j < n; j += CVF_WAVE) L[j] = table_word(pp, j);
  int32_t* alL = L + 6 * pp.n_rec;
  return Tables{L, alL, reinterpret_cast<const float*>(alL + pp.n_align)};
}
__device__ __forceinline__ V3 atom(const float* my, int a) { return V3{my[3 * a], my[3 * a + 1], my[3 * a + 2]}; }

// per-lane alignment: centroid (fp64), covariance (fp64), rotation, Kinv
// CONTIG: align atom b is frame atom b (CVF_PP_ALIGN_CONTIG): every LDS address is affine in the loop
// counter, so the compiler batches the reads of several atoms behind one wait.
template <bool CONTIG>
__device__ __forceinline__ void align_lane(const cvf_pp_desc& pp, const Tables& tb, const float* my, double (&c)[3],
                                           KabschOut& ko) {
  // One pass: H = sum_b (x_b - c) r_b^T = sum_b x_b r_b^T - c (sum_b r_b)^T, and the reference is stored centred,
  // so the second term is n_align * c * (fp32 rounding residue of the reference's mean)^T - kept, it costs nothing.
  double cx = 0, cy = 0, cz = 0, rs0 = 0, rs1 = 0, rs2 = 0;
  double H[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll 2
  for (int b = 0; b < pp.n_align; ++b) {
    const int a = CONTIG ? b : tb.align_idx[b];
    const double x0 = (double)my[3 * a], x1 = (double)my[3 * a + 1], x2 = (double)my[3 * a + 2];
    const double r0 = (double)tb.ref_c[3 * b], r1 = (double)tb.ref_c[3 * b + 1], r2 = (double)tb.ref_c[3 * b + 2];
    if (!CONTIG && pp.align_w) {   // weighted centroid (cvf_pp_desc.align_w has mean 1; ref_c carries the weights already)
      const double wb = (double)pp.align_w[b];
      cx = fma(wb, x0, cx); cy = fma(wb, x1, cy); cz = fma(wb, x2, cz);
    } else {
      cx += x0; cy += x1; cz += x2;
    }
    rs0 += r0; rs1 += r1; rs2 += r2;
    H[0][0] = fma(x0, r0, H[0][0]); H[0][1] = fma(x0, r1, H[0][1]); H[0][2] = fma(x0, r2, H[0][2]);
    H[1][0] = fma(x1, r0, H[1][0]); H[1][1] = fma(x1, r1, H[1][1]); H[1][2] = fma(x1, r2, H[1][2]);
    H[2][0] = fma(x2, r0, H[2][0]); H[2][1] = fma(x2, r1, H[2][1]); H[2][2] = fma(x2, r2, H[2][2]);
  }
  const double inv = fast_rcp((double)pp.n_align);
  c[0] = cx * inv;
  c[1] = cy * inv;
  c[2] = cz * inv;
  const double rs[3] = {rs0, rs1, rs2};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) H[i][j] = fma(-c[i], rs[j], H[i][j]);
  CVF_STAMP(3);
  kabsch_from_H(H, ko);
  CVF_STAMP(4);
}

// TILED / ROWS: which of the two output layouts is written - compile-time, because a run-time `if (ft)` around
// each of the d_r stores became an exec-mask branch per store (2/3 of the feature loop's time).
template <bool FAST, bool TILED, bool ROWS>
__global__ __launch_bounds__(64) void k1_align_kernel(cvf_pp_desc pp, const float* __restrict__ x, int64_t B,
                                                       float* __restrict__ feat_tiled, float* __restrict__ feat_rows,
                                                       float* __restrict__ aux_tiled) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  const int64_t tile = blockIdx.x;
  const int nc = pp.n_coord;
  CVF_STAMP(0);
  const TablePrefetch tp = tables_prefetch(pp, lane);
  load_x_tile(x, B, nc, tile, lds, lane);
  CVF_STAMP(1);
  const Tables tb = tables_commit(pp, tp, lds + CVF_TILE * x_tile_stride(nc), lane);
  __syncthreads();
  CVF_STAMP(2);
  const float* my = lds + lane * x_tile_stride(nc);
  double cd[3];
  KabschOut ko;
  align_lane<FAST>(pp, tb, my, cd, ko);
  const Centre c = centre_of(cd);
  CVF_STAMP(5);
  if (aux_tiled) {
    float* ax = aux_tiled + tile * CVF_AUX_ROWS * CVF_TILE + lane;
#pragma unroll
    for (int i = 0; i < 9; ++i) ax[i * CVF_TILE] = ko.R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) ax[(9 + i) * CVF_TILE] = c.hi[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) ax[(12 + i) * CVF_TILE] = ko.Kinv[i];
  }
  const int64_t frame = tile * CVF_TILE + lane;
  float* ft = TILED ? feat_tiled + tile * pp.d_r * CVF_TILE + lane : nullptr;
  // frames past B (tail tile) write their row into the last valid frame's slot: same value, no predicate
  float* fr = ROWS ? feat_rows + (frame < B ? frame : B - 1) * pp.d_r : nullptr;
  auto emit = [&](int o, float v) {
    if (TILED) ft[o * CVF_TILE] = v;
    if (ROWS) fr[o] = v;
  };
  if (FAST) {  // CVF_PP_PURE_POSITION: record r = position of atom r -> outputs 3r..3r+2
#pragma unroll 4
    for (int a = 0; a < pp.n_rec; ++a) {
      const V3 al = row_times(centred(my, a, c), ko.R);
      emit(3 * a, al.x);
      emit(3 * a + 1, al.y);
      emit(3 * a + 2, al.z);
    }
    CVF_STAMP(6);
    return;
  }
#pragma unroll 2
  for (int r = 0; r < pp.n_rec; ++r) {
    const Rec rc = load_rec(tb.rec, r);
    if (rc.type == CVF_FEAT_POSITION) {
      const V3 al = row_times(centred(my, rc.a0, c), ko.R);
      emit(rc.out, al.x);
      emit(rc.out + 1, al.y);
      emit(rc.out + 2, al.z);
    } else if (rc.type == CVF_FEAT_BOND) {
      emit(rc.out, bond_eval(atom(my, rc.a0), atom(my, rc.a1)).val);
    } else if (rc.type == CVF_FEAT_ANGLE) {
      const float cs = angle_eval(atom(my, rc.a0), atom(my, rc.a1), atom(my, rc.a2)).cs;
      emit(rc.out, pp.use_angle_value ? acosf(cs) : cs);
    } else {
      const DihedralG dg = dihedral_eval(atom(my, rc.a0), atom(my, rc.a1), atom(my, rc.a2), atom(my, rc.a3));
      if (pp.use_angle_value) {
        emit(rc.out, atan2f(dg.sn, dg.cs));
      } else {
        emit(rc.out, dg.cs);
        emit(rc.out + 1, dg.sn);
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// K1, fast layouts, large launches: persistent waves streaming whole 64-frame tiles (lane = frame).
//  * the tile is copied to LDS as it lies in memory (16-byte loads -> 16-byte LDS writes, no index arithmetic): a
//    lane's frame starts at dword lane * nc, and for nc = 2 (mod 4) the lanes' 8-byte reads of one instruction
//    cover each bank exactly once - so the per-lane loops read atom PAIRS (three 8-byte reads);
//  * the wave walks tiles blockIdx, blockIdx + grid, ...; the NEXT tile's loads are issued into NV x 4 registers
//    before the current tile is worked on and land in LDS at the top of the next iteration, so every resident
//    wave always has 64 * nc * 4 bytes in flight (the one-tile-per-wave kernel spent half its time waiting for
//    its own loads: 16 k of 32 k cycles at one million frames);
//  * the reference is converted to fp64 once per wave and kept in LDS;
//  * the row-major output (ROWS) is written over the tile in LDS and leaves as it came, in 16-byte stores.
// Odd 3 N: the same with 4-byte reads (an odd stride is conflict-free as it is); 3 N = 0 (mod 4) keeps the one-tile kernel.
// Frames past the last whole tile are left to k1_align_kernel.
// ------------------------------------------------------------------------------------
template <int NV, bool TILED, bool ROWS, bool PAIR>
__global__ __launch_bounds__(64) void k1_stream_kernel(cvf_pp_desc pp, const float* __restrict__ x, int64_t n_tiles,
                                                        float* __restrict__ feat_tiled, float* __restrict__ feat_rows,
                                                        float* __restrict__ aux_tiled) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  const int nc = pp.n_coord;
  const int nvec = 16 * nc;   // 16-byte vectors per tile
  float4* tile4 = reinterpret_cast<float4*>(lds);
  float4* ref4 = reinterpret_cast<float4*>(lds + CVF_TILE * nc);   // (r0, r1, r2, 0) per align atom: one broadcast read
  for (int b = lane; b < pp.n_align; b += CVF_WAVE)
    ref4[b] = float4{pp.ref_c[3 * b], pp.ref_c[3 * b + 1], pp.ref_c[3 * b + 2], 0.0f};
  float4 pre[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) pre[i] = float4{0.0f, 0.0f, 0.0f, 0.0f};
  auto fetch = [&](int64_t tile) {
    const float4* src = reinterpret_cast<const float4*>(x + tile * CVF_TILE * nc);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = i * CVF_WAVE + lane;
      if (v < nvec) pre[i] = src[v];
    }
  };
  int64_t tile = blockIdx.x;
  if (tile < n_tiles) fetch(tile);
  __syncthreads();
  // sum of the (centred) reference: the fp32 rounding residue of its mean, see align_lane
  double rs[3] = {0, 0, 0};
  for (int b = 0; b < pp.n_align; ++b) {
    const float4 r = ref4[b];
    rs[0] += (double)r.x;
    rs[1] += (double)r.y;
    rs[2] += (double)r.z;
  }
  const double inv_n = fast_rcp((double)pp.n_align);
  float* my = lds + lane * nc;
  f2* my2 = reinterpret_cast<f2*>(my);
  for (int it = 0; tile < n_tiles; tile += gridDim.x, ++it) {
    if (it == 2) CVF_STAMP(0);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = i * CVF_WAVE + lane;
      if (v < nvec) tile4[v] = pre[i];
    }
    if (it == 2) CVF_STAMP(1);
    if (tile + gridDim.x < n_tiles) fetch(tile + gridDim.x);
    if (it == 2) CVF_STAMP(2);
    // ---- centroid + covariance in one pass over d = x - (atom 0 of the frame): the differences are molecule-sized
    // whatever the frame's distance from the origin, so fp32 products and sums (packed, two per instruction) carry
    // the reference's own fp32 accuracy without the 3 N quarter-rate fp32 -> fp64 conversions of the fp64 pass.
    // Register pairs follow the tile: [x y] [z x'] [y' z'] per atom pair.
    const f2 P0 = PAIR ? my2[0] : f2{my[0], my[1]};   // (an odd stride leaves odd lanes' rows 4-byte aligned only)
    const float pz = my[2];
    const f2 P1 = f2{pz, P0.x}, P2 = f2{P0.y, pz};
    f2 sA = {0, 0}, sB = {0, 0}, sC = {0, 0};
    f2 H01[3] = {{0, 0}, {0, 0}, {0, 0}};   // H[i][0], H[i][1]
    float H2[3] = {0, 0, 0};                // H[i][2]
    auto acc = [&](float d0, float d1, float d2, const float4 r) {
      const f2 r01 = f2{r.x, r.y};
      H01[0] = fma2(splat2(d0), r01, H01[0]); H2[0] = fmaf(d0, r.z, H2[0]);
      H01[1] = fma2(splat2(d1), r01, H01[1]); H2[1] = fmaf(d1, r.z, H2[1]);
      H01[2] = fma2(splat2(d2), r01, H01[2]); H2[2] = fmaf(d2, r.z, H2[2]);
    };
    const int npa = PAIR ? pp.n_align >> 1 : 0;   // PAIR: 3N = 2 (mod 4), 8-byte reads of atom pairs; else 3N odd, 4-byte reads
#pragma unroll 2
    for (int m = 0; m < npa; ++m) {
      const f2 d0 = my2[3 * m] - P0, d1 = my2[3 * m + 1] - P1, d2 = my2[3 * m + 2] - P2;
      sA += d0; sB += d1; sC += d2;
      acc(d0.x, d0.y, d1.x, ref4[2 * m]);
      acc(d1.y, d2.x, d2.y, ref4[2 * m + 1]);
    }
    float sx = sA.x + sB.y, sy = sA.y + sC.x, sz = sB.x + sC.y;
#pragma unroll 2
    for (int a = 2 * npa; a < pp.n_align; ++a) {   // the odd last atom, or every atom when the stride is odd
      const float d0 = my[3 * a] - P0.x, d1 = my[3 * a + 1] - P0.y, d2 = my[3 * a + 2] - pz;
      sx += d0; sy += d1; sz += d2;
      acc(d0, d1, d2, ref4[a]);
    }
    // centroid relative to the pivot (fp64 -> fp32), absolute centroid for the aux rows
    const double cr[3] = {(double)sx * inv_n, (double)sy * inv_n, (double)sz * inv_n};
    const float cf[3] = {(float)cr[0], (float)cr[1], (float)cr[2]};
    double H[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      H[i][0] = fma(-cr[i], rs[0], (double)H01[i].x);
      H[i][1] = fma(-cr[i], rs[1], (double)H01[i].y);
      H[i][2] = fma(-cr[i], rs[2], (double)H2[i]);
    }
    if (it == 2) CVF_STAMP(3);
    KabschOut ko;
    if (aux_tiled) kabsch_from_H<true>(H, ko);    // (uniform) rotation + K^-1 for the derivative kernels
    else kabsch_from_H<false>(H, ko);             // features only: the rotation alone, one Newton step less
    if (it == 2) CVF_STAMP(4);
    if (aux_tiled) {
      float* ax = aux_tiled + tile * CVF_AUX_ROWS * CVF_TILE + lane;
#pragma unroll
      for (int i = 0; i < 9; ++i) ax[i * CVF_TILE] = ko.R[i];
      ax[9 * CVF_TILE] = (float)((double)P0.x + cr[0]);
      ax[10 * CVF_TILE] = (float)((double)P0.y + cr[1]);
      ax[11 * CVF_TILE] = (float)((double)pz + cr[2]);
#pragma unroll
      for (int i = 0; i < 6; ++i) ax[(12 + i) * CVF_TILE] = ko.Kinv[i];
    }
    if (it == 2) CVF_STAMP(5);
    // ---- features: aligned positions of atoms 0 .. n_rec-1, (x - pivot) - (centroid - pivot) in two fp32 steps
    float* ft = TILED ? feat_tiled + tile * pp.d_r * CVF_TILE + lane : nullptr;
    const f2 C0 = f2{cf[0], cf[1]}, C1 = f2{cf[2], cf[0]}, C2 = f2{cf[1], cf[2]};
    const f2 R01 = f2{ko.R[0], ko.R[1]}, R34 = f2{ko.R[3], ko.R[4]}, R67 = f2{ko.R[6], ko.R[7]};
    auto rot_xy = [&](float a0, float a1, float a2) { return fma2(splat2(a0), R01, fma2(splat2(a1), R34, splat2(a2) * R67)); };
    auto rot_z = [&](float a0, float a1, float a2) { return fmaf(a0, ko.R[2], fmaf(a1, ko.R[5], a2 * ko.R[8])); };
    const int npr = PAIR ? pp.n_rec >> 1 : 0;
#pragma unroll 2
    for (int m = 0; m < npr; ++m) {
      const f2 d0 = (my2[3 * m] - P0) - C0, d1 = (my2[3 * m + 1] - P1) - C1, d2 = (my2[3 * m + 2] - P2) - C2;
      const f2 axy = rot_xy(d0.x, d0.y, d1.x), bxy = rot_xy(d1.y, d2.x, d2.y);
      const float az = rot_z(d0.x, d0.y, d1.x), bz = rot_z(d1.y, d2.x, d2.y);
      if (TILED) {
        float* f = ft + 6 * m * CVF_TILE;
        f[0] = axy.x; f[CVF_TILE] = axy.y; f[2 * CVF_TILE] = az;
        f[3 * CVF_TILE] = bxy.x; f[4 * CVF_TILE] = bxy.y; f[5 * CVF_TILE] = bz;
      }
      if (ROWS) {
        my2[3 * m] = axy;
        my2[3 * m + 1] = f2{az, bxy.x};
        my2[3 * m + 2] = f2{bxy.y, bz};
      }
    }
#pragma unroll 2
    for (int a = 2 * npr; a < pp.n_rec; ++a) {
      const float d0 = (my[3 * a] - P0.x) - cf[0], d1 = (my[3 * a + 1] - P0.y) - cf[1], d2 = (my[3 * a + 2] - pz) - cf[2];
      const f2 axy = rot_xy(d0, d1, d2);
      const float az = rot_z(d0, d1, d2);
      if (TILED) {
        ft[3 * a * CVF_TILE] = axy.x; ft[(3 * a + 1) * CVF_TILE] = axy.y; ft[(3 * a + 2) * CVF_TILE] = az;
      }
      if (ROWS) {
        my[3 * a] = axy.x; my[3 * a + 1] = axy.y; my[3 * a + 2] = az;
      }
    }
    if (ROWS) {   // d_r == nc (host check): the tile of rows leaves as the coordinates came
      float4* dst = reinterpret_cast<float4*>(feat_rows + tile * CVF_TILE * nc);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int v = i * CVF_WAVE + lane;
        if (v < nvec) dst[v] = tile4[v];
      }
    }
    if (it == 2) CVF_STAMP(6);
  }
}

// ------------------------------------------------------------------------------------
// K1, fast layouts, small launches: FOUR lanes per frame (lane = 4 f + p; a wave covers 16 frames).
// With one lane per frame a 20 000-frame batch is 313 waves on a chip with 1024 SIMDs and each wave walks all
// N atoms three times: the kernel's time is one wave's serial latency.  Here lane p takes the atoms a = p (mod 4)
// in the covariance and feature loops and the four partial sums are combined by two quad-permute DPP steps;
// the 3x3 solve is done (redundantly) by all four lanes.  Same arithmetic per frame, a quarter of the
// per-wave chain, four times as many waves.  Chosen by the host for launches of at most kSplitMaxTiles tiles.
// ------------------------------------------------------------------------------------
constexpr int kQuadFrames = 16;

__device__ __forceinline__ double quad_sum(double v) {
  v += dpp_movd<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dpp_movd<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  return v;
}
__device__ __forceinline__ float quad_sumf(float v) {
  v += dpp_movf<0xB1, 0xf>(v);
  v += dpp_movf<0x4E, 0xf>(v);
  return v;
}

template <bool TILED, bool ROWS>
__global__ __launch_bounds__(64) void k1_align_quad_kernel(cvf_pp_desc pp, const float* __restrict__ x, int64_t B,
                                                            float* __restrict__ feat_tiled, float* __restrict__ feat_rows,
                                                            float* __restrict__ aux_tiled) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x, f = lane >> 2, p = lane & 3;
  const int nc = pp.n_coord, nal = pp.n_align;
  const int stride = x_tile_stride(nc);
  const int64_t f0 = (int64_t)blockIdx.x * kQuadFrames;
  float* refL = lds + kQuadFrames * stride;
  // ---- stage the 16 frames (one contiguous run of 16 nc floats) and the reference
  float rv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int j = lane + 64 * i;
    rv[i] = pp.ref_c[j < 3 * nal ? j : 3 * nal - 1];
  }
  load_x_tile<6>(x, B, nc, blockIdx.x, lds, lane, CVF_WAVE, kQuadFrames);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int j = lane + 64 * i;
    if (j < 3 * nal) refL[j] = rv[i];
  }
  for (int j = lane + 128; j < 3 * nal; j += 64) refL[j] = pp.ref_c[j];
  __syncthreads();
  const float* my = lds + f * stride;
  // ---- centroid and covariance: this lane's quarter of the align atoms, then the quad sums
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
  for (int i = 0; i < 15; ++i) acc[i] = quad_sum(acc[i]);
  const double inv = fast_rcp((double)nal);
  double cd[3] = {acc[0] * inv, acc[1] * inv, acc[2] * inv};
  double H[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) H[i][j] = fma(-cd[i], acc[12 + j], acc[3 + 3 * i + j]);
  KabschOut ko;
  kabsch_from_H(H, ko);
  const Centre c = centre_of(cd);
  // ---- outputs: the frame's column in its 64-frame tile
  const int64_t frame = f0 + f;
  const int64_t tile = f0 / CVF_TILE;
  const int col = (int)(f0 % CVF_TILE) + f;
  if (aux_tiled) {
    float* ax = aux_tiled + tile * CVF_AUX_ROWS * CVF_TILE + col;
    const float av[CVF_AUX_ROWS] = {ko.R[0], ko.R[1], ko.R[2], ko.R[3], ko.R[4], ko.R[5], ko.R[6], ko.R[7], ko.R[8],
                                    c.hi[0], c.hi[1], c.hi[2], ko.Kinv[0], ko.Kinv[1], ko.Kinv[2], ko.Kinv[3], ko.Kinv[4], ko.Kinv[5]};
#pragma unroll
    for (int i = 0; i < CVF_AUX_ROWS; ++i)
      if ((i & 3) == p) ax[i * CVF_TILE] = av[i];   // the four lanes of a frame share the 18 rows
  }
  float* ft = TILED ? feat_tiled + tile * pp.d_r * CVF_TILE + col : nullptr;
  float* fr = ROWS ? feat_rows + (frame < B ? frame : B - 1) * pp.d_r : nullptr;
#pragma unroll 2
  for (int a = p; a < pp.n_rec; a += 4) {
    const V3 al = row_times(centred(my, a, c), ko.R);
    if (TILED) {
      ft[(3 * a) * CVF_TILE] = al.x;
      ft[(3 * a + 1) * CVF_TILE] = al.y;
      ft[(3 * a + 2) * CVF_TILE] = al.z;
    }
    if (ROWS) {
      fr[3 * a] = al.x;
      fr[3 * a + 1] = al.y;
      fr[3 * a + 2] = al.z;
    }
  }
}

// identity preprocessing: features = coordinates; only the tiling changes
__global__ __launch_bounds__(64) void k1_identity_kernel(int d, const float* __restrict__ x, int64_t B,
                                                          float* __restrict__ feat_tiled, float* __restrict__ feat_rows) {
  const int lane = threadIdx.x;
  const int64_t tile = blockIdx.x;
  int64_t frame = tile * CVF_TILE + lane;
  const bool valid = frame < B;
  if (!valid) frame = B - 1;
  for (int j = 0; j < d; ++j) {
    const float v = x[frame * d + j];
    if (feat_tiled) feat_tiled[(tile * d + j) * CVF_TILE + lane] = v;
    if (feat_rows && valid) feat_rows[frame * d + j] = v;
  }
}

// ------------------------------------------------------------------------------------
// metric: per (tile, net)   G = J^T g ; E = sum_j a_j G_j^2 ; q = J (a .* G)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void metric_align_kernel(cvf_pp_desc pp, const float* __restrict__ x, int64_t B,
                                                           const float* __restrict__ aux_tiled,
                                                           const float* __restrict__ a, int k,
                                                           const float* __restrict__ g_tiled,
                                                           float* __restrict__ q_tiled, float* __restrict__ e_tiled) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  const int64_t tile = blockIdx.x;
  const int net = blockIdx.y;
  const int nc = pp.n_coord;
  const int stride = x_tile_stride(nc);
  const TablePrefetch tp = tables_prefetch(pp, lane);
  load_x_tile(x, B, nc, tile, lds, lane);
  float* Gl = lds + CVF_TILE * stride + lane;  // G(j) = Gl[j*64]
  for (int j = 0; j < nc; ++j) Gl[j * CVF_TILE] = 0.0f;
  const Tables tb = tables_commit(pp, tp, lds + CVF_TILE * (stride + nc), lane);
  float* aL = lds + CVF_TILE * (stride + nc) + tables_dwords(pp);   // diag_coeff, staged like the tables
  for (int i = lane; i < nc; i += CVF_WAVE) aL[i] = a[i];
  __syncthreads();
  const float* my = lds + lane * stride;
  const float* ax = aux_tiled + tile * CVF_AUX_ROWS * CVF_TILE + lane;
  float R[9], Kinv[6];
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = ax[i * CVF_TILE];
  const Centre c = centre_of(ax[9 * CVF_TILE], ax[10 * CVF_TILE], ax[11 * CVF_TILE]);
#pragma unroll
  for (int i = 0; i < 6; ++i) Kinv[i] = ax[(12 + i) * CVF_TILE];
  const int64_t base = (tile * k + net) * (int64_t)pp.d_r * CVF_TILE + lane;
  const float* gt = g_tiled + base;
  float* qt = q_tiled + base;
  auto addG = [&](int atm, V3 v) {
    Gl[(3 * atm) * CVF_TILE] += v.x;
    Gl[(3 * atm + 1) * CVF_TILE] += v.y;
    Gl[(3 * atm + 2) * CVF_TILE] += v.z;
  };
  auto getU = [&](int atm) {
    return V3{Gl[(3 * atm) * CVF_TILE], Gl[(3 * atm + 1) * CVF_TILE], Gl[(3 * atm + 2) * CVF_TILE]};
  };
  // ---- VJP: G = J^T g
  V3 sump = v3(0, 0, 0);
  float M[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int r = 0; r < pp.n_rec; ++r) {
    const Rec rc = load_rec(tb.rec, r);
    if (rc.type == CVF_FEAT_POSITION) {
      const V3 g = v3(gt[rc.out * CVF_TILE], gt[(rc.out + 1) * CVF_TILE], gt[(rc.out + 2) * CVF_TILE]);
      const V3 p = mat_times(R, g);
      addG(rc.a0, p);
      sump = sump + p;
      const V3 xc = centred(my, rc.a0, c);
      M[0] += xc.x * g.x; M[1] += xc.x * g.y; M[2] += xc.x * g.z;
      M[3] += xc.y * g.x; M[4] += xc.y * g.y; M[5] += xc.y * g.z;
      M[6] += xc.z * g.x; M[7] += xc.z * g.y; M[8] += xc.z * g.z;
    } else if (rc.type == CVF_FEAT_BOND) {
      const BondG e = bond_eval(atom(my, rc.a0), atom(my, rc.a1));
      const float gs = gt[rc.out * CVF_TILE];
      addG(rc.a0, gs * e.ga);
      addG(rc.a1, gs * e.gb);
    } else if (rc.type == CVF_FEAT_ANGLE) {
      const AngleG e = angle_eval(atom(my, rc.a0), atom(my, rc.a1), atom(my, rc.a2));
      float gs = gt[rc.out * CVF_TILE];
      if (pp.use_angle_value) gs = -gs / sqrtf(fmaxf(1.0f - e.cs * e.cs, 1e-30f));
      addG(rc.a0, gs * e.ga);
      addG(rc.a1, gs * e.gb);
      addG(rc.a2, gs * e.gc);
    } else {
      const DihedralG e = dihedral_eval(atom(my, rc.a0), atom(my, rc.a1), atom(my, rc.a2), atom(my, rc.a3));
      const float gs = pp.use_angle_value ? gt[rc.out * CVF_TILE]
                                          : (gt[(rc.out + 1) * CVF_TILE] * e.cs - gt[rc.out * CVF_TILE] * e.sn);
      addG(rc.a0, gs * e.g1);
      addG(rc.a1, gs * e.g2);
      addG(rc.a2, gs * e.g3);
      addG(rc.a3, gs * e.g4);
    }
  }
  const float inv_nal = 1.0f / (float)pp.n_align;
  if (pp.has_position) {
    // s = Kinv ax(R^T M);  Z = R [s]x;  G_b += Z ref_b - sum_p / n_align   (b over align atoms)
    float T[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) T[3 * i + j] = R[i] * M[j] + R[3 + i] * M[3 + j] + R[6 + i] * M[6 + j];
    const V3 s = sym_times(Kinv, v3(T[7] - T[5], T[2] - T[6], T[3] - T[1]));
    // [s]x rows: (0,-sz,sy), (sz,0,-sx), (-sy,sx,0)
    float Z[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      Z[3 * i + 0] = R[3 * i + 1] * s.z - R[3 * i + 2] * s.y;
      Z[3 * i + 1] = -R[3 * i + 0] * s.z + R[3 * i + 2] * s.x;
      Z[3 * i + 2] = R[3 * i + 0] * s.y - R[3 * i + 1] * s.x;
    }
    // (weighted alignment: atom b's share of the centroid is align_w[b] / n_align, and ref_c holds align_w[b] * ref_b)
    const V3 shift = inv_nal * sump;
    for (int b = 0; b < pp.n_align; ++b) {
      const V3 rf = v3(tb.ref_c[3 * b], tb.ref_c[3 * b + 1], tb.ref_c[3 * b + 2]);
      const float wb = pp.align_w ? pp.align_w[b] : 1.0f;
      addG(tb.align_idx[b], mat_times(Z, rf) - wb * shift);
    }
  }
  // ---- E = sum a G^2 ; u = a .* G (in place)
  float E = 0.0f;
#pragma unroll 6
  for (int j = 0; j < nc; ++j) {
    const float Gj = Gl[j * CVF_TILE];
    const float uj = aL[j] * Gj;
    E += uj * Gj;
    Gl[j * CVF_TILE] = uj;
  }
  e_tiled[(tile * k + net) * CVF_TILE + lane] = E;
  // ---- JVP: q = J u
  V3 ubar = v3(0, 0, 0);
  float dR[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (pp.has_position) {
    for (int b = 0; b < pp.n_align; ++b) ubar = ubar + (pp.align_w ? pp.align_w[b] : 1.0f) * getU(tb.align_idx[b]);
    ubar = inv_nal * ubar;
    float dH[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < pp.n_align; ++b) {
      const V3 u = getU(tb.align_idx[b]) - ubar;
      const V3 rf = v3(tb.ref_c[3 * b], tb.ref_c[3 * b + 1], tb.ref_c[3 * b + 2]);
      dH[0] += u.x * rf.x; dH[1] += u.x * rf.y; dH[2] += u.x * rf.z;
      dH[3] += u.y * rf.x; dH[4] += u.y * rf.y; dH[5] += u.y * rf.z;
      dH[6] += u.z * rf.x; dH[7] += u.z * rf.y; dH[8] += u.z * rf.z;
    }
    float T[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) T[3 * i + j] = R[i] * dH[j] + R[3 + i] * dH[3 + j] + R[6 + i] * dH[6 + j];
    const V3 w = sym_times(Kinv, v3(T[7] - T[5], T[2] - T[6], T[3] - T[1]));
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      dR[3 * i + 0] = R[3 * i + 1] * w.z - R[3 * i + 2] * w.y;
      dR[3 * i + 1] = -R[3 * i + 0] * w.z + R[3 * i + 2] * w.x;
      dR[3 * i + 2] = R[3 * i + 0] * w.y - R[3 * i + 1] * w.x;
    }
  }
  for (int r = 0; r < pp.n_rec; ++r) {
    const Rec rc = load_rec(tb.rec, r);
    if (rc.type == CVF_FEAT_POSITION) {
      const V3 qa = row_times(getU(rc.a0) - ubar, R) + row_times(centred(my, rc.a0, c), dR);
      qt[rc.out * CVF_TILE] = qa.x;
      qt[(rc.out + 1) * CVF_TILE] = qa.y;
      qt[(rc.out + 2) * CVF_TILE] = qa.z;
    } else if (rc.type == CVF_FEAT_BOND) {
      const BondG e = bond_eval(atom(my, rc.a0), atom(my, rc.a1));
      qt[rc.out * CVF_TILE] = dot(e.ga, getU(rc.a0)) + dot(e.gb, getU(rc.a1));
    } else if (rc.type == CVF_FEAT_ANGLE) {
      const AngleG e = angle_eval(atom(my, rc.a0), atom(my, rc.a1), atom(my, rc.a2));
      float dv = dot(e.ga, getU(rc.a0)) + dot(e.gb, getU(rc.a1)) + dot(e.gc, getU(rc.a2));
      if (pp.use_angle_value) dv = -dv / sqrtf(fmaxf(1.0f - e.cs * e.cs, 1e-30f));
      qt[rc.out * CVF_TILE] = dv;
    } else {
      const DihedralG e = dihedral_eval(atom(my, rc.a0), atom(my, rc.a1), atom(my, rc.a2), atom(my, rc.a3));
      const float dphi = dot(e.g1, getU(rc.a0)) + dot(e.g2, getU(rc.a1)) + dot(e.g3, getU(rc.a2)) + dot(e.g4, getU(rc.a3));
      if (pp.use_angle_value) {
        qt[rc.out * CVF_TILE] = dphi;
      } else {
        qt[rc.out * CVF_TILE] = -e.sn * dphi;
        qt[(rc.out + 1) * CVF_TILE] = e.cs * dphi;
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// (A four-lanes-per-frame variant of this kernel, as for K1, was built and measured: same 20 us.  At this batch the
// kernel is bound by the g read / q write phases of 16 MB each, which all waves enter together, not by the
// per-wave chain.)
// metric, fast path (CVF_PP_ALIGN_CONTIG | CVF_PP_PURE_POSITION): feature 3a..3a+2 = aligned position of atom a,
// align atom b = atom b.  Block = one 64-frame tile x `wpb` nets (one wave per net): the waves stage the
// coordinate tile together, once, and each keeps its own [3N][64] image U in LDS that first holds g, then
// u = a .* G.  Three passes, none with a read-modify-write of LDS or a re-read of global memory:
//   1  g -> U, sum_a R g_a, M = sum_a xc_a (x) g_a            (g prefetched one chunk of atoms ahead)
//   2  G_a = R g_a [+ Z ref_a - shift], E, u_a -> U, sums for the rotation's tangent   (LDS only)
//   3  q_a = (u_a - ubar) R + xc_a dR                          (LDS -> global)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void metric_pure_kernel(cvf_pp_desc pp, const float* __restrict__ x, int64_t B,
                                                           const float* __restrict__ aux_tiled,
                                                           const float* __restrict__ a, int k,
                                                           const float* __restrict__ g_tiled,
                                                           float* __restrict__ q_tiled, float* __restrict__ e_tiled,
                                                           MetricFuse fuse) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x;
  const int64_t tile = blockIdx.x;
  const int net = blockIdx.y * (nthreads >> 6) + wave;   // the host launches wpb | k, so net < k
  const int nc = pp.n_coord, N = pp.n_rec, nal = pp.n_align;
  const int stride = x_tile_stride(nc);
  float* refL = lds + CVF_TILE * stride;                              // [3*nal]
  float* aL = refL + 3 * nal;                                         // [nc]
  float* Ul = aL + nc + (size_t)wave * nc * CVF_TILE + lane;          // this wave's image: U(j) = Ul[j*64]
  CVF_STAMP(8);
  // everything the passes need from global memory besides g is requested before the tile is staged
  const float* ax = aux_tiled + tile * CVF_AUX_ROWS * CVF_TILE + lane;
  float auxv[CVF_AUX_ROWS];
#pragma unroll
  for (int i = 0; i < CVF_AUX_ROWS; ++i) auxv[i] = ax[i * CVF_TILE];
  float wv = 0.0f, yv[CVF_MAX_NETS];
  if (fuse.on) {
    const int64_t frame = tile * CVF_TILE + lane;
    wv = fuse.w[frame < B ? frame : B - 1];
    if (frame >= B) wv = 0.0f;
#pragma unroll
    for (int j = 0; j < CVF_MAX_NETS; ++j) yv[j] = fuse.y_tiled[(tile * k + (j < k ? j : k - 1)) * CVF_TILE + lane];
  }
  const int ntab = 3 * nal + nc;
  float tabv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int j = tid + nthreads * i;
    const int jc = j < ntab ? j : ntab - 1;
    tabv[i] = jc < 3 * nal ? pp.ref_c[jc] : a[jc - 3 * nal];
  }
  // ... including the first chunk of g
  const int N0 = pp.n_rec;
  const int64_t base = (tile * k + net) * (int64_t)pp.d_r * CVF_TILE + lane;
  const float* gt = g_tiled + base;
  float* qt = q_tiled + base;
  // atoms past N (tail of the last chunk) are clamped to atom N-1 and masked out of the sums: no branch per atom,
  // so the LDS reads and the arithmetic of a chunk's eight atoms interleave
  float cur[3 * kGChunk], nxt[3 * kGChunk];
#pragma unroll
  for (int i = 0; i < kGChunk; ++i) {
    const int at = i < N0 ? i : N0 - 1;
#pragma unroll
    for (int d = 0; d < 3; ++d) cur[3 * i + d] = gt[(3 * at + d) * CVF_TILE];
  }
  load_x_tile(x, B, nc, tile, lds, tid, nthreads);
  CVF_STAMP(9);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int j = tid + nthreads * i;
    if (j < ntab) refL[j] = tabv[i];
  }
  for (int j = tid + 2 * nthreads; j < ntab; j += nthreads) refL[j] = j < 3 * nal ? pp.ref_c[j] : a[j - 3 * nal];
  __syncthreads();
  metric_pure_passes<false>(pp, lane, net, k, tile, B, lds + lane * stride, refL, aL, Ul, auxv, gt, qt, e_tiled, fuse, wv, yv, cur);
}

__global__ __launch_bounds__(64) void metric_identity_kernel(int d, const float* __restrict__ a, int k,
                                                              const float* __restrict__ g_tiled,
                                                              float* __restrict__ q_tiled, float* __restrict__ e_tiled) {
  const int lane = threadIdx.x;
  const int64_t tile = blockIdx.x;
  const int net = blockIdx.y;
  const int64_t base = (tile * k + net) * (int64_t)d * CVF_TILE + lane;
  float E = 0.0f;
  for (int j = 0; j < d; ++j) {
    const float g = g_tiled[base + j * CVF_TILE];
    const float u = a[j] * g;
    E += u * g;
    q_tiled[base + j * CVF_TILE] = u;
  }
  e_tiled[(tile * k + net) * CVF_TILE + lane] = E;
}

}  // namespace

int cvf_k1_large_launch(const cvf_pp_desc* pp, const float* x, int64_t B, float* feat_tiled, float* feat_rows,
                        float* aux_tiled, float* slot_xyz, hipStream_t s);
size_t cvf_k1_large_scratch_bytes(const cvf_pp_desc* pp, int64_t B);
size_t cvf_metric_large_lds(const cvf_pp_desc* pp);
int cvf_ef_stats_finish(const cvf_ef_cfg* cfg, int n_rows, const double* partial, double* stats, double* loss_vec, double* coef,
                        hipStream_t s);
int cvf_metric_large_launch(const cvf_pp_desc* pp, int64_t B, const float* aux_tiled, const float* a, int k,
                            const float* slot_xyz, const double* dense, const float* g_tiled, float* q_tiled, float* e_tiled,
                            const MetricFuse* fuse, int* fused_rows, hipStream_t s);
int cvf_ef_stats_finish_impl(const cvf_ef_cfg* cfg, int n_rows, int stat_major, const double* partial, double* stats,
                             double* loss_vec, double* coef, hipStream_t s);

// frames larger than this use the streaming workgroup-per-frame kernel (k1_large.hip)
static constexpr int kLanePerFrameMaxCoord = 192;
// launches of at most this many 64-frame tiles use the four-lanes-per-frame kernels (fast layouts)
static constexpr int kSplitMaxTiles = 1024;

extern "C" int64_t cvf_align_feature_scratch_bytes(const cvf_pp_desc* pp, int64_t B) {
  if (!pp || pp->mode != CVF_PP_ALIGN || pp->n_coord <= kLanePerFrameMaxCoord) return 0;
  return (int64_t)cvf_k1_large_scratch_bytes(pp, B);
}

extern "C" int cvf_align_feature_fwd(const cvf_pp_desc* pp, const float* x, int64_t B, float* feat_tiled,
                                     float* feat_rows, float* aux_tiled, void* scratch, void* stream) {
  CVF_REQUIRE(pp && x && B > 0, "cvf_align_feature_fwd: null argument or empty batch (B=%lld)", (long long)B);
  CVF_REQUIRE(feat_tiled || feat_rows, "cvf_align_feature_fwd: no output buffer");
  const int64_t T = cvf_ntiles(B);
  hipStream_t s = (hipStream_t)stream;
  if (pp->mode == CVF_PP_IDENTITY) {
    CVF_REQUIRE(pp->d_r == pp->n_coord, "identity preprocessing needs d_r == n_coord");
    hipLaunchKernelGGL(k1_identity_kernel, dim3((unsigned)T), dim3(64), 0, s, pp->n_coord, x, B, feat_tiled, feat_rows);
    return cvf_check_launch("k1_identity_kernel");
  }
  CVF_REQUIRE(pp->mode == CVF_PP_ALIGN, "unknown pp mode %d", pp->mode);
  CVF_REQUIRE(pp->n_coord % 3 == 0 && pp->n_align >= 3 && pp->align_idx && pp->ref_c && pp->rec,
              "cvf_align_feature_fwd: malformed descriptor (n_coord=%d n_align=%d)", pp->n_coord, pp->n_align);
  CVF_REQUIRE(!pp->align_w || (pp->flags == 0 && pp->n_coord <= kLanePerFrameMaxCoord),
              "cvf_align_feature_fwd: per-atom alignment weights need flags == 0 and at most %d coordinates per frame",
              kLanePerFrameMaxCoord);
  if (pp->n_coord > kLanePerFrameMaxCoord) {
    return cvf_k1_large_launch(pp, x, B, feat_tiled, feat_rows, aux_tiled, (float*)scratch, s);
  }
  const size_t lds = ((size_t)CVF_TILE * x_tile_stride(pp->n_coord) + tables_dwords(*pp)) * sizeof(float);
  CVF_REQUIRE(lds <= 160 * 1024, "frames of %d coordinates do not fit the lane-per-frame kernel's LDS tile", pp->n_coord);
  const bool fast = (pp->flags & (CVF_PP_ALIGN_CONTIG | CVF_PP_PURE_POSITION)) == (CVF_PP_ALIGN_CONTIG | CVF_PP_PURE_POSITION);
  if (fast)
    CVF_REQUIRE(pp->d_r == 3 * pp->n_rec && 3 * pp->n_rec <= pp->n_coord && pp->n_align * 3 <= pp->n_coord,
                "cvf_align_feature_fwd: flags do not match the descriptor");
  // small launches of the fast layout: four lanes per frame (latency), otherwise one lane per frame (throughput)
  static const int split_env = [] { const char* e = getenv("CVF_K1_QUAD"); return e ? atoi(e) : -1; }();
  const bool quad = fast && (split_env >= 0 ? split_env != 0 : T <= kSplitMaxTiles);
  if (quad) {
    const size_t ldsq = ((size_t)kQuadFrames * x_tile_stride(pp->n_coord) + 3 * (size_t)pp->n_align) * sizeof(float);
    const unsigned nb = (unsigned)((T * CVF_TILE) / kQuadFrames);   // whole tiles: padded frames replicate the last one
    auto launchq = [&](auto kernel) {
      if (ldsq > 48 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsq);
      hipLaunchKernelGGL(kernel, dim3(nb), dim3(64), ldsq, s, *pp, x, B, feat_tiled, feat_rows, aux_tiled);
    };
    if (feat_tiled && feat_rows) launchq(k1_align_quad_kernel<true, true>);
    else if (feat_tiled) launchq(k1_align_quad_kernel<true, false>);
    else launchq(k1_align_quad_kernel<false, true>);
    return cvf_check_launch("k1_align_quad_kernel");
  }
  // large launches of the fast layout whose tile can sit in LDS as it lies in memory: persistent streaming waves
  // over the whole tiles, the remainder (B mod 64 frames) below
  static const bool no_stream = getenv("CVF_K1_NOSTREAM") != nullptr;
  const int nc = pp->n_coord;
  const int64_t T_full = B / CVF_TILE;
  if (fast && !no_stream && (nc % 4 == 2 || nc % 2 == 1) && nc <= 4 * 26 && T_full > kSplitMaxTiles && (!feat_rows || pp->d_r == nc)) {
    const size_t ldss = (size_t)CVF_TILE * nc * sizeof(float) + (size_t)pp->n_align * sizeof(float4);
    auto launchs = [&](auto kernel) {
      if (ldss > 48 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldss);
      int per_cu = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 64, ldss) != hipSuccess || per_cu < 1) per_cu = 4;
      const int64_t resident = (int64_t)per_cu * cvf_cu_count();
      hipLaunchKernelGGL(kernel, dim3((unsigned)(T_full < resident ? T_full : resident)), dim3(64), ldss, s, *pp, x, T_full,
                         feat_tiled, feat_rows, aux_tiled);
    };
    const int nv = (nc + 3) / 4;
#define CVF_K1_STREAM2(NV, PAIR)                                                         \
    do {                                                                                 \
      if (feat_tiled && feat_rows) launchs(k1_stream_kernel<NV, true, true, PAIR>);      \
      else if (feat_tiled) launchs(k1_stream_kernel<NV, true, false, PAIR>);             \
      else launchs(k1_stream_kernel<NV, false, true, PAIR>);                             \
    } while (0)
#define CVF_K1_STREAM(NV)                                                                \
    do {                                                                                 \
      if (nc % 2 == 0) CVF_K1_STREAM2(NV, true);                                         \
      else CVF_K1_STREAM2(NV, false);                                                    \
    } while (0)
    if (nv <= 8) CVF_K1_STREAM(8);
    else if (nv <= 17) CVF_K1_STREAM(17);
    else CVF_K1_STREAM(26);
#undef CVF_K1_STREAM
#undef CVF_K1_STREAM2
    const int rc = cvf_check_launch("k1_stream_kernel");
    const int64_t done = T_full * CVF_TILE;
    if (rc || done == B) return rc;
    return cvf_align_feature_fwd(pp, x + done * nc, B - done, feat_tiled ? feat_tiled + T_full * pp->d_r * CVF_TILE : nullptr,
                                 feat_rows ? feat_rows + done * pp->d_r : nullptr,
                                 aux_tiled ? aux_tiled + T_full * CVF_AUX_ROWS * CVF_TILE : nullptr, scratch, stream);
  }
  auto launch = [&](auto kernel) {
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kernel, dim3((unsigned)T), dim3(64), lds, s, *pp, x, B, feat_tiled, feat_rows, aux_tiled);
  };
  const int variant = (fast ? 4 : 0) | (feat_tiled ? 2 : 0) | (feat_rows ? 1 : 0);
  switch (variant) {
    case 1: launch(k1_align_kernel<false, false, true>); break;
    case 2: launch(k1_align_kernel<false, true, false>); break;
    case 3: launch(k1_align_kernel<false, true, true>); break;
    case 5: launch(k1_align_kernel<true, false, true>); break;
    case 6: launch(k1_align_kernel<true, true, false>); break;
    default: launch(k1_align_kernel<true, true, true>); break;
  }
  return cvf_check_launch("k1_align_kernel");
}

static int metric_apply_impl(const cvf_pp_desc* pp, const float* x, int64_t B, const float* aux_tiled, const float* a,
                             int k, const float* g_tiled, float* q_tiled, float* e_tiled, const float* slot_xyz,
                             const double* dense, const MetricFuse& fuse, bool* fused, void* stream, int* major_rows = nullptr) {
  if (fused) *fused = false;
  if (major_rows) *major_rows = 0;   // > 0: the launch left that many rows of [statistic][row] partial sums (large molecules)
  CVF_REQUIRE(pp && a && g_tiled && q_tiled && e_tiled && B > 0 && k >= 1 && k <= CVF_MAX_NETS,
              "cvf_metric_apply: bad argument (B=%lld k=%d)", (long long)B, k);
  const int64_t T = cvf_ntiles(B);
  hipStream_t s = (hipStream_t)stream;
  if (pp->mode == CVF_PP_IDENTITY) {
    hipLaunchKernelGGL(metric_identity_kernel, dim3((unsigned)T, k), dim3(64), 0, s, pp->n_coord, a, k, g_tiled, q_tiled,
                       e_tiled);
    return cvf_check_launch("metric_identity_kernel");
  }
  CVF_REQUIRE(aux_tiled, "cvf_metric_apply: align mode needs aux");
  CVF_REQUIRE(!pp->align_w || (pp->flags == 0 && pp->n_coord <= kLanePerFrameMaxCoord),
              "cvf_metric_apply: per-atom alignment weights need flags == 0 and at most %d coordinates per frame",
              kLanePerFrameMaxCoord);
  if (pp->n_coord > kLanePerFrameMaxCoord) {
    CVF_REQUIRE(slot_xyz && dense && pp->rec_slot && pp->slot_atom && pp->atom_align && pp->n_slot > 0,
                "cvf_metric_apply: large molecules need the slot tables, slot_xyz (cvf_align_feature_fwd scratch) and "
                "dense (cvf_metric_dense_tensors)");
    return cvf_metric_large_launch(pp, B, aux_tiled, a, k, slot_xyz, dense, g_tiled, q_tiled, e_tiled, &fuse, major_rows, s);
  }
  CVF_REQUIRE(x, "cvf_metric_apply: align mode needs x");
  const size_t lds = ((size_t)CVF_TILE * (x_tile_stride(pp->n_coord) + pp->n_coord) + tables_dwords(*pp) + pp->n_coord) * sizeof(float);
  CVF_REQUIRE(lds <= 160 * 1024, "frames of %d coordinates do not fit the lane-per-frame metric kernel's LDS", pp->n_coord);
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)metric_align_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  // the three-pass kernel additionally wants every align atom to be a feature atom (atoms 0..n_align-1)
  const bool fast = (pp->flags & (CVF_PP_ALIGN_CONTIG | CVF_PP_PURE_POSITION)) == (CVF_PP_ALIGN_CONTIG | CVF_PP_PURE_POSITION) &&
                    pp->n_align <= pp->n_rec;
  if (fast) {
    CVF_REQUIRE(pp->d_r == 3 * pp->n_rec && 3 * pp->n_rec <= pp->n_coord, "cvf_metric_apply: flags do not match the descriptor");
    // all k nets of a tile in one block (one staged coordinate tile) when their images fit, else one net per block
    auto lds_for = [&](int wpb) {
      return ((size_t)CVF_TILE * x_tile_stride(pp->n_coord) + 3 * (size_t)pp->n_align + pp->n_coord +
              (size_t)wpb * pp->n_coord * CVF_TILE) * sizeof(float);
    };
    const int wpb = lds_for(k) <= 80 * 1024 ? k : 1;
    const size_t lds_p = lds_for(wpb);
    CVF_REQUIRE(lds_p <= 160 * 1024, "frames of %d coordinates do not fit the lane-per-frame metric kernel's LDS", pp->n_coord);
    if (lds_p > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)metric_pure_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p);
    MetricFuse f = fuse;
    if (T > kFuseMaxTiles) f.on = 0;
    hipLaunchKernelGGL(metric_pure_kernel, dim3((unsigned)T, k / wpb), dim3(64 * wpb), lds_p, s, *pp, x, B, aux_tiled, a, k,
                       g_tiled, q_tiled, e_tiled, f);
    if (fused) *fused = f.on != 0;
    return cvf_check_launch("metric_pure_kernel");
  }
  hipLaunchKernelGGL(metric_align_kernel, dim3((unsigned)T, k), dim3(64), lds, s, *pp, x, B, aux_tiled, a, k, g_tiled,
                     q_tiled, e_tiled);
  return cvf_check_launch("metric_align_kernel");
}

extern "C" int cvf_metric_apply(const cvf_pp_desc* pp, const float* x, int64_t B, const float* aux_tiled,
                                const float* a, int k, const float* g_tiled, float* q_tiled, float* e_tiled,
                                const float* slot_xyz, const double* dense, void* stream) {
  MetricFuse none = {};
  return metric_apply_impl(pp, x, B, aux_tiled, a, k, g_tiled, q_tiled, e_tiled, slot_xyz, dense, none, nullptr, stream);
}

extern "C" int64_t cvf_metric_stats_scratch_doubles(int64_t B, int k) {
  // (up to two rows per tile; the large-molecule derivative kernel: one row per group of 4..16 frames, while those are few)
  const int64_t T = cvf_ntiles(B), per_group = 16 * T < kFuseMaxTiles ? 16 * T : kFuseMaxTiles;
  return (2 * T > per_group ? 2 * T : per_group) * (int64_t)cvf_ef_nstats(k, 0) + cvf_ef_stats_scratch_doubles(k, 0);
}

extern "C" int cvf_metric_apply_stats(const cvf_pp_desc* pp, const float* x, int64_t B, const float* aux_tiled,
                                      const float* a, int k, const float* g_tiled, float* q_tiled, float* e_tiled,
                                      const float* slot_xyz, const double* dense, const cvf_ef_cfg* cfg, const float* w,
                                      const float* y_tiled, double* scratch, double* stats, double* loss_vec, double* coef,
                                      void* stream) {
  CVF_REQUIRE(cfg && w && y_tiled && scratch && stats, "cvf_metric_apply_stats: bad argument");
  CVF_REQUIRE(cfg->k == k && cfg->lag_idx == 0, "cvf_metric_apply_stats: generator mode only, cfg.k must equal k");
  CVF_REQUIRE(loss_vec == nullptr || coef != nullptr, "cvf_metric_apply_stats: loss_vec without coef");
  const int ns = cvf_ef_nstats(k, 0);
  const int64_t T = cvf_ntiles(B);
  MetricFuse f = {};
  f.on = 1;
  f.ns = ns;
  f.w = w;
  f.y_tiled = y_tiled;
  f.partial = scratch;
  bool fused = false;
  int major_rows = 0;
  const int rc = metric_apply_impl(pp, x, B, aux_tiled, a, k, g_tiled, q_tiled, e_tiled, slot_xyz, dense, f, &fused, stream, &major_rows);
  if (rc) return rc;
  if (major_rows > 0) return cvf_ef_stats_finish_impl(cfg, major_rows, 1, scratch, stats, loss_vec, coef, (hipStream_t)stream);
  if (fused) return cvf_ef_stats_finish(cfg, (int)T, scratch, stats, loss_vec, coef, (hipStream_t)stream);
  // shapes without the fused first stage: the two-stage reduction on the rest of the scratch
  return cvf_ef_stats(cfg, B, w, y_tiled, e_tiled, nullptr, nullptr, scratch + T * ns, stats, loss_vec, coef, stream);
}
