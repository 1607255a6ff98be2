// Packed MFMA weight fragments of the eigenfunction nets (shared by the net kernels and the optimiser).
#pragma once
#include "cvf_common.hpp"

// ------------------------------------------------------------------------------------------------
// Packed weights.  The MFMA A-operand fragments of every layer are kept in a second buffer in
// exactly the order the waves consume them (fragment (step, row-tile) = 64 consecutive floats, one
// per lane), so a fragment load is one coalesced 256-byte read instead of a 16-row gather.  The
// buffer is refreshed by the optimiser kernel itself (each parameter is scattered to its <= 2
// fragment slots when it is updated) and by cvf_ef_pack after the parameters were set from outside.
// Per net: [F0: S1*RT][Fh_l: NG*RT, l=1..NH-1][Th_l: NG*RT, l=1..NH-1][T0: CT*NG] fragments.
// ------------------------------------------------------------------------------------------------
struct PackLayout {
  int NG, RT, S1, CT, NH, per_net;
  __host__ __device__ int f0() const { return 0; }
  __host__ __device__ int fh(int l) const { return (S1 * RT + (l - 1) * NG * RT) * 64; }
  __host__ __device__ int th(int l) const { return (S1 * RT + (NH - 1 + l - 1) * NG * RT) * 64; }
  __host__ __device__ int t0() const { return (S1 * RT + 2 * (NH - 1) * NG * RT) * 64; }
};
__host__ __device__ inline PackLayout pack_layout(int H, int NH, int D) {
  PackLayout L;
  L.NG = (H + 3) / 4;
  L.RT = (L.NG + 3) / 4;
  L.S1 = (D + 3) / 4;
  L.CT = (D + 15) / 16;
  L.NH = NH;
  L.per_net = (L.S1 * L.RT + 2 * (NH - 1) * L.NG * L.RT + L.CT * L.NG) * 64;
  return L;
}
// lane and row-tile of hidden-order row feature o
__host__ __device__ inline void hid_row_slot(int o, int& rt, int& rho) {
  const int g = o >> 2, qq = o & 3;
  rt = g >> 2;
  rho = 4 * qq + (g & 3);
}
// The nets' offset table for pack_scatter, kept in LDS: indexed with a run-time (net, layer), the by-value kernel argument
// turns into a private (scratch-memory) copy and ~700 cycles per access - the serial part of the reduce + Adam kernel.
struct PackTab {
  int n_nets, NH, H, D;
  int w[CVF_MAX_NETS][CVF_MAX_LAYERS];
  int bend[CVF_MAX_NETS];   // last parameter of net n
};
// one thread, constant indices only (scalar loads from the kernel arguments); callers follow with a barrier
__device__ __forceinline__ void pack_tab_fill(PackTab& t, const cvf_mlp_desc& mlp) {
  t.n_nets = mlp.n_nets;
  t.NH = mlp.n_layers - 1;
  t.H = mlp.dims[1];
  t.D = mlp.dims[0];
#pragma unroll
  for (int n = 0; n < CVF_MAX_NETS; ++n) {
#pragma unroll
    for (int l = 0; l < CVF_MAX_LAYERS; ++l) t.w[n][l] = mlp.w_off[n][l];
    int be = 0;
#pragma unroll
    for (int l = 0; l < CVF_MAX_LAYERS; ++l) be = (l == mlp.n_layers - 1) ? mlp.b_off[n][l] : be;
    t.bend[n] = be;
  }
}
// scatter parameter p (new value v) of the flat buffer into its fragment slots
__device__ __forceinline__ void pack_scatter(const PackTab& t, int p, float v, float* __restrict__ packed) {
  const int NH = t.NH, H = t.H, D = t.D;
  const PackLayout L = pack_layout(H, NH, D);
  for (int n = 0; n < t.n_nets; ++n) {
    if (p < t.w[n][0] || p > t.bend[n]) continue;
    float* base = packed + (int64_t)n * L.per_net;
    {
      const int rel = p - t.w[n][0];
      if (rel >= 0 && rel < H * D) {
        const int o = rel / D, i = rel - o * D;
        int rt, rho;
        hid_row_slot(o, rt, rho);
        base[L.f0() + ((i >> 2) * L.RT + rt) * 64 + (i & 3) * 16 + rho] = v;
        base[L.t0() + ((i >> 4) * L.NG + (o >> 2)) * 64 + (o & 3) * 16 + (i & 15)] = v;
        return;
      }
    }
    for (int l = 1; l < NH; ++l) {
      const int rel = p - t.w[n][l];
      if (rel >= 0 && rel < H * H) {
        const int o = rel / H, i = rel - o * H;
        int rt, rho;
        hid_row_slot(o, rt, rho);
        base[L.fh(l) + ((i >> 2) * L.RT + rt) * 64 + (i & 3) * 16 + rho] = v;
        hid_row_slot(i, rt, rho);
        base[L.th(l) + ((o >> 2) * L.RT + rt) * 64 + (o & 3) * 16 + rho] = v;
        return;
      }
    }
    return;
  }
}

// feature held by register r of row-tile rt in lane-group q
__device__ __forceinline__ int hid_feature(int rt, int r, int q) { return 4 * (4 * rt + r) + q; }
// feature computed by A-operand row rho (= lane&15) of row-tile rt
__device__ __forceinline__ int hid_row_feature(int rt, int rho) { return 4 * (4 * rt + (rho & 3)) + (rho >> 2); }

