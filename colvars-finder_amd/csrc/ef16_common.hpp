// Shared by ef16_front.hip and ef16_back.hip (the 16-frames-per-wave step, split in two translation units so that they compile
// side by side): constants of the unit decomposition, the front kernel's LDS layout, the net shapes covered and their dispatch.
#pragma once
#include "cvf_metric.hpp"
#include "ef_frag.hpp"
#include <stdlib.h>
#include <type_traits>


namespace {

constexpr int kU = 16;       // frames per unit
constexpr int kImgP = 76;    // pitch of the [frame][feature] images: = 12 (mod 32), so the four-lanes-per-frame reads (address
                             // 76 f + 3 p + c) and the matrix cores' 16-byte row writes (76 col + 4 q) touch every bank once
constexpr int kAuxP = 21;    // pitch of the per-frame alignment record: R (9), centroid hi (3), K^-1 (6), centroid lo (3)
constexpr int kMaxRows16 = 16384;   // units whose rows of batch sums one finishing launch adds (above: cvf_ef_stats)
template <int NH>
__host__ __device__ constexpr int kHand() { return 2 * NH; }   // vectors of the front -> back hand-off per (tile, net)

__device__ __forceinline__ float quad_sumf16(float v) {
  v += dpp_movf<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dpp_movf<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  return v;
}
__device__ __forceinline__ double quad_sumd16(double v) {
  v += dpp_movd<0xB1, 0xf>(v);
  v += dpp_movd<0x4E, 0xf>(v);
  return v;
}

struct Front16Lds {   // offsets in floats
  int ref, a, aux, w, rs, y, e, feat, g, total;
};
__host__ __device__ inline Front16Lds front16_lds(int nc, int nal, int k) {
  Front16Lds L;
  const int stride = x_tile_stride(nc);
  L.ref = kU * stride;
  L.a = L.ref + 3 * nal;
  L.aux = (L.a + nc + 3) & ~3;
  L.w = L.aux + ((kU * kAuxP + 3) & ~3);
  L.rs = L.w + kU;      // sum of the (centred) reference over the align atoms: 3 floats (+ 1 pad)
  L.y = L.rs + 4;
  L.e = L.y + k * kU;
  L.feat = L.e + k * kU;              // 16-byte aligned: every term above is a multiple of 4 floats
  L.g = L.feat + kU * kImgP;
  L.total = L.g + k * kU * kImgP;
  return L;
}

// ------------------------------------------------------------------------------------------------------------------
// front
// ------------------------------------------------------------------------------------------------------------------
// NIT = ceil(N / 4) exactly (atoms per lane in the four-lanes-per-frame passes): every iteration but the last is complete, so
// only the last one carries the masks of the ragged end.  ALLAL: every feature atom is an align atom (n_align == n_rec).
bool ef16_shape(const cvf_mlp_desc* m, int* H, int* NH) {
  if (m->n_layers < 2 || m->n_layers > 4 || m->dims[m->n_layers] != 1) return false;
  *H = m->dims[1];
  *NH = m->n_layers - 1;
  for (int l = 1; l < m->n_layers; ++l)
    if (m->dims[l] != *H) return false;
  for (int l = 0; l < m->n_layers; ++l)
    if (m->act[l] != (l + 1 < m->n_layers ? 1 : 0)) return false;
  return true;
}

template <class F>
bool ef16_dispatch(int H, int NH, F&& f) {
#define EF_CASE(H_, NH_)                                                        \
  if (H == H_ && NH == NH_) {                                                   \
    f(std::integral_constant<int, H_>{}, std::integral_constant<int, NH_>{});   \
    return true;                                                                \
  }
#ifdef CVF_DEV_SHAPES   // developer builds (tools/*.hip probes, -S listings): the config-3 instance only - seconds instead of minutes
  EF_CASE(20, 3)
#else
  EF_CASE(8, 1) EF_CASE(8, 2) EF_CASE(8, 3)
  EF_CASE(12, 1) EF_CASE(12, 2) EF_CASE(12, 3)
  EF_CASE(16, 1) EF_CASE(16, 2) EF_CASE(16, 3)
  EF_CASE(20, 1) EF_CASE(20, 2) EF_CASE(20, 3)
  EF_CASE(24, 2) EF_CASE(24, 3)
  EF_CASE(32, 2) EF_CASE(32, 3)
#endif
#undef EF_CASE
  return false;
}

}  // namespace
