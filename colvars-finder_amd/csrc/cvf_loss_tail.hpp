// The scalar tail of EigenFunctionTask.loss_func (core.py:426-457) and its partial derivatives with respect to the batch
// sums, fp64.  Used by stats.hip (the finishing launch of the batch sums; cvf_ef_loss of the data-parallel path).
#pragma once
#include "cvf_common.hpp"

constexpr int kMaxStats = 1 + CVF_MAX_NETS + CVF_NPAIR(CVF_MAX_NETS) + 1 + 3 * CVF_MAX_NETS;

// ------------------------------------------------------------------------------------------------------------------
// Executed by ONE WAVE (all 64 lanes active), for any k <= 8 at run time: lane l = i + 8 j works on net i (and on the pair
// (i, j)); values of other nets travel by v_readlane (uniform source) or ds_bpermute (per-lane source), sums over nets run
// in a fixed order (idx = 0..k-1; pairs row-major).  Rounds 1-2 ran this on ONE thread with everything in registers (k a
// template parameter): ~400 dependent fp64 instructions incl. five IEEE divisions, ~2 us at the end of the launch the whole
// step waits for.  This form is three divisions deep, needs no template parameter - so it can sit at the end of a kernel
// that is not instantiated per k - and a handful of registers.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double tail_readlane(double v, int src_lane) {   // src_lane wave-uniform
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, src_lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src_lane);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double tail_shfl(double v, int src_lane) {       // src_lane per lane
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(unsigned)u);
  const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(unsigned)(u >> 32));
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ void ef_loss_tail_wave(const cvf_ef_cfg& cfg, const double* stats, double* loss_vec, double* coef) {
  const int lane = threadIdx.x & 63;
  const int k = cfg.k, npair = CVF_NPAIR(k);
  const int i = lane & 7, j = lane >> 3;
  const int ic = i < k ? i : k - 1, jc = j < k ? j : k - 1;   // (clamped: every lane loads and computes, stores are predicated)
  const bool gen = cfg.lag_idx == 0;
  const int o = 1 + k + npair;
  auto pidx = [&](int a, int b) {   // S2 is stored for a <= b, row-major
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    return lo * k - (lo * (lo - 1)) / 2 + (hi - lo);
  };
  const double W = stats[0];
  const double S1i = stats[1 + ic], S1j = stats[1 + jc];
  const double s2ii = stats[1 + k + pidx(ic, ic)], s2ij = stats[1 + k + pidx(ic, jc)];
  const double num = stats[gen ? o + ic : o + 1 + 2 * k + ic];
  const double Wl = gen ? 1.0 : stats[o];
  const double S1l = gen ? 0.0 : stats[o + 1 + ic], S2l = gen ? 0.0 : stats[o + 1 + k + ic];
  double ew = cfg.eig_w[0];
#pragma unroll
  for (int t = 1; t < CVF_MAX_NETS; ++t) ew = (ic == t) ? cfg.eig_w[t] : ew;
  const double iW = 1.0 / W;
  const double mi = S1i * iW, mj = S1j * iW;                   // core.py:409
  const double v = s2ii * iW - mi * mi;                        // core.py:410
  const double iWl = 1.0 / Wl;
  const double ml = S1l * iWl;                                 // core.py:415
  const double vl = gen ? 0.0 : S2l * iWl - ml * ml;           // core.py:416
  const double pref = gen ? iW / cfg.beta : iW / (cfg.dt * cfg.lag_idx);   // core.py:426,438 / 428,440
  const double iden = 1.0 / (gen ? v : v + vl);
  const double eig = pref * num * iden;
  // cvec = argsort(eig) (core.py:432), stable: position of i = number of entries that sort before it
  int rank = ic;
  if (cfg.sort_eigvals) {
    rank = 0;
    for (int t = 0; t < k; ++t) {
      const double e = tail_readlane(eig, t);   // lane t = (net t, j = 0)
      rank += (e < eig || (e == eig && t < ic)) ? 1 : 0;
    }
  }
  int c = 0;   // cvec[i]: the net whose rank is this lane's position i
  for (int t = 0; t < k; ++t) c = (__builtin_amdgcn_readlane(rank, t) == ic) ? t : c;
  // variational objective; generator: numerator AND denominator at cvec[idx] (core.py:438);
  // transfer: numerator at idx, denominator at cvec[idx] (core.py:440, reproduced as is)
  const int nsrc = gen ? c : ic;
  const double num_s = tail_shfl(num, nsrc), iden_c = tail_shfl(iden, c);
  const double term = ew * num_s * iden_c;
  const double dn = pref * ew * iden_c, dd = -pref * ew * num_s * (iden_c * iden_c);
  double npl = 0.0, gnum = 0.0, gden = 0.0;
  for (int t = 0; t < k; ++t) {
    npl += tail_readlane(term, t);
    gnum += (__builtin_amdgcn_readlane(nsrc, t) == ic) ? tail_readlane(dn, t) : 0.0;
    gden += (__builtin_amdgcn_readlane(c, t) == ic) ? tail_readlane(dd, t) : 0.0;
  }
  npl *= pref;
  const double dv = (v - 1.0) * (v - 1.0);                     // core.py:446
  const double cov = s2ij * iW - mi * mj;                      // core.py:452 (lane (i, j); symmetric)
  const double cov2 = cov * cov;
  double pen = 0.0;
  for (int a = 0; a < k; ++a) pen += tail_readlane(dv, a);
  for (int a = 0; a < k; ++a)
    for (int b = a + 1; b < k; ++b) pen += tail_readlane(cov2, a + 8 * b);
  const double loss = npl + cfg.alpha * pen;                   // core.py:455
  // ---- partial derivatives (eigenvalues and cvec are constants: core.py:426,428 detach them)
  const double Lv = gden + 2.0 * cfg.alpha * (v - 1.0);        // d loss / d var_i
  const double off = (i == j) ? 0.0 : 2.0 * cfg.alpha * cov * (-mj * iW);   // term of gS1[i] held by lane (i, j)
  double g1 = Lv * (-2.0 * mi * iW);
  for (int t = 0; t < k; ++t) {
    const double o_t = tail_shfl(off, ic + 8 * t);
    g1 += (t != ic) ? o_t : 0.0;
  }
  const double eig_sorted = tail_shfl(eig, c);                 // core.py:434
  if (lane == 0) {
    loss_vec[0] = loss;
    loss_vec[1] = npl;
    loss_vec[2] = pen;
  }
  if (i < k && j == 0) {
    loss_vec[3 + i] = eig_sorted;
    loss_vec[3 + k + i] = (double)c;
    coef[i] = g1;                                              // gS1
    coef[k + k * k + i] = gnum;                                // gEt
    coef[2 * k + k * k + i] = gen ? 0.0 : gden * (-2.0 * ml * iWl);   // gS1'
    coef[3 * k + k * k + i] = gen ? 0.0 : gden * iWl;                  // gS2'_ii
  }
  if (i < k && j < k) coef[k + i * k + j] = (i == j) ? Lv * iW : 2.0 * cfg.alpha * cov * iW;   // gS2
}
