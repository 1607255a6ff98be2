// The scalar tail of EigenFunctionTask.loss_func (core.py:426-457) and its partial derivatives with respect to
// the batch sums, fp64, one thread.  Everything lives in REGISTERS: the loops run to the template parameter KT (fully
// unrolled, static indices) and the accesses through the sorted order cvec - dynamic indices - are written as selects
// over the KT candidates.  (Round 1 kept the small arrays in LDS because dynamically indexed private arrays go to scratch
// memory: ~150 dependent LDS round trips of ~100 cycles each were most of the finishing launch the whole step waits for.)
// Shared by stats.hip (stand-alone launches, data-parallel path) and the derivative kernel's fused epilogue (k1_align.hip).
#pragma once
#include "cvf_common.hpp"

constexpr int kMaxStats = 1 + CVF_MAX_NETS + CVF_NPAIR(CVF_MAX_NETS) + 1 + 3 * CVF_MAX_NETS;

template <int KT>
__device__ __forceinline__ double tail_pick(const double (&a)[KT], int c) {
  double r = a[0];
#pragma unroll
  for (int j = 1; j < KT; ++j) r = (c == j) ? a[j] : r;
  return r;
}

// the scalar tail of loss_func, one thread, fp64
template <int KT>
__device__ void ef_loss_tail(const cvf_ef_cfg& cfg, const double* __restrict__ stats, double* __restrict__ loss_vec,
                             double* __restrict__ coef) {
  constexpr int k = KT;
  constexpr int npair = CVF_NPAIR(k);
  const double W = stats[0];
  const double* S1 = stats + 1;
  const double* S2 = stats + 1 + k;
  double m[KT], v[KT], s2[KT][KT];
  {
    int p = 0;
#pragma unroll
    for (int i = 0; i < k; ++i)
#pragma unroll
      for (int j = i; j < k; ++j) {
        s2[i][j] = s2[j][i] = S2[p];
        ++p;
      }
  }
  // (one reciprocal of W, one of each denominator: a dozen dependent fp64 divisions otherwise)
  const double iW = 1.0 / W;
#pragma unroll
  for (int i = 0; i < k; ++i) {
    m[i] = S1[i] * iW;                      // core.py:409
    v[i] = s2[i][i] * iW - m[i] * m[i];     // core.py:410
  }
  double eig[KT], num[KT], iden[KT];
  double pref;
  double vl[KT], ml[KT];
  double Wl = 1.0;
  const int o = 1 + k + npair;
  const bool gen = cfg.lag_idx == 0;
  if (gen) {
    pref = iW / cfg.beta;                   // core.py:426,438
#pragma unroll
    for (int i = 0; i < k; ++i) {
      num[i] = stats[o + i];
      iden[i] = 1.0 / v[i];
      eig[i] = pref * num[i] * iden[i];
      vl[i] = ml[i] = 0.0;
    }
  } else {
    Wl = stats[o];
    const double iWl0 = 1.0 / Wl;
#pragma unroll
    for (int i = 0; i < k; ++i) {
      ml[i] = stats[o + 1 + i] * iWl0;                          // core.py:415
      vl[i] = stats[o + 1 + k + i] * iWl0 - ml[i] * ml[i];      // core.py:416
      num[i] = stats[o + 1 + 2 * k + i];
      iden[i] = 1.0 / (v[i] + vl[i]);
    }
    pref = iW / (cfg.dt * cfg.lag_idx);                       // core.py:428,440
#pragma unroll
    for (int i = 0; i < k; ++i) eig[i] = pref * num[i] * iden[i];
  }
  // cvec = argsort(eig) (core.py:432), stable: position of i = number of entries that sort before it
  int cvec[KT];
#pragma unroll
  for (int i = 0; i < k; ++i) cvec[i] = i;
  if (cfg.sort_eigvals) {
    int rank[KT];
#pragma unroll
    for (int i = 0; i < k; ++i) {
      int r = 0;
#pragma unroll
      for (int j = 0; j < k; ++j) r += (eig[j] < eig[i] || (eig[j] == eig[i] && j < i)) ? 1 : 0;
      rank[i] = r;
    }
#pragma unroll
    for (int pos = 0; pos < k; ++pos) {
      int c = 0;
#pragma unroll
      for (int i = 0; i < k; ++i) c = (rank[i] == pos) ? i : c;
      cvec[pos] = c;
    }
  }
  // variational objective; generator: numerator AND denominator at cvec[idx] (core.py:438);
  // transfer: numerator at idx, denominator at cvec[idx] (core.py:440, reproduced as is)
  double npl = 0.0;
  double gnum[KT], gden[KT];
#pragma unroll
  for (int i = 0; i < k; ++i) gnum[i] = gden[i] = 0.0;
#pragma unroll
  for (int idx = 0; idx < k; ++idx) {
    const int c = cvec[idx];
    const int nsrc = gen ? c : idx;
    const double ew = cfg.eig_w[idx];
    const double num_s = tail_pick<KT>(num, nsrc), iden_c = tail_pick<KT>(iden, c);
    npl += ew * num_s * iden_c;
    const double dn = pref * ew * iden_c, dd = -pref * ew * num_s * (iden_c * iden_c);
#pragma unroll
    for (int j = 0; j < k; ++j) {
      gnum[j] += (nsrc == j) ? dn : 0.0;
      gden[j] += (c == j) ? dd : 0.0;
    }
  }
  npl *= pref;
  double pen = 0.0;
  double cov[KT][KT];
#pragma unroll
  for (int i = 0; i < k; ++i) pen += (v[i] - 1.0) * (v[i] - 1.0);           // core.py:446
#pragma unroll
  for (int i = 0; i < k; ++i) {
    cov[i][i] = 0.0;
#pragma unroll
    for (int j = i + 1; j < k; ++j) {
      cov[i][j] = cov[j][i] = s2[i][j] * iW - m[i] * m[j];                  // core.py:452
      pen += cov[i][j] * cov[i][j];
    }
  }
  const double loss = npl + cfg.alpha * pen;                                // core.py:455
  loss_vec[0] = loss;
  loss_vec[1] = npl;
  loss_vec[2] = pen;
#pragma unroll
  for (int idx = 0; idx < k; ++idx) {
    loss_vec[3 + idx] = tail_pick<KT>(eig, cvec[idx]);                      // core.py:434
    loss_vec[3 + k + idx] = (double)cvec[idx];
  }
  // ---- partial derivatives (eigenvalues and cvec are constants: core.py:426,428 detach them)
  double* gS1 = coef;
  double* gS2 = coef + k;
  double* gEt = coef + k + k * k;
  double* gS1l = coef + 2 * k + k * k;
  double* gS2l = coef + 3 * k + k * k;
  const double iWl = 1.0 / Wl;
#pragma unroll
  for (int i = 0; i < k; ++i) {
    const double Lv = gden[i] + 2.0 * cfg.alpha * (v[i] - 1.0);   // d loss / d var_i
    double g1 = Lv * (-2.0 * m[i] * iW);
#pragma unroll
    for (int j = 0; j < k; ++j)
      if (j != i) g1 += 2.0 * cfg.alpha * cov[i][j] * (-m[j] * iW);
    gS1[i] = g1;
#pragma unroll
    for (int j = 0; j < k; ++j) gS2[i * k + j] = (i == j) ? Lv * iW : 2.0 * cfg.alpha * cov[i][j] * iW;
    gEt[i] = gnum[i];
    if (!gen) {
      const double Lvl = gden[i];                                 // d loss / d var'_i
      gS1l[i] = Lvl * (-2.0 * ml[i] * iWl);
      gS2l[i] = Lvl * iWl;
    } else {
      gS1l[i] = 0.0;
      gS2l[i] = 0.0;
    }
  }
}
