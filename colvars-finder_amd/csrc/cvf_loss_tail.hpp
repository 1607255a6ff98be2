// The scalar tail of EigenFunctionTask.loss_func (core.py:426-457) and its partial derivatives with respect to
// the batch sums, fp64, one thread.  Its small arrays are indexed through the sorted order cvec, i.e. dynamically:
// as private arrays they live in scratch memory (a global-memory round trip per access, ~6 us for the whole tail);
// declared __shared__ they cost an LDS access each.  One thread of the block runs it, so there is nothing to race.  Shared by stats.hip (stand-alone launches, data-parallel path) and the
// derivative kernel's fused epilogue (k1_align.hip, single process).
#pragma once
#include "cvf_common.hpp"

constexpr int kMaxStats = 1 + CVF_MAX_NETS + CVF_NPAIR(CVF_MAX_NETS) + 1 + 3 * CVF_MAX_NETS;

// the scalar tail of loss_func, one thread, fp64
template <int KT>
__device__ void ef_loss_tail(const cvf_ef_cfg& cfg, const double* __restrict__ stats, double* __restrict__ loss_vec,
                             double* __restrict__ coef) {
  constexpr int k = KT;
  const int npair = CVF_NPAIR(k);
  const double W = stats[0];
  const double* S1 = stats + 1;
  const double* S2 = stats + 1 + k;
  __shared__ double m[KT], v[KT], s2[KT][KT];
  {
    int p = 0;
    for (int i = 0; i < k; ++i)
      for (int j = i; j < k; ++j) {
        s2[i][j] = s2[j][i] = S2[p++];
      }
  }
  // (one reciprocal of W, one of each denominator: the dozen fp64 divisions this replaces were a dependent chain of
  //  ~200 cycles each in a single thread that the whole step waits for)
  const double iW = 1.0 / W;
  for (int i = 0; i < k; ++i) {
    m[i] = S1[i] * iW;                      // core.py:409
    v[i] = s2[i][i] * iW - m[i] * m[i];     // core.py:410
  }
  __shared__ double eig[KT], num[KT], den[KT], iden[KT];
  double pref;
  __shared__ double vl[KT], ml[KT];
  double Wl = 1.0;
  const int o = 1 + k + npair;
  if (cfg.lag_idx == 0) {
    pref = iW / cfg.beta;                   // core.py:426,438
    for (int i = 0; i < k; ++i) {
      num[i] = stats[o + i];
      den[i] = v[i];
      iden[i] = 1.0 / den[i];
      eig[i] = pref * num[i] * iden[i];
    }
  } else {
    Wl = stats[o];
    const double iWl0 = 1.0 / Wl;
    for (int i = 0; i < k; ++i) {
      ml[i] = stats[o + 1 + i] * iWl0;                          // core.py:415
      vl[i] = stats[o + 1 + k + i] * iWl0 - ml[i] * ml[i];      // core.py:416
      num[i] = stats[o + 1 + 2 * k + i];
      den[i] = v[i] + vl[i];
      iden[i] = 1.0 / den[i];
    }
    pref = iW / (cfg.dt * cfg.lag_idx);                       // core.py:428,440
    for (int i = 0; i < k; ++i) eig[i] = pref * num[i] * iden[i];
  }
  // cvec = argsort(eig) (core.py:432), stable insertion sort
  __shared__ int cvec[KT];
#pragma unroll
  for (int i = 0; i < k; ++i) cvec[i] = i;
  if (cfg.sort_eigvals) {
    for (int i = 1; i < k; ++i) {
      const int c = cvec[i];
      int j = i - 1;
      while (j >= 0 && eig[cvec[j]] > eig[c]) {
        cvec[j + 1] = cvec[j];
        --j;
      }
      cvec[j + 1] = c;
    }
  }
  // variational objective; generator: numerator AND denominator at cvec[idx] (core.py:438);
  // transfer: numerator at idx, denominator at cvec[idx] (core.py:440, reproduced as is)
  double npl = 0.0;
  __shared__ double gnum[KT], gden[KT];
  for (int i = 0; i < k; ++i) gnum[i] = gden[i] = 0.0;
  for (int idx = 0; idx < k; ++idx) {
    const int c = cvec[idx];
    const int nsrc = cfg.lag_idx == 0 ? c : idx;
    npl += cfg.eig_w[idx] * num[nsrc] * iden[c];
    gnum[nsrc] += pref * cfg.eig_w[idx] * iden[c];
    gden[c] += -pref * cfg.eig_w[idx] * num[nsrc] * (iden[c] * iden[c]);
  }
  npl *= pref;
  double pen = 0.0;
  __shared__ double cov[KT][KT];
  for (int i = 0; i < k; ++i) pen += (v[i] - 1.0) * (v[i] - 1.0);           // core.py:446
  for (int i = 0; i < k; ++i)
    for (int j = i + 1; j < k; ++j) {
      cov[i][j] = cov[j][i] = s2[i][j] * iW - m[i] * m[j];                  // core.py:452
      pen += cov[i][j] * cov[i][j];
    }
  const double loss = npl + cfg.alpha * pen;                                // core.py:455
  loss_vec[0] = loss;
  loss_vec[1] = npl;
  loss_vec[2] = pen;
  for (int idx = 0; idx < k; ++idx) {
    loss_vec[3 + idx] = eig[cvec[idx]];                                     // core.py:434
    loss_vec[3 + k + idx] = (double)cvec[idx];
  }
  // ---- partial derivatives (eigenvalues and cvec are constants: core.py:426,428 detach them)
  double* gS1 = coef;
  double* gS2 = coef + k;
  double* gEt = coef + k + k * k;
  double* gS1l = coef + 2 * k + k * k;
  double* gS2l = coef + 3 * k + k * k;
  for (int i = 0; i < k; ++i) {
    const double Lv = gden[i] + 2.0 * cfg.alpha * (v[i] - 1.0);   // d loss / d var_i
    double g1 = Lv * (-2.0 * m[i] * iW);
    for (int j = 0; j < k; ++j)
      if (j != i) g1 += 2.0 * cfg.alpha * cov[i][j] * (-m[j] * iW);
    gS1[i] = g1;
    for (int j = 0; j < k; ++j) gS2[i * k + j] = (i == j) ? Lv * iW : 2.0 * cfg.alpha * cov[i][j] * iW;
    gEt[i] = gnum[i];
    if (cfg.lag_idx > 0) {
      const double Lvl = gden[i];                                 // d loss / d var'_i
      const double iWl = 1.0 / Wl;
      gS1l[i] = Lvl * (-2.0 * ml[i] * iWl);
      gS2l[i] = Lvl * iWl;
    } else {
      gS1l[i] = 0.0;
      gS2l[i] = 0.0;
    }
  }
}


