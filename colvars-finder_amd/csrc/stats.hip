// K5: batch statistics of EigenFunctionTask.loss_func (core.py:406-416,426,428,446-452) in
// fp64 with a fixed-order two-stage reduction (bitwise reproducible: every rank of a
// data-parallel job must derive the same ordering cvec from the reduced vector), the scalar
// tail of the loss (core.py:426-457) with its partial derivatives, and K6: Adam.
#include "cvf_common.hpp"
#include "cvf_adam.hpp"
#include "cvf_loss_tail.hpp"
#include "cvf_p2p.hpp"
#include <stdarg.h>
#include <stdio.h>
#include <type_traits>

// ---------------------------------------------------------------------------------------
// error plumbing shared by all translation units
// ---------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void cvf_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int cvf_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    cvf_set_error("%s: %s", what, hipGetErrorString(e));
    return -2;
  }
  return 0;
}
extern "C" const char* cvf_last_error(void) { return g_err; }
extern "C" int cvf_version(void) { return 100; }

extern "C" int cvf_ef_nstats(int k, int lag_idx) {
  const int base = 1 + k + CVF_NPAIR(k);
  return lag_idx == 0 ? base + k : base + 1 + 2 * k + k;
}

namespace {

constexpr int kStatBlocks = 128;

// one wave per block; block g handles tiles g, g+G, ...; lane = frame.  K is a template
// parameter so that every accumulator lives in a register (no runtime-indexed arrays).
template <int K, bool LAG>
__global__ __launch_bounds__(64) void ef_stats_partial_kernel(int64_t B, const float* __restrict__ w,
                                                               const float* __restrict__ y, const float* __restrict__ e,
                                                               const float* __restrict__ w_lag,
                                                               const float* __restrict__ y_lag,
                                                               double* __restrict__ partial) {
  constexpr int NP = K * (K + 1) / 2;
  constexpr int NS = LAG ? 1 + K + NP + 1 + 3 * K : 1 + K + NP + K;
  constexpr int O = 1 + K + NP;
  const int lane = threadIdx.x;
  const int64_t T = (B + CVF_TILE - 1) / CVF_TILE;
  double acc[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) acc[i] = 0.0;
  for (int64_t tile = blockIdx.x; tile < T; tile += gridDim.x) {
    const int64_t frame = tile * CVF_TILE + lane;
    const bool valid = frame < B;
    const double wb = valid ? (double)w[frame] : 0.0;
    double yv[K];
#pragma unroll
    for (int i = 0; i < K; ++i) yv[i] = (double)y[(tile * K + i) * CVF_TILE + lane];
    acc[0] += wb;
    int p = 0;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      acc[1 + i] += wb * yv[i];
#pragma unroll
      for (int j = i; j < K; ++j) acc[1 + K + p++] += wb * yv[i] * yv[j];
    }
    if (!LAG) {
#pragma unroll
      for (int i = 0; i < K; ++i) acc[O + i] += wb * (double)e[(tile * K + i) * CVF_TILE + lane];
    } else {
      const double wl = valid ? (double)w_lag[frame] : 0.0;
      acc[O] += wl;
#pragma unroll
      for (int i = 0; i < K; ++i) {
        const double yl = (double)y_lag[(tile * K + i) * CVF_TILE + lane];
        acc[O + 1 + i] += wl * yl;
        acc[O + 1 + K + i] += wl * yl * yl;
        const double df = yl - yv[i];
        acc[O + 1 + 2 * K + i] += wb * df * df;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const double s = wave_sum(acc[i]);
    if (lane == 0) partial[(int64_t)blockIdx.x * NS + i] = s;
  }
}

__global__ __launch_bounds__(64) void ef_loss_kernel(cvf_ef_cfg cfg, const double* __restrict__ stats, double* __restrict__ loss_vec,
                                                     double* __restrict__ coef) {
  if (blockIdx.x == 0) ef_loss_tail_wave(cfg, stats, loss_vec, coef);   // one wave, every lane active
}

// data-parallel step: this rank's batch sums -> the sum over ranks (low-latency peer-to-peer exchange, cvf_p2p.hpp; written
// back to `stats`) -> the loss tail, in ONE launch (was: all-reduce launch + ef_loss_kernel)
__global__ __launch_bounds__(256) void ef_loss_dp_kernel(cvf_ef_cfg cfg, int ns, double* __restrict__ stats, double* __restrict__ loss_vec,
                                                         double* __restrict__ coef, P2PLL ll) {
  __shared__ double fin[kMaxStats];
  __shared__ unsigned parts[kP2PMaxWorld * 2 * kMaxStats];
  const unsigned ex = p2p_ll_next(ll);
  for (int i = threadIdx.x; i < ns; i += blockDim.x) fin[i] = stats[i];
  p2p_ll_allreduce_stats(ll, fin, ns, parts, ex);
  for (int i = threadIdx.x; i < ns; i += blockDim.x) stats[i] = fin[i];
  if (threadIdx.x < CVF_WAVE) ef_loss_tail_wave(cfg, fin, loss_vec, coef);
}

// Second reduction stage (+ the scalar tail when loss_vec != NULL, i.e. no cross-rank reduction in between) in ONE
// launch.  Stat i is summed by one wave: lane l adds rows l, l+64, ... (loads issued eight at a time), the 64 lane
// sums are combined by the fixed-order DPP reduction -> bitwise reproducible for a given number of rows.
__global__ __launch_bounds__(1024) void ef_stats_finish_kernel(cvf_ef_cfg cfg, int ns, int n_rows, int stat_major,
                                                               const double* __restrict__ partial,
                                                               double* __restrict__ stats, double* __restrict__ loss_vec,
                                                               double* __restrict__ coef, P2PLL ll) {
  __shared__ double fin[kMaxStats];
  __shared__ unsigned parts[kP2PMaxWorld * 2 * kMaxStats];   // (data-parallel exchange only)
  const unsigned ex = p2p_ll_next(ll);                       // (requested now, used behind the row sums)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  // (24 row loads in flight per lane: the 1250 unit rows of a 20 000-frame batch (csrc/ef16_front.hip, ef16_back.hip) are ONE round trip per
  //  statistic; with eight it was three dependent ones, 9 us for this launch)
  // stat_major: partial is [statistic][row] - a wave's 64 lanes read 512 consecutive bytes per load instead of one 8-byte
  // word in each of 64 rows (64 cache lines per instruction: with 1250 rows that access pattern alone was 8 us)
  const int64_t rs = stat_major ? 1 : ns, is = stat_major ? n_rows : 1;
  // kSt statistics per wave at a time, kR row loads in flight for each.  More statistics than waves (k >= 4: 19..53) used to be
  // ceil(ns / 16) passes of one statistic per wave - as many DEPENDENT memory round trips (11 us at k = 6, the rows just written
  // by other CUs); four statistics share a pass now.
  auto pass = [&](auto kst_, auto kr_) {
    constexpr int kSt = decltype(kst_)::value, kR = decltype(kr_)::value;
    for (int i0 = wave; i0 < ns; i0 += nw * kSt) {
      double acc[kSt];
#pragma unroll
      for (int j = 0; j < kSt; ++j) acc[j] = 0.0;
      for (int g0 = lane; g0 < n_rows; g0 += CVF_WAVE * kR) {
        double v[kSt][kR];
#pragma unroll
        for (int j = 0; j < kSt; ++j) {
          const int i = i0 + nw * j < ns ? i0 + nw * j : ns - 1;
#pragma unroll
          for (int b = 0; b < kR; ++b) {
            const int g = g0 + CVF_WAVE * b;
            v[j][b] = partial[(int64_t)(g < n_rows ? g : n_rows - 1) * rs + i * is];
          }
        }
#pragma unroll
        for (int j = 0; j < kSt; ++j)
#pragma unroll
          for (int b = 0; b < kR; ++b) acc[j] += (g0 + CVF_WAVE * b < n_rows) ? v[j][b] : 0.0;
      }
#pragma unroll
      for (int j = 0; j < kSt; ++j) {
        const double sum = wave_sum(acc[j]);
        const int i = i0 + nw * j;
        if (lane == 0 && i < ns) {
          fin[i] = sum;
          if (ll.world == 0) stats[i] = sum;
        }
      }
    }
  };
  if (ns <= nw) pass(std::integral_constant<int, 1>{}, std::integral_constant<int, 24>{});
  else if (ns <= 2 * nw) pass(std::integral_constant<int, 2>{}, std::integral_constant<int, 20>{});   // (the 20 time-lagged sums at k = 3: 40 loads in flight per lane - the 1250 unit rows of a 20 000-frame batch in ONE round trip, not four)
  else pass(std::integral_constant<int, 4>{}, std::integral_constant<int, 6>{});
  if (ll.world > 0) {   // data-parallel step: the sum over ranks of the local sums, inside this launch (cvf_p2p.hpp)
    p2p_ll_allreduce_stats(ll, fin, ns, parts, ex);
    for (int i = threadIdx.x; i < ns; i += blockDim.x) stats[i] = fin[i];
  }
  if (loss_vec == nullptr) return;
  __syncthreads();
  if (wave == 0) ef_loss_tail_wave(cfg, fin, loss_vec, coef);
}

__global__ void adam_kernel(AdamDev a, const float* __restrict__ grad, int64_t n, cvf_mlp_desc mlp) {
  __shared__ PackTab tab;
  __shared__ AdamScalars sc;
  if (threadIdx.x == 0 && a.packed != nullptr) pack_tab_fill(tab, mlp);
  if (threadIdx.x == 64 % blockDim.x) sc = adam_scalars(a);
  __syncthreads();
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    adam_apply(a, sc, tab, i, grad[i]);
}

__global__ void sgd_kernel(float* __restrict__ theta, const float* __restrict__ grad, int64_t n, float lr,
                           const float* __restrict__ lr_dev, cvf_mlp_desc mlp, float* __restrict__ packed) {
  __shared__ PackTab tab;
  if (lr_dev != nullptr) lr = *lr_dev;
  if (threadIdx.x == 0 && packed != nullptr) pack_tab_fill(tab, mlp);
  __syncthreads();
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float th = theta[i] - lr * grad[i];
    theta[i] = th;
    if (packed != nullptr) pack_scatter(tab, (int)i, th, packed);
  }
}

}  // namespace

extern "C" int64_t cvf_ef_stats_scratch_doubles(int k, int lag_idx) {
  return (int64_t)kStatBlocks * cvf_ef_nstats(k, lag_idx);
}

template <class F>
static bool k_dispatch(int k, F&& f) {
#define K_CASE(K_) if (k == K_) { f(std::integral_constant<int, K_>{}); return true; }
  K_CASE(1) K_CASE(2) K_CASE(3) K_CASE(4) K_CASE(5) K_CASE(6) K_CASE(7) K_CASE(8)
#undef K_CASE
  return false;
}

// rows of per-block (or per-tile, see k1_align.hip) partial sums -> stats [+ loss_vec, coef]
// (ll: NULL, or the peer-to-peer communicator's device view - the sums are then exchanged inside the finishing launch)
int cvf_ef_stats_finish_ll(const cvf_ef_cfg* cfg, int n_rows, int stat_major, const double* partial, double* stats,
                           double* loss_vec, double* coef, hipStream_t s, const P2PLL* ll);
int cvf_ef_stats_finish_impl(const cvf_ef_cfg* cfg, int n_rows, int stat_major, const double* partial, double* stats,
                             double* loss_vec, double* coef, hipStream_t s) {
  return cvf_ef_stats_finish_ll(cfg, n_rows, stat_major, partial, stats, loss_vec, coef, s, nullptr);
}
int cvf_ef_stats_finish(const cvf_ef_cfg* cfg, int n_rows, const double* partial, double* stats, double* loss_vec, double* coef,
                        hipStream_t s) {
  return cvf_ef_stats_finish_impl(cfg, n_rows, 0, partial, stats, loss_vec, coef, s);
}
int cvf_ef_stats_finish_ll(const cvf_ef_cfg* cfg, int n_rows, int stat_major, const double* partial, double* stats,
                           double* loss_vec, double* coef, hipStream_t s, const P2PLL* ll) {
  const int ns = cvf_ef_nstats(cfg->k, cfg->lag_idx);
  const int waves = ns < 16 ? ns : 16;
  P2PLL none = {};
  hipLaunchKernelGGL(ef_stats_finish_kernel, dim3(1), dim3(64 * waves), 0, s, *cfg, ns, n_rows, stat_major, partial, stats, loss_vec,
                     coef, ll != nullptr ? *ll : none);
  return cvf_check_launch("ef_stats_finish_kernel");
}

static int ef_stats_impl(const cvf_ef_cfg* cfg, int64_t B, const float* w, const float* y_tiled, const float* e_tiled,
                         const float* w_lag, const float* y_lag_tiled, double* scratch, double* stats, double* loss_vec,
                         double* coef, void* stream, const P2PLL* ll) {
  CVF_REQUIRE(cfg && w && y_tiled && scratch && stats && B > 0, "cvf_ef_stats: bad argument");
  CVF_REQUIRE(cfg->k >= 1 && cfg->k <= CVF_MAX_NETS, "cvf_ef_stats: k=%d out of range", cfg->k);
  if (cfg->lag_idx == 0) CVF_REQUIRE(e_tiled, "cvf_ef_stats: generator mode needs e_tiled");
  else CVF_REQUIRE(w_lag && y_lag_tiled, "cvf_ef_stats: transfer mode needs lagged inputs");
  const int ns = cvf_ef_nstats(cfg->k, cfg->lag_idx);
  const int64_t T = cvf_ntiles(B);
  const int G = (int)(T < kStatBlocks ? T : kStatBlocks);
  hipStream_t s = (hipStream_t)stream;
  const bool lag = cfg->lag_idx > 0;
  k_dispatch(cfg->k, [&](auto kc) {
    constexpr int K = decltype(kc)::value;
    if (lag)
      hipLaunchKernelGGL((ef_stats_partial_kernel<K, true>), dim3(G), dim3(64), 0, s, B, w, y_tiled, e_tiled, w_lag,
                         y_lag_tiled, scratch);
    else
      hipLaunchKernelGGL((ef_stats_partial_kernel<K, false>), dim3(G), dim3(64), 0, s, B, w, y_tiled, e_tiled, w_lag,
                         y_lag_tiled, scratch);
  });
  int rc = cvf_check_launch("ef_stats_partial_kernel");
  if (rc) return rc;
  CVF_REQUIRE(loss_vec == nullptr || coef != nullptr, "cvf_ef_stats: loss_vec without coef");
  return cvf_ef_stats_finish_ll(cfg, G, 0, scratch, stats, loss_vec, coef, s, ll);
}
extern "C" int cvf_ef_stats(const cvf_ef_cfg* cfg, int64_t B, const float* w, const float* y_tiled, const float* e_tiled,
                            const float* w_lag, const float* y_lag_tiled, double* scratch, double* stats, double* loss_vec,
                            double* coef, void* stream) {
  return ef_stats_impl(cfg, B, w, y_tiled, e_tiled, w_lag, y_lag_tiled, scratch, stats, loss_vec, coef, stream, nullptr);
}
extern "C" int cvf_ef_stats_dp(const cvf_ef_cfg* cfg, int64_t B, const float* w, const float* y_tiled, const float* e_tiled,
                               const float* w_lag, const float* y_lag_tiled, double* scratch, double* stats, double* loss_vec,
                               double* coef, void* p2p_comm, void* stream) {
  const P2PLL* ll = cvf_p2p_ll(p2p_comm, 0);
  if (ll == nullptr) return -1;
  CVF_REQUIRE(loss_vec != nullptr && coef != nullptr, "cvf_ef_stats_dp: needs loss_vec and coef (the loss tail runs behind the exchange)");
  return ef_stats_impl(cfg, B, w, y_tiled, e_tiled, w_lag, y_lag_tiled, scratch, stats, loss_vec, coef, stream, ll);
}

extern "C" int cvf_ef_loss(const cvf_ef_cfg* cfg, const double* stats, double* loss_vec, double* coef, void* stream) {
  CVF_REQUIRE(cfg && stats && loss_vec && coef, "cvf_ef_loss: bad argument");
  CVF_REQUIRE(cfg->k >= 1 && cfg->k <= CVF_MAX_NETS, "cvf_ef_loss: k=%d out of range", cfg->k);
  hipLaunchKernelGGL(ef_loss_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, *cfg, stats, loss_vec, coef);
  return cvf_check_launch("ef_loss_kernel");
}

extern "C" int cvf_ef_loss_dp(const cvf_ef_cfg* cfg, double* stats, double* loss_vec, double* coef, void* p2p_comm, void* stream) {
  CVF_REQUIRE(cfg && stats && loss_vec && coef, "cvf_ef_loss_dp: bad argument");
  CVF_REQUIRE(cfg->k >= 1 && cfg->k <= CVF_MAX_NETS, "cvf_ef_loss_dp: k=%d out of range", cfg->k);
  const P2PLL* ll = cvf_p2p_ll(p2p_comm, 0);
  if (ll == nullptr) return -1;
  hipLaunchKernelGGL(ef_loss_dp_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, *cfg, cvf_ef_nstats(cfg->k, cfg->lag_idx), stats,
                     loss_vec, coef, *ll);
  return cvf_check_launch("ef_loss_dp_kernel");
}

extern "C" int cvf_adam_step(float* theta, const float* grad, float* m, float* v, int64_t n, double lr, const float* lr_dev,
                             double beta1, double beta2, double eps, int32_t* step_count, const cvf_mlp_desc* mlp,
                             float* packed, void* stream) {
  CVF_REQUIRE(theta && grad && m && v && step_count && n > 0, "cvf_adam_step: bad argument");
  CVF_REQUIRE(packed == nullptr || (mlp != nullptr && mlp->n_params == n), "cvf_adam_step: packed buffer needs its mlp desc");
  cvf_mlp_desc none = {};
  const cvf_mlp_desc& md = packed ? *mlp : none;
  const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  AdamDev ad{theta, m, v, (float)lr, (float)beta1, (float)beta2, (float)eps, step_count, packed, lr_dev};
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ad, grad, n, md);
  return cvf_check_launch("adam_kernel");
}

extern "C" int cvf_sgd_step(float* theta, const float* grad, int64_t n, double lr, const float* lr_dev,
                            const cvf_mlp_desc* mlp, float* packed, void* stream) {
  CVF_REQUIRE(theta && grad && n > 0, "cvf_sgd_step: bad argument");
  CVF_REQUIRE(packed == nullptr || (mlp != nullptr && mlp->n_params == n), "cvf_sgd_step: packed buffer needs its mlp desc");
  cvf_mlp_desc none = {};
  const cvf_mlp_desc& md = packed ? *mlp : none;
  const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  hipLaunchKernelGGL(sgd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, theta, grad, n, (float)lr, lr_dev, md, packed);
  return cvf_check_launch("sgd_kernel");
}
