// Device side of the peer-to-peer communicator (csrc/p2p.hip) for kernels that fold a cross-rank sum into their own launch:
// the finishing launch of the batch sums (collective #1 of the data-parallel step) and the slab reduction
// + Adam (collective #2).  Both messages are latency-bound (13-70 doubles; 6.6 k - 51 k floats), so the exchange is the
// "low-latency" form: every 4-byte payload travels as ONE naturally aligned 8-byte word {payload, tag} written by one store
// (untorn on this fabric), tag = the number of the exchange.  The receiver polls the word itself - no separate flag, no fence
// between payload and flag, no barrier between the sender's threads.  Every rank reads the `world` words of an element from
// its OWN window (its own share straight from its registers) and adds them in RANK ORDER: bit for bit the same sum on every rank (the ranks must agree on argsort(eig),
// core.py:432), whatever the arrival order.
//
//   LL window of a rank (fine-grained device memory, mapped by every peer through HIP IPC), in 8-byte words:
//     statistics region   [parity 2][source rank][cap_s]      cap_s = 2 * kMaxStats (a double = two payloads)
//     gradient region     [parity 2][source rank][cap_g]      cap_g = max_bytes / 4
//   Two parities: a rank can be ONE exchange ahead of a peer (it needs the peer's words of exchange e to finish e, and the peer
//   sends those only after it has read everything of e - 1), never two.  The exchange number lives on the device and is advanced
//   by the (one-workgroup) statistics exchange itself, so captured hipGraphs replay correctly; the gradient exchange of a step
//   carries the number of that step's statistics exchange (csrc/ef_mfma.hip slab_reduce_kernel).
//   All launches that use one communicator must be ordered on ONE stream (the numbers are read at kernel entry).
//   A peer whose word does not arrive within the communicator's time-out (default 20 s; CVF_P2P_TIMEOUT_MS) makes the kernel
//   write NaN into the result and set the communicator's error word (host-visible: cvf_p2p_error), which the shipped host
//   checks at every point where it reads results back.
#pragma once
#include "cvf_common.hpp"

constexpr int kP2PMaxWorld = 16;
constexpr int kP2PStatWords = 2 * 80;   // >= 2 * kMaxStats

struct P2PLL {
  int rank, world;                       // world == 0: no exchange (single process)
  unsigned long long* win[kP2PMaxWorld]; // this process's mapping of every rank's LL window (win[rank] = its own)
  long long cap_g;                       // payload words per rank in the gradient region
  unsigned* epoch;                       // device [2]: [0] statistics exchanges completed, [1] number the last gradient exchange used
  unsigned* ticket;                      // (unused)
  unsigned* error;                       // host-visible word: number of the exchange that timed out (0: none)
  unsigned long long timeout_ticks;      // of the 100 MHz s_memrealtime clock
};

__host__ __device__ inline long long p2p_ll_stat_words(int world) { return 2ll * world * kP2PStatWords; }
__host__ __device__ inline long long p2p_ll_words(int world, long long cap_g) { return p2p_ll_stat_words(world) + 2ll * world * cap_g; }

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
// word i of source rank `src` in region `stat ? statistics : gradient`, parity of exchange e, inside window `w`
__device__ __forceinline__ unsigned long long* p2p_ll_word(const P2PLL& d, unsigned long long* w, bool stat, unsigned e, int src, long long i) {
  const long long cap = stat ? kP2PStatWords : d.cap_g;
  const long long base = stat ? 0 : p2p_ll_stat_words(d.world);
  return w + base + ((long long)(e & 1u) * d.world + src) * cap + i;
}
// my payload i of exchange e into every PEER's window (my own share never leaves the registers / LDS: the sum below takes it from
// there, in its place of the rank order - a store -> load round trip through fine-grained memory is ~2 us)
__device__ __forceinline__ void p2p_ll_put(const P2PLL& d, bool stat, unsigned e, long long i, unsigned payload) {
  const unsigned long long word = ((unsigned long long)e << 32) | payload;
  for (int peer = 0; peer < d.world; ++peer)
    if (peer != d.rank)
      __hip_atomic_store(p2p_ll_word(d, d.win[peer], stat, e, d.rank, i), word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// payload i of source rank `src`, exchange e, from my own window; false: timed out (error word set)
__device__ __forceinline__ bool p2p_ll_get(const P2PLL& d, bool stat, unsigned e, int src, long long i, unsigned& payload) {
  const unsigned long long* p = p2p_ll_word(d, d.win[d.rank], stat, e, src, i);
  unsigned long long v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if ((unsigned)(v >> 32) != e) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    do {
      __builtin_amdgcn_s_sleep(4);
      v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if ((unsigned)(v >> 32) == e) break;
      if (__builtin_amdgcn_s_memrealtime() - t0 > d.timeout_ticks) {
        __hip_atomic_store(d.error, e != 0u ? e : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return false;
      }
    } while (true);
  }
  payload = (unsigned)v;
  return true;
}

// One workgroup: vec[0..n) (doubles in LDS, n <= kP2PStatWords / 2) <- sum over ranks, rank order.  `parts` is LDS scratch of
// world * 2 n unsigned.  Exchange number e = p2p_ll_next(d), stored back at the end (one workgroup per launch uses this region).
// Every thread of the workgroup must call it; ends with a barrier.
// the number of the statistics exchange a kernel is about to make: ask for it at kernel ENTRY (a global round trip that then
// hides behind the kernel's own work) and hand it to p2p_ll_allreduce_stats
__device__ __forceinline__ unsigned p2p_ll_next(const P2PLL& d) {
  return d.world > 0 ? __hip_atomic_load(d.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u : 0u;
}
__device__ inline void p2p_ll_allreduce_stats(const P2PLL& d, double* vec, int n, unsigned* parts, unsigned e) {
  const int tid = threadIdx.x, nt = blockDim.x;
  __syncthreads();                                       // vec complete
  for (int t = tid; t < 2 * n; t += nt) {
    const unsigned long long bits = __builtin_bit_cast(unsigned long long, vec[t >> 1]);
    p2p_ll_put(d, true, e, t, (unsigned)(t & 1 ? bits >> 32 : bits));
  }
  for (int t = tid; t < 2 * n * d.world; t += nt) {      // (rank, word) pairs over the threads: all polls in flight together
    const int src = t / (2 * n), i = t - src * 2 * n;
    unsigned v = 0x7ff80000u;                            // (high half of a NaN: what a time-out leaves)
    if (src == d.rank) {
      const unsigned long long bits = __builtin_bit_cast(unsigned long long, vec[i >> 1]);
      v = (unsigned)(i & 1 ? bits >> 32 : bits);
    } else if (!p2p_ll_get(d, true, e, src, i, v)) {
      v = 0x7ff80000u;
    }
    parts[t] = v;
  }
  __syncthreads();
  for (int i = tid; i < n; i += nt) {
    double acc = 0.0;
    for (int r = 0; r < d.world; ++r) {
      const unsigned lo = parts[r * 2 * n + 2 * i], hi = parts[r * 2 * n + 2 * i + 1];
      const double v = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
      acc = r == 0 ? v : acc + v;
    }
    vec[i] = acc;
  }
  if (tid == 0) __hip_atomic_store(d.epoch, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
}

// One thread: the sum over ranks (rank order) of element i of the gradient exchange e; my share is g.  The `world` words of the
// element are requested together (one round trip when the peers have already written), then the late ones are polled.
__device__ __forceinline__ float p2p_ll_allreduce_grad(const P2PLL& d, unsigned e, long long i, float g) {
  p2p_ll_put(d, false, e, i, __builtin_bit_cast(unsigned, g));
  unsigned long long v[kP2PMaxWorld];
#pragma unroll
  for (int r = 0; r < kP2PMaxWorld; ++r)
    v[r] = r < d.world && r != d.rank
               ? __hip_atomic_load(p2p_ll_word(d, d.win[d.rank], false, e, r, i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
               : (((unsigned long long)e << 32) | __builtin_bit_cast(unsigned, g));
  float acc = 0.0f;
  bool ok = true;
#pragma unroll
  for (int r = 0; r < kP2PMaxWorld; ++r) {
    if (r < d.world) {
      unsigned pay = (unsigned)v[r];
      if ((unsigned)(v[r] >> 32) != e) ok = p2p_ll_get(d, false, e, r, i, pay) && ok;
      const float x = __builtin_bit_cast(float, pay);
      acc = r == 0 ? x : acc + x;
    }
  }
  return ok ? acc : __builtin_nanf("");
}
#endif

// the communicator's device view (csrc/p2p.hip); nullptr + error message when `comm` is not a connected communicator
const P2PLL* cvf_p2p_ll(void* comm, int64_t n_grad);
