// cvf_p2p_*: the step's two cross-rank sums as a ONE-SHOT peer-to-peer reduce ("RCCL or one-shot P2P
// write + fixed-order local reduce").  Both messages are tiny (13-70 doubles of batch sums; 6.6 k - 51 k floats of gradient) and
// latency-bound; the 8 GPUs of a node are fully connected over xGMI.  A ring all-reduce takes 2 (N - 1) dependent hops; here
// every rank writes its vector straight into a slot of EVERY peer's window (one hop, all links at once), raises a flag there, waits
// for the N flags in its own window and adds the N slots in rank order - so every rank forms bit-for-bit the same sum (the ranks
// must agree on argsort(eig), core.py:432), whatever the arrival order.
//
//   window (one per rank, fine-grained device memory shared with the peers through HIP IPC handles):
//     for parity p in {0, 1}:  flags[p][world] (uint32: the epoch whose data slot r holds) | slots[p][world][slot_bytes]
//   The epoch lives on the device and is advanced by the kernel itself, so a captured hipGraph replays correctly.  Two
//   parities: rank A can be one all-reduce ahead of rank B (it cannot be two: its next one needs B's flag of that epoch), and
//   then writes the other half.
//   Cross-device visibility: the window is FINE-GRAINED memory; payload and flags are written and read with system-scope relaxed
//   atomics (global_store / global_load ... sc0 sc1), every writing thread drains its stores (s_waitcnt vmcnt(0)) and the block
//   meets at a barrier before the flags go out; the reader polls the flags with system-scope loads and orders the slot reads
//   behind them.  Every spin is bounded (s_memrealtime; 20 s by default, CVF_P2P_TIMEOUT_MS): a peer that never arrives makes the
//   kernel fill `buf` with NaN and set the communicator's error word.  The word lives in host-visible memory: cvf_p2p_error()
//   reads it without touching the device, and the shipped host does so wherever it reads results back (the epoch log, before
//   save_model, in loss_func) and raises - a time-out cannot go unnoticed, and the NaN poisons every number derived from the
//   unreduced vector.  After an error the communicator is dead (the two-parity protocol assumes no rank skips an exchange).
//   All launches on one communicator must be ordered on ONE stream: the exchange number is read at kernel entry.
//
// The same window carries the LOW-LATENCY exchange (cvf_p2p.hpp: 8-byte {payload, tag} words, no flags) that the finishing launch
// of the batch sums and the slab reduction fold into their own kernels (cvf_ef16_finish_dp, cvf_ef_loss_dp, cvf_slab_reduce_dp).
//
// The host hands the handles around by any means (the shipped Python host: torch.distributed all_gather over the existing group).
#include <cstring>
#include <new>

#include <cstdlib>

#include "cvf_common.hpp"
#include "cvf_p2p.hpp"

namespace {

constexpr int kP2PThreads = 1024;

struct P2PDev {
  int rank, world;
  int64_t slot_bytes;          // bytes of one rank's slot (a multiple of 256)
  char* win[kP2PMaxWorld];     // this process's mapping of every rank's window (win[rank] = its own)
  unsigned* epoch;             // device: all-reduces completed so far (private to the rank)
  unsigned* error;             // host-visible: set to the epoch that timed out
  unsigned long long timeout_ticks;
};

struct P2PComm {
  P2PDev d;
  void* own_window = nullptr;
  void* peer_window[kP2PMaxWorld] = {};
  unsigned* state = nullptr;   // device: [epoch of the flag protocol, LL epoch statistics, LL epoch gradient, LL ticket]
  unsigned* error_host = nullptr;   // page-locked, mapped: the error word as the host reads it
  int64_t max_bytes = 0;
  int64_t flag_window_bytes = 0;    // the LL words follow the flag-protocol window in the same allocation
  P2PLL ll = {};
  bool connected = false;
};

__host__ __device__ inline int64_t p2p_flags_bytes(int world) { return ((int64_t)world * 4 + 255) & ~(int64_t)255; }
__host__ __device__ inline int64_t p2p_half_bytes(int world, int64_t slot_bytes) { return p2p_flags_bytes(world) + world * slot_bytes; }
inline int64_t p2p_window_bytes(int world, int64_t slot_bytes) { return 2 * p2p_half_bytes(world, slot_bytes); }

template <class T>
__device__ __forceinline__ void store_sys(T* p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <class T>
__device__ __forceinline__ T load_sys(const T* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// buf <- sum over ranks of buf, in rank order.  ONE workgroup (the messages are a few KB to 200 KB).
template <class T, class U>   // U: the unsigned integer type of T's size (the atomics move bit patterns)
__global__ __launch_bounds__(kP2PThreads) void p2p_allreduce_kernel(P2PDev d, T* __restrict__ buf, int64_t n) {
  const int tid = threadIdx.x;
  const unsigned e = load_sys(d.epoch) + 1u;            // this all-reduce's number (never 0 in a flag)
  const int64_t half = (e & 1u) * p2p_half_bytes(d.world, d.slot_bytes);
  const int64_t fb = p2p_flags_bytes(d.world);
  // ---- 1. my vector into slot [rank] of every window (my own included: one code path, one summation order)
  for (int peer = 0; peer < d.world; ++peer) {
    U* dst = reinterpret_cast<U*>(d.win[peer] + half + fb + d.rank * d.slot_bytes);
    for (int64_t i = tid; i < n; i += kP2PThreads) store_sys(dst + i, __builtin_bit_cast(U, buf[i]));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every writing thread: its stores have left
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");         // system scope
  __syncthreads();
  if (tid < d.world) store_sys(reinterpret_cast<unsigned*>(d.win[tid] + half) + d.rank, e);   // "slot [rank] of your window holds epoch e"
  // ---- 2. wait for every rank's flag in MY window
  __shared__ int s_ok;
  if (tid == 0) s_ok = 1;
  __syncthreads();
  if (tid < d.world) {
    const unsigned* flag = reinterpret_cast<const unsigned*>(d.win[d.rank] + half) + tid;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool ok = true;
    while (load_sys(flag) != e) {
      __builtin_amdgcn_s_sleep(8);
      if (__builtin_amdgcn_s_memrealtime() - t0 > d.timeout_ticks) {
        ok = false;
        break;
      }
    }
    if (!ok) {
      s_ok = 0;
      store_sys(d.error, e != 0u ? e : 1u);
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");         // system scope: the slot reads below stay behind the flags
  // ---- 3. the sum, rank order (a rank that timed out poisons buf with NaN: the error word says why)
  if (!s_ok) {
    for (int64_t i = tid; i < n; i += kP2PThreads) buf[i] = (T)__builtin_nan("");
  } else {
    const char* mine = d.win[d.rank] + half + fb;
    for (int64_t i = tid; i < n; i += kP2PThreads) {
      T acc = __builtin_bit_cast(T, load_sys(reinterpret_cast<const U*>(mine) + i));
      for (int r = 1; r < d.world; ++r)
        acc += __builtin_bit_cast(T, load_sys(reinterpret_cast<const U*>(mine + r * d.slot_bytes) + i));
      buf[i] = acc;
    }
  }
  __syncthreads();
  if (tid == 0) store_sys(d.epoch, e);
}

// buf[0..n) (doubles) <- sum over ranks, rank order, as ONE low-latency exchange of the statistics region (cvf_p2p.hpp)
__global__ __launch_bounds__(256) void p2p_ll_exchange_f64_kernel(P2PLL ll, double* __restrict__ buf, int n) {
  __shared__ double vec[kP2PStatWords / 2];
  __shared__ unsigned parts[kP2PMaxWorld * kP2PStatWords];
  const unsigned ex = p2p_ll_next(ll);
  for (int i = threadIdx.x; i < n; i += blockDim.x) vec[i] = buf[i];
  p2p_ll_allreduce_stats(ll, vec, n, parts, ex);
  for (int i = threadIdx.x; i < n; i += blockDim.x) buf[i] = vec[i];
}

int hip_fail(const char* what, hipError_t e) {
  cvf_set_error("%s: %s", what, hipGetErrorString(e));
  return -1;
}

template <class T, class U>
int p2p_allreduce(void* comm, T* buf, int64_t n, void* stream, const char* what) {
  P2PComm* c = static_cast<P2PComm*>(comm);
  CVF_REQUIRE(c != nullptr && buf != nullptr && n > 0, "%s: bad argument", what);
  CVF_REQUIRE(c->connected, "%s: cvf_p2p_connect has not been called", what);
  CVF_REQUIRE(n * (int64_t)sizeof(T) <= c->d.slot_bytes, "%s: %lld elements do not fit the window's %lld-byte slots", what, (long long)n,
              (long long)c->d.slot_bytes);
  hipLaunchKernelGGL((p2p_allreduce_kernel<T, U>), dim3(1), dim3(kP2PThreads), 0, (hipStream_t)stream, c->d, buf, n);
  return cvf_check_launch("p2p_allreduce_kernel");
}

}  // namespace

extern "C" int cvf_p2p_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

extern "C" int cvf_p2p_create(void** comm, int rank, int world, int64_t max_bytes, void* handle_out_host) {
  CVF_REQUIRE(comm != nullptr && handle_out_host != nullptr && world >= 1 && world <= kP2PMaxWorld && rank >= 0 && rank < world && max_bytes > 0,
              "cvf_p2p_create: bad argument (1 <= world <= %d)", kP2PMaxWorld);
  P2PComm* c = new (std::nothrow) P2PComm();
  CVF_REQUIRE(c != nullptr, "cvf_p2p_create: out of host memory");
  c->d.rank = rank;
  c->d.world = world;
  c->d.slot_bytes = (max_bytes + 255) & ~(int64_t)255;
  c->max_bytes = max_bytes;
  c->flag_window_bytes = p2p_window_bytes(world, c->d.slot_bytes);
  const long long cap_g = (long long)((max_bytes + 3) / 4);
  const size_t bytes = (size_t)c->flag_window_bytes + 8 * (size_t)p2p_ll_words(world, cap_g);
  const char* tmo = getenv("CVF_P2P_TIMEOUT_MS");
  const unsigned long long ticks = (tmo != nullptr && atoll(tmo) > 0 ? (unsigned long long)atoll(tmo) : 20000ull) * 100000ull;   // 100 MHz clock
  // fine-grained device memory: writes of a peer (over xGMI) and reads of the owner are coherent without a kernel boundary
  hipError_t e = hipExtMallocWithFlags(&c->own_window, bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) { delete c; return hip_fail("cvf_p2p_create: hipExtMallocWithFlags", e); }
  if ((e = hipMemset(c->own_window, 0, bytes)) != hipSuccess) { (void)hipFree(c->own_window); delete c; return hip_fail("cvf_p2p_create: hipMemset", e); }
  if ((e = hipMalloc(reinterpret_cast<void**>(&c->state), 256)) != hipSuccess || (e = hipMemset(c->state, 0, 256)) != hipSuccess) {
    (void)hipFree(c->own_window); delete c; return hip_fail("cvf_p2p_create: state", e);
  }
  if ((e = hipDeviceSynchronize()) != hipSuccess) { (void)hipFree(c->own_window); (void)hipFree(c->state); delete c; return hip_fail("cvf_p2p_create", e); }
  void* err_dev = nullptr;
  if ((e = hipHostMalloc(reinterpret_cast<void**>(&c->error_host), 64, hipHostMallocMapped)) != hipSuccess ||
      (e = hipHostGetDevicePointer(&err_dev, c->error_host, 0)) != hipSuccess) {
    (void)hipFree(c->own_window); (void)hipFree(c->state); delete c;
    return hip_fail("cvf_p2p_create: error word", e);
  }
  *c->error_host = 0u;
  c->d.epoch = c->state;
  c->d.error = static_cast<unsigned*>(err_dev);
  c->d.timeout_ticks = ticks;
  c->ll.rank = rank;
  c->ll.world = world;
  c->ll.cap_g = cap_g;
  c->ll.epoch = c->state + 1;
  c->ll.ticket = c->state + 3;
  c->ll.error = c->d.error;
  c->ll.timeout_ticks = ticks;
  hipIpcMemHandle_t h;
  if ((e = hipIpcGetMemHandle(&h, c->own_window)) != hipSuccess) {
    (void)hipFree(c->own_window); (void)hipFree(c->state); delete c;
    return hip_fail("cvf_p2p_create: hipIpcGetMemHandle", e);
  }
  std::memcpy(handle_out_host, &h, sizeof(h));
  *comm = c;
  return 0;
}

extern "C" int cvf_p2p_connect(void* comm, const void* all_handles_host) {
  P2PComm* c = static_cast<P2PComm*>(comm);
  CVF_REQUIRE(c != nullptr && all_handles_host != nullptr && !c->connected, "cvf_p2p_connect: bad argument");
  for (int r = 0; r < c->d.world; ++r) {
    if (r == c->d.rank) {
      c->d.win[r] = static_cast<char*>(c->own_window);
      continue;
    }
    hipIpcMemHandle_t h;
    std::memcpy(&h, static_cast<const char*>(all_handles_host) + (size_t)r * sizeof(h), sizeof(h));
    void* p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return hip_fail("cvf_p2p_connect: hipIpcOpenMemHandle", e);
    c->peer_window[r] = p;
    c->d.win[r] = static_cast<char*>(p);
  }
  for (int r = 0; r < c->d.world; ++r) c->ll.win[r] = reinterpret_cast<unsigned long long*>(c->d.win[r] + c->flag_window_bytes);
  c->connected = true;
  return 0;
}

extern "C" int cvf_p2p_allreduce_f64(void* comm, double* buf, int64_t n, void* stream) {
  return p2p_allreduce<double, unsigned long long>(comm, buf, n, stream, "cvf_p2p_allreduce_f64");
}
extern "C" int cvf_p2p_allreduce_f32(void* comm, float* buf, int64_t n, void* stream) {
  return p2p_allreduce<float, unsigned>(comm, buf, n, stream, "cvf_p2p_allreduce_f32");
}

// 0, or the number of the exchange in which a peer's data did not arrive within the time-out.  Reads a host-visible word: no
// device call, no synchronisation - what it reports is what the kernels that have FINISHED by now have found.
extern "C" int cvf_p2p_error(void* comm) {
  P2PComm* c = static_cast<P2PComm*>(comm);
  CVF_REQUIRE(c != nullptr, "cvf_p2p_error: bad argument");
  const unsigned v = *static_cast<volatile unsigned*>(c->error_host);
  return (int)(v & 0x7fffffffu);
}

// small fp64 vectors (n <= 80: batch sums, loss terms) over the low-latency words instead of the flag protocol: one hop, no fence
extern "C" int cvf_p2p_exchange_f64(void* comm, double* buf, int64_t n, void* stream) {
  P2PComm* c = static_cast<P2PComm*>(comm);
  CVF_REQUIRE(c != nullptr && buf != nullptr && n > 0 && n <= kP2PStatWords / 2, "cvf_p2p_exchange_f64: bad argument (1 <= n <= %d)", kP2PStatWords / 2);
  CVF_REQUIRE(c->connected, "cvf_p2p_exchange_f64: cvf_p2p_connect has not been called");
  hipLaunchKernelGGL(p2p_ll_exchange_f64_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, c->ll, buf, (int)n);
  return cvf_check_launch("p2p_ll_exchange_f64_kernel");
}

// the communicator's device view for the kernels that fold an exchange into their own launch (cvf_p2p.hpp)
const P2PLL* cvf_p2p_ll(void* comm, int64_t n_grad) {
  P2PComm* c = static_cast<P2PComm*>(comm);
  if (c == nullptr || !c->connected) {
    cvf_set_error("p2p exchange: the communicator is not connected (cvf_p2p_create + cvf_p2p_connect)");
    return nullptr;
  }
  if (n_grad > c->ll.cap_g) {
    cvf_set_error("p2p exchange: %lld gradient entries do not fit the window's %lld", (long long)n_grad, (long long)c->ll.cap_g);
    return nullptr;
  }
  return &c->ll;
}

extern "C" int cvf_p2p_destroy(void* comm) {
  P2PComm* c = static_cast<P2PComm*>(comm);
  if (c == nullptr) return 0;
  (void)hipDeviceSynchronize();
  for (int r = 0; r < c->d.world; ++r)
    if (c->peer_window[r] != nullptr) (void)hipIpcCloseMemHandle(c->peer_window[r]);
  if (c->own_window != nullptr) (void)hipFree(c->own_window);
  if (c->state != nullptr) (void)hipFree(c->state);
  if (c->error_host != nullptr) (void)hipHostFree(c->error_host);
  delete c;
  return 0;
}
