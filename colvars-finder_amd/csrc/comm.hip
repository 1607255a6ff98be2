// cvf_comm_*: the step's two cross-rank sums (the fp64 batch sums before the backward pass, the flat
// fp32 gradient after it) behind the C ABI, as RCCL all-reduces issued on the caller's stream - for hosts that bind the
// library without PyTorch (the shipped Python host goes through torch.distributed, whose "nccl" backend is the same RCCL).
// RCCL is looked up at run time (dlopen; an already loaded copy - e.g. PyTorch's - is reused), so the library itself carries no
// link-time dependency and loads on machines without it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>

#include "cvf_common.hpp"

namespace {

struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
};

const Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (!tried) {
    tried = true;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)   // a copy that is already in the process first
      if ((r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) != nullptr) break;
    for (const char* n : names) {
      if (r.lib != nullptr) break;
      r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    }
    if (r.lib != nullptr) {
      r.get_unique_id = reinterpret_cast<decltype(r.get_unique_id)>(dlsym(r.lib, "ncclGetUniqueId"));
      r.comm_init_rank = reinterpret_cast<decltype(r.comm_init_rank)>(dlsym(r.lib, "ncclCommInitRank"));
      r.all_reduce = reinterpret_cast<decltype(r.all_reduce)>(dlsym(r.lib, "ncclAllReduce"));
      r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(dlsym(r.lib, "ncclCommDestroy"));
      r.error_string = reinterpret_cast<decltype(r.error_string)>(dlsym(r.lib, "ncclGetErrorString"));
    }
  }
  return (r.lib && r.get_unique_id && r.comm_init_rank && r.all_reduce && r.comm_destroy) ? &r : nullptr;
}

int fail(const Rccl* r, const char* what, ncclResult_t rc) {
  cvf_set_error("%s: RCCL error %d (%s)", what, (int)rc, r->error_string ? r->error_string(rc) : "?");
  return -1;
}

}  // namespace

extern "C" int cvf_comm_unique_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

extern "C" int cvf_comm_unique_id(void* id_host) {
  const Rccl* r = rccl();
  CVF_REQUIRE(r != nullptr, "cvf_comm_unique_id: librccl.so not found");
  CVF_REQUIRE(id_host != nullptr, "cvf_comm_unique_id: bad argument");
  ncclUniqueId id;
  const ncclResult_t rc = r->get_unique_id(&id);
  if (rc != ncclSuccess) return fail(r, "cvf_comm_unique_id", rc);
  std::memcpy(id_host, &id, sizeof(id));
  return 0;
}

extern "C" int cvf_comm_init(void** comm, int rank, int world, const void* id_host) {
  const Rccl* r = rccl();
  CVF_REQUIRE(r != nullptr, "cvf_comm_init: librccl.so not found");
  CVF_REQUIRE(comm != nullptr && id_host != nullptr && world >= 1 && rank >= 0 && rank < world, "cvf_comm_init: bad argument");
  ncclUniqueId id;
  std::memcpy(&id, id_host, sizeof(id));
  ncclComm_t c = nullptr;
  const ncclResult_t rc = r->comm_init_rank(&c, world, id, rank);   // (uses the calling thread's current HIP device)
  if (rc != ncclSuccess) return fail(r, "cvf_comm_init", rc);
  *comm = c;
  return 0;
}

static int allreduce(void* comm, void* buf, int64_t n, ncclDataType_t dt, void* stream, const char* what) {
  const Rccl* r = rccl();
  CVF_REQUIRE(r != nullptr, "%s: librccl.so not found", what);
  CVF_REQUIRE(comm != nullptr && buf != nullptr && n > 0, "%s: bad argument", what);
  const ncclResult_t rc = r->all_reduce(buf, buf, (size_t)n, dt, ncclSum, (ncclComm_t)comm, (hipStream_t)stream);
  return rc == ncclSuccess ? 0 : fail(r, what, rc);
}

extern "C" int cvf_comm_allreduce_f64(void* comm, double* buf, int64_t n, void* stream) {
  return allreduce(comm, buf, n, ncclFloat64, stream, "cvf_comm_allreduce_f64");
}

extern "C" int cvf_comm_allreduce_f32(void* comm, float* buf, int64_t n, void* stream) {
  return allreduce(comm, buf, n, ncclFloat32, stream, "cvf_comm_allreduce_f32");
}

extern "C" int cvf_comm_destroy(void* comm) {
  const Rccl* r = rccl();
  CVF_REQUIRE(r != nullptr, "cvf_comm_destroy: librccl.so not found");
  if (comm == nullptr) return 0;
  const ncclResult_t rc = r->comm_destroy((ncclComm_t)comm);
  return rc == ncclSuccess ? 0 : fail(r, "cvf_comm_destroy", rc);
}
