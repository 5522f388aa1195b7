// RCCL behind the C ABI: capnet_allreduce_grads(comm, flat, count, stream)  (SURVEY 8b's export list).
//
// The reference has no multi-GPU code; this is the data-parallel step's one collective -- an in-place SUM
// all-reduce of the flat fp32 gradient buffer on a caller-given HIP stream (one process per GPU, rings over xGMI).
// librccl is NOT a link-time dependency: the functions are resolved at run time, first from an RCCL the process has
// already mapped (torch's own copy when the caller is a torch program: two RCCLs in one process would each want the
// GPUs' IPC handles), then from /opt/rocm. A host that never calls these never loads RCCL.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "common.h"
#include "kernels.h"

namespace capnet {

namespace {
struct Rccl {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r = [] {
    Rccl x;
    void* h = nullptr;
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names)
      if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);          // an RCCL this process already uses
    for (const char* n : names)
      if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) return x;
    x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(dlsym(h, "ncclAllReduce"));
    x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    x.ok = x.GetUniqueId && x.CommInitRank && x.AllReduce && x.CommDestroy && x.GetErrorString;
    return x;
  }();
  return r;
}

#define CAPNET_RCCL_CHECK(expr)                                                                       \
  do {                                                                                                \
    ncclResult_t _r = (expr);                                                                         \
    if (_r != ncclSuccess) {                                                                          \
      set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, rccl().GetErrorString(_r));        \
      return kErrHip;                                                                                 \
    }                                                                                                 \
  } while (0)
}  // namespace

struct Comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
};

static_assert(sizeof(ncclUniqueId) == 128, "capnet.h promises a 128-byte id");

int comm_unique_id(void* id128) {
  CAPNET_REQUIRE(id128, "comm_unique_id: null argument");
  CAPNET_REQUIRE(rccl().ok, "comm_unique_id: no usable librccl in this process or under /opt/rocm/lib");
  ncclUniqueId id;
  CAPNET_RCCL_CHECK(rccl().GetUniqueId(&id));
  memcpy(id128, &id, sizeof(id));
  return kOk;
}

int comm_create(const void* id128, int rank, int world, Comm** out) {
  CAPNET_REQUIRE(id128 && out && world >= 1 && rank >= 0 && rank < world, "comm_create: rank %d of %d", rank, world);
  CAPNET_REQUIRE(rccl().ok, "comm_create: no usable librccl in this process or under /opt/rocm/lib");
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  Comm* c = new Comm;
  c->rank = rank; c->world = world;
  ncclResult_t r = rccl().CommInitRank(&c->comm, world, id, rank);      // collective: every rank of the job calls it
  if (r != ncclSuccess) {
    set_error("comm_create: ncclCommInitRank failed: %s", rccl().GetErrorString(r));
    delete c;
    return kErrHip;
  }
  *out = c;
  return kOk;
}

int comm_destroy(Comm* c) {
  if (!c) return kOk;
  if (c->comm) CAPNET_RCCL_CHECK(rccl().CommDestroy(c->comm));
  delete c;
  return kOk;
}

int allreduce_grads(Comm* c, float* flat, long count, hipStream_t stream) {
  CAPNET_REQUIRE(c && c->comm && (count == 0 || flat) && count >= 0, "allreduce_grads: bad argument");
  if (count == 0) return kOk;
  CAPNET_RCCL_CHECK(rccl().AllReduce(flat, flat, (size_t)count, ncclFloat32, ncclSum, c->comm, stream));
  return kOk;
}

}  // namespace capnet
