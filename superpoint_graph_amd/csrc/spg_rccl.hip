// RCCL straight from the C library: the data-parallel collectives of the hot path -- ONE flat
// fp32 all-reduce of the gradient arena per step and, in the synchronised-BatchNorm mode, one small fp64 all-reduce per
// BatchNorm layer and direction -- are enqueued by this library on the caller's stream, with a communicator of its
// own.  No Python between a BatchNorm finalize kernel and its collective (round 1 bounced 26 sub-6 KB all-reduces per
// step through a ctypes callback).  RCCL is bound at run time (dlopen of the librccl the process already has --
// torch ships one -- or the system one), so libspg_hip.so has no link-time dependency on it and single-GPU users never
// load it.  Bootstrap: rank 0 calls spg_rccl_unique_id, the 128 bytes travel over whatever channel the host has
// (torch.distributed broadcast in superpoint_graph_amd/dist.py), every rank calls spg_rccl_init.
#include "../../include/spg_hip.h"
#include "spg_common.h"
#include "spg_gemm.h"
#include <dlfcn.h>

namespace {

typedef struct { char internal[128]; } UniqueId;       // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* Comm;                                     // ncclComm_t
enum { kInt64 = 4, kFloat32 = 7, kFloat64 = 8, kSum = 0 };          // ncclInt64, ncclFloat, ncclDouble, ncclSum (rccl.h)

struct Api {
  void* handle = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Api g_api;
Comm g_comm = nullptr;
int g_world = 0, g_rank = -1;

int load_api() {
  if (g_api.handle != nullptr) return 0;
  const char* names[] = {"librccl.so.1", "librccl.so"};
  void* h = nullptr;
  for (const char* n : names)
    if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) != nullptr) break;          // the copy the process already uses (torch's)
  for (const char* n : names)
    if (h == nullptr && (h = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
  if (h == nullptr) { spg_set_error("RCCL not found: %s", dlerror()); return -1; }
  g_api.GetUniqueId = (int (*)(UniqueId*))dlsym(h, "ncclGetUniqueId");
  g_api.CommInitRank = (int (*)(Comm*, int, UniqueId, int))dlsym(h, "ncclCommInitRank");
  g_api.AllReduce = (int (*)(const void*, void*, size_t, int, int, Comm, hipStream_t))dlsym(h, "ncclAllReduce");
  g_api.CommDestroy = (int (*)(Comm))dlsym(h, "ncclCommDestroy");
  g_api.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
  if (!g_api.GetUniqueId || !g_api.CommInitRank || !g_api.AllReduce || !g_api.CommDestroy) {
    spg_set_error("librccl lacks a required symbol");
    return -1;
  }
  g_api.handle = h;
  return 0;
}

int nccl_check(int rc, const char* what) {
  if (rc == 0) return 0;
  spg_set_error("%s failed: %s", what, g_api.GetErrorString ? g_api.GetErrorString(rc) : "?");
  return rc;
}

int bn_allreduce(void* /*ctx*/, double* buf, long n, void* stream) {
  return nccl_check(g_api.AllReduce(buf, buf, (size_t)n, kFloat64, kSum, g_comm, (hipStream_t)stream), "ncclAllReduce (BatchNorm sums)");
}

// slot-synchronised BatchNorm: the fixed-point statistics slots themselves, summed as 64-bit integers in place (exact)
int slot_allreduce(void* /*ctx*/, unsigned long long* words, long n, void* stream) {
  return nccl_check(g_api.AllReduce(words, words, (size_t)n, kInt64, kSum, g_comm, (hipStream_t)stream), "ncclAllReduce (statistics slots)");
}

}  // namespace

extern "C" int spg_rccl_unique_id(void* out_128_bytes) {
  SPG_CHECK_ARG(out_128_bytes != nullptr, "null pointer");
  SPG_TRY(load_api());
  UniqueId id;
  SPG_TRY(nccl_check(g_api.GetUniqueId(&id), "ncclGetUniqueId"));
  memcpy(out_128_bytes, &id, sizeof(id));
  return 0;
}

extern "C" int spg_rccl_init(const void* unique_id_128_bytes, int world_size, int rank) {
  SPG_CHECK_ARG(unique_id_128_bytes != nullptr && world_size >= 1 && rank >= 0 && rank < world_size, "bad argument");
  SPG_CHECK_ARG(g_comm == nullptr, "the library's RCCL communicator exists already (spg_rccl_destroy first)");
  SPG_TRY(load_api());
  UniqueId id;
  memcpy(&id, unique_id_128_bytes, sizeof(id));
  SPG_TRY(nccl_check(g_api.CommInitRank(&g_comm, world_size, id, rank), "ncclCommInitRank"));    // binds the current HIP device
  g_world = world_size; g_rank = rank;
  return 0;
}

extern "C" int spg_rccl_world_size(void) { return g_comm ? g_world : 0; }

extern "C" int spg_rccl_allreduce_sum_f32(float* buf, long n, void* stream) {
  SPG_CHECK_ARG(g_comm != nullptr, "spg_rccl_init has not been called");
  SPG_CHECK_ARG(buf != nullptr && n > 0, "bad argument");
  return nccl_check(g_api.AllReduce(buf, buf, (size_t)n, kFloat32, kSum, g_comm, (hipStream_t)stream), "ncclAllReduce");
}

extern "C" int spg_rccl_sync_bn(double* buf, long buf_doubles) {
  if (buf == nullptr) return spg_set_bn_allreduce(nullptr, nullptr, nullptr, 0);
  SPG_CHECK_ARG(g_comm != nullptr, "spg_rccl_init has not been called");
  return spg_set_bn_allreduce(&bn_allreduce, nullptr, buf, buf_doubles);
}

extern "C" int spg_rccl_sync_slots(int on) {
  if (!on) return spg_set_slot_allreduce(nullptr, nullptr, 1);
  SPG_CHECK_ARG(g_comm != nullptr, "spg_rccl_init has not been called");
  return spg_set_slot_allreduce(&slot_allreduce, nullptr, g_world);
}

extern "C" int spg_rccl_allreduce_sum_f64(double* buf, long n, void* stream) {
  SPG_CHECK_ARG(g_comm != nullptr, "spg_rccl_init has not been called");
  SPG_CHECK_ARG(buf != nullptr && n > 0, "bad argument");
  return nccl_check(g_api.AllReduce(buf, buf, (size_t)n, kFloat64, kSum, g_comm, (hipStream_t)stream), "ncclAllReduce");
}

extern "C" int spg_rccl_destroy(void) {
  if (g_comm != nullptr) {
    (void)spg_set_bn_allreduce(nullptr, nullptr, nullptr, 0);
    (void)spg_set_slot_allreduce(nullptr, nullptr, 1);
    const int rc = g_api.CommDestroy(g_comm);
    g_comm = nullptr; g_world = 0; g_rank = -1;
    return nccl_check(rc, "ncclCommDestroy");
  }
  return 0;
}
