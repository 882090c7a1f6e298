// Data-parallel collective of the training step through the C ABI (nq_allreduce(buf, n, comm, stream)).
// Reference semantics: Lightning DDPStrategy's gradient all-reduce (nablaDFT/utils/pipelines.py:65-68) -- the flat gradient buffer is
// summed over the ranks (one process per GPU) and divided by the world size; nothing else is ever exchanged (one conformer = one graph).
// RCCL is bound at run time (dlopen of librccl.so: the library torch.distributed's "nccl" backend uses on ROCm), so libnablaq.so has no
// link-time dependency on it and a single-GPU user never loads it.  The collectives are enqueued on the caller's HIP stream: they are
// ordered after the kernels that produced the buffer and before the optimiser kernel that consumes it, with no host synchronisation.
#include "common.h"
#include <dlfcn.h>
#include <cstring>
#include <rccl/rccl.h>
#include <mutex>

namespace {
struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  const char* error = "not loaded";
};
RcclApi g_api;
std::once_flag g_once;

void load_api() {
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    g_api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (g_api.handle) break;
  }
  if (!g_api.handle) { g_api.error = "librccl.so not found (dlopen)"; return; }
#define NQ_SYM(field, sym)                                                     \
  g_api.field = reinterpret_cast<decltype(g_api.field)>(dlsym(g_api.handle, sym)); \
  if (!g_api.field) { g_api.error = "librccl.so lacks " sym; g_api.handle = nullptr; return; }
  NQ_SYM(GetUniqueId, "ncclGetUniqueId")
  NQ_SYM(CommInitRank, "ncclCommInitRank")
  NQ_SYM(CommDestroy, "ncclCommDestroy")
  NQ_SYM(AllReduce, "ncclAllReduce")
  NQ_SYM(Broadcast, "ncclBroadcast")
  NQ_SYM(CommCount, "ncclCommCount")
  NQ_SYM(GetErrorString, "ncclGetErrorString")
#undef NQ_SYM
  g_api.error = nullptr;
}
int need_api() {
  std::call_once(g_once, load_api);
  return g_api.handle ? NQ_OK : nq_fail(NQ_ERR_ARG, "RCCL unavailable: %s", g_api.error);
}
#define NQ_RCCL(call)                                                                                   \
  do {                                                                                                  \
    ncclResult_t r__ = (call);                                                                          \
    if (r__ != ncclSuccess) return nq_fail(NQ_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call, g_api.GetErrorString(r__)); \
  } while (0)

__global__ void k_scale_inplace(float* __restrict__ x, size_t count, float s) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) x[i] *= s;
}
}  // namespace

extern "C" {

int nq_rccl_available(void) { return need_api() == NQ_OK ? 1 : 0; }

// 128 bytes that rank 0 creates and every rank of the job must receive (through any side channel: torch.distributed's store, MPI, a file)
int nq_rccl_unique_id(void* id128) {
  NQ_TRY(need_api());
  if (!id128) return nq_fail(NQ_ERR_ARG, "null id buffer");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  NQ_RCCL(g_api.GetUniqueId(reinterpret_cast<ncclUniqueId*>(id128)));
  return NQ_OK;
}

// Collective call: every rank of the job calls it with the same id; the communicator is bound to the calling thread's current HIP device.
int nq_rccl_comm_create(const void* id128, int32_t world, int32_t rank, void** comm) {
  NQ_TRY(need_api());
  if (!id128 || !comm || world < 1 || rank < 0 || rank >= world) return nq_fail(NQ_ERR_ARG, "bad communicator arguments (world %d, rank %d)", world, rank);
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  NQ_RCCL(g_api.CommInitRank(&c, world, id, rank));
  *comm = c;
  return NQ_OK;
}

int nq_rccl_comm_destroy(void* comm) {
  NQ_TRY(need_api());
  if (comm) NQ_RCCL(g_api.CommDestroy(reinterpret_cast<ncclComm_t>(comm)));
  return NQ_OK;
}

// number of ranks in the communicator as RCCL reports it (ncclCommCount)
int nq_rccl_comm_count(void* comm, int32_t* count) {
  NQ_TRY(need_api());
  if (!comm || !count) return nq_fail(NQ_ERR_ARG, "null communicator or output");
  int world = 0;
  NQ_RCCL(g_api.CommCount(reinterpret_cast<ncclComm_t>(comm), &world));
  *count = world;
  return NQ_OK;
}

// buf <- sum over ranks of buf (fp32, in place), enqueued on `stream`
int nq_allreduce(float* buf, size_t n, void* comm, void* stream) {
  NQ_TRY(need_api());
  if (!comm || (!buf && n)) return nq_fail(NQ_ERR_ARG, "null communicator or buffer");
  if (n == 0) return NQ_OK;
  NQ_RCCL(g_api.AllReduce(buf, buf, n, ncclFloat32, ncclSum, reinterpret_cast<ncclComm_t>(comm), (hipStream_t)stream));
  return NQ_OK;
}

// buf <- mean over ranks of buf: the gradient averaging of a data-parallel step (sum, then one scaling kernel on the same stream)
int nq_allreduce_mean(float* buf, size_t n, void* comm, void* stream) {
  NQ_TRY(nq_allreduce(buf, n, comm, stream));
  int world = 1;
  NQ_RCCL(g_api.CommCount(reinterpret_cast<ncclComm_t>(comm), &world));
  if (world > 1 && n) {
    hipLaunchKernelGGL(k_scale_inplace, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, buf, n, 1.0f / (float)world);
    NQ_LAUNCH_CHECK();
  }
  return NQ_OK;
}

// buf of rank `root` -> every rank (initial parameters)
int nq_rccl_broadcast(float* buf, size_t n, int32_t root, void* comm, void* stream) {
  NQ_TRY(need_api());
  if (!comm || (!buf && n)) return nq_fail(NQ_ERR_ARG, "null communicator or buffer");
  if (n == 0) return NQ_OK;
  NQ_RCCL(g_api.Broadcast(buf, buf, n, ncclFloat32, root, reinterpret_cast<ncclComm_t>(comm), (hipStream_t)stream));
  return NQ_OK;
}

}  // extern "C"
