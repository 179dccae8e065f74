// sgk_comm.hip -- the one collective of the path, in the C-ABI: the 16-word int64 episode-metrics vector (what track_metrics
// feeds its four meters, reference meters.py:66-84) all-reduced over the GPUs of a node with RCCL over xGMI -- SUM on [0..7],
// MAX on [8..11]. 64 B + 32 B per flush: pure latency, issued once per metrics flush, never per step; no board, state or
// Q-table byte ever crosses xGMI (the reference itself has no parallelism to bind: main.py:40-56 is Ray trial fan-out).
//
// RCCL is bound at run time (dlopen / dlsym of librccl.so.1): libsgk.so has no link-time dependency on it, a single-GPU user
// never loads it, and inside a PyTorch process the copy PyTorch already loaded is the one that is found.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <mutex>

#include "sgk_host_core.h"
#include "sgk_kernels.h"

namespace {

// the slice of rccl.h this file needs (values from /opt/rocm/include/rccl/rccl.h)
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { NCCL_SUCCESS = 0, NCCL_INT64 = 4, NCCL_SUM = 0, NCCL_MAX = 2 };

struct Rccl {
  void *dl = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;     // optional: what RCCL itself says the communicator spans
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;  // optional
  ncclResult_t (*GetVersion)(int *) = nullptr;                      // optional
  char error[256] = "";  // why RCCL cannot be used ("" = it can); a fixed buffer: nothing on this path allocates
};

void load_rccl(Rccl &r);

Rccl &rccl() {  // loaded once, whichever thread asks first
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] { load_rccl(r); });
  return r;
}

void load_rccl(Rccl &r) {
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *n : names) {
    r.dl = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (r.dl) break;
  }
  if (!r.dl) {
    const char *de = dlerror();  // one call: a second one returns NULL (the first clears the error)
    snprintf(r.error, sizeof(r.error), "librccl.so.1 could not be loaded: %s", de ? de : "?");
    return;
  }
#define SGK_SYM(field, name)                                                             \
  do {                                                                                   \
    *reinterpret_cast<void **>(&r.field) = dlsym(r.dl, name);                            \
    if (!r.field) { snprintf(r.error, sizeof(r.error), "librccl has no symbol %s", name); return; } \
  } while (0)
  SGK_SYM(GetUniqueId, "ncclGetUniqueId");
  SGK_SYM(CommInitRank, "ncclCommInitRank");
  SGK_SYM(CommDestroy, "ncclCommDestroy");
  SGK_SYM(AllReduce, "ncclAllReduce");
  SGK_SYM(GroupStart, "ncclGroupStart");
  SGK_SYM(GroupEnd, "ncclGroupEnd");
  SGK_SYM(GetErrorString, "ncclGetErrorString");
#undef SGK_SYM
  *reinterpret_cast<void **>(&r.CommCount) = dlsym(r.dl, "ncclCommCount");
  *reinterpret_cast<void **>(&r.CommUserRank) = dlsym(r.dl, "ncclCommUserRank");
  *reinterpret_cast<void **>(&r.GetVersion) = dlsym(r.dl, "ncclGetVersion");
}

__global__ void set_word_kernel(long long *p, long long v) { *p = v; }

}  // namespace

struct sgk_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
};

extern "C" {

int sgk_set_error(int code, const char *msg);  // sgk_api.hip: stores the thread's last error, returns code

static int rccl_fail(const char *what, ncclResult_t e) {
  Rccl &r = rccl();
  return sgk::host::fail(SGK_ERR_HIP, "%s: %s", what, r.GetErrorString ? r.GetErrorString(e) : "RCCL error");
}

int sgk_comm_available(int32_t *version_out) try {
  Rccl &r = rccl();  // dlopen + dlsym of every entry point the path needs: no socket, no thread, nothing to clean up
  if (r.error[0]) return sgk_set_error(SGK_ERR_NODEVICE, r.error);
  int v = 0;
  if (r.GetVersion && r.GetVersion(&v) != NCCL_SUCCESS) v = 0;
  if (version_out) *version_out = v;
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_comm_unique_id(uint8_t id_out[SGK_COMM_ID_BYTES]) try {
  if (!id_out) return sgk_set_error(SGK_ERR_INVALID, "id_out is NULL");
  Rccl &r = rccl();
  if (r.error[0]) return sgk_set_error(SGK_ERR_NODEVICE, r.error);
  ncclUniqueId id;
  ncclResult_t e = r.GetUniqueId(&id);
  if (e != NCCL_SUCCESS) return rccl_fail("ncclGetUniqueId", e);
  static_assert(sizeof(id) == SGK_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  std::memcpy(id_out, &id, sizeof(id));
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_comm_create(const uint8_t id[SGK_COMM_ID_BYTES], int rank, int world_size, int device, sgk_comm **out) try {
  if (!out) return sgk_set_error(SGK_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (!id || world_size < 1 || rank < 0 || rank >= world_size) return sgk_set_error(SGK_ERR_INVALID, "bad id / rank / world_size");
  Rccl &r = rccl();
  if (r.error[0]) return sgk_set_error(SGK_ERR_NODEVICE, r.error);
  hipError_t he = hipSetDevice(device);
  if (he != hipSuccess) return sgk_set_error(SGK_ERR_HIP, hipGetErrorString(he));
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  sgk_comm *c = new (std::nothrow) sgk_comm();
  if (!c) return sgk_set_error(SGK_ERR_NOMEM, "host allocation failed");
  ncclResult_t e = r.CommInitRank(&c->comm, world_size, uid, rank);
  if (e != NCCL_SUCCESS) {
    delete c;
    return rccl_fail("ncclCommInitRank", e);
  }
  c->rank = rank;
  c->world = world_size;
  c->device = device;
  *out = c;
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_comm_destroy(sgk_comm *c) try {
  if (!c) return SGK_OK;
  Rccl &r = rccl();
  if (c->comm && r.CommDestroy) (void)r.CommDestroy(c->comm);
  delete c;
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_comm_info(const sgk_comm *c, int32_t *rank_out, int32_t *world_out, int32_t *device_out) try {
  if (!c) return sgk_set_error(SGK_ERR_INVALID, "comm is NULL");
  int rank = c->rank, world = c->world;
  Rccl &r = rccl();
  // RCCL's own view where the library offers it (a check that the communicator spans what the caller meant it to)
  if (c->comm && r.CommCount && r.CommCount(c->comm, &world) != NCCL_SUCCESS) world = c->world;
  if (c->comm && r.CommUserRank && r.CommUserRank(c->comm, &rank) != NCCL_SUCCESS) rank = c->rank;
  if (rank_out) *rank_out = rank;
  if (world_out) *world_out = world;
  if (device_out) *device_out = c->device;
  return SGK_OK;
} SGK_CATCH_STATUS

int sgk_allreduce_metrics(sgk_comm *c, int64_t *inout_dev, void *hip_stream) try {
  if (!c || !inout_dev) return sgk_set_error(SGK_ERR_INVALID, "NULL argument");
  Rccl &r = rccl();
  hipError_t he = hipSetDevice(c->device);
  if (he != hipSuccess) return sgk_set_error(SGK_ERR_HIP, hipGetErrorString(he));
  hipStream_t st = (hipStream_t)hip_stream;
  // two reductions in one group: sums / counts, then maxima (words 12..15 are not exchanged)
  ncclResult_t e = r.GroupStart();
  if (e == NCCL_SUCCESS) e = r.AllReduce(inout_dev, inout_dev, 8, NCCL_INT64, NCCL_SUM, c->comm, st);
  if (e == NCCL_SUCCESS) e = r.AllReduce(inout_dev + SGK_M_MAX_RETURN, inout_dev + SGK_M_MAX_RETURN, 4, NCCL_INT64, NCCL_MAX, c->comm, st);
  ncclResult_t ge = r.GroupEnd();
  if (e != NCCL_SUCCESS) return rccl_fail("ncclAllReduce", e);
  if (ge != NCCL_SUCCESS) return rccl_fail("ncclGroupEnd", ge);
  return SGK_OK;
} SGK_CATCH_STATUS

// sgk_metrics() of this shard all-reduced over the communicator's ranks: defined in sgk_api.hip (needs the handle's internals)

}  // extern "C"
