// Data-parallel gradient exchange behind the C ABI (focr_comm_init / focr_allreduce_async /
// focr_comm_destroy): one RCCL communicator per process (= per GPU), in-place sum all-reduce of a gradient range on a
// caller-chosen stream.  This replaces what nn.DataParallel's replicate / scatter / gather does for the gradients
// (reference interfaces/base.py:178-179) and is the same call torch.distributed's "nccl" backend ends in; the engine
// uses torch.distributed by default and this entry when FOCR_COMM=native.
//
// RCCL is bound at run time (dlopen of librccl.so.1): a process that already holds RCCL (PyTorch loads its own copy)
// gets THAT instance back, and the library has no link-time dependency on it -- single-GPU users never load it.
#include <dlfcn.h>
#include <chrono>
#include <thread>
#include <rccl/rccl.h>
#include "focr_common.h"

namespace {
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
} g_rccl;
ncclComm_t g_comm = nullptr;
int g_nranks = 0;

int load_rccl() {
  if (g_rccl.h) return FOCR_OK;
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    focr_set_error("focr_comm: cannot load librccl.so (%s)", dlerror());
    return FOCR_EUNSUPPORTED;
  }
  g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(dlsym(h, "ncclAllReduce"));
  g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  g_rccl.CommGetAsyncError = reinterpret_cast<decltype(g_rccl.CommGetAsyncError)>(dlsym(h, "ncclCommGetAsyncError"));
  g_rccl.CommAbort = reinterpret_cast<decltype(g_rccl.CommAbort)>(dlsym(h, "ncclCommAbort"));
  g_rccl.CommCount = reinterpret_cast<decltype(g_rccl.CommCount)>(dlsym(h, "ncclCommCount"));
  g_rccl.GetVersion = reinterpret_cast<decltype(g_rccl.GetVersion)>(dlsym(h, "ncclGetVersion"));
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy) {
    focr_set_error("focr_comm: librccl.so lacks an expected symbol");
    dlclose(h);
    return FOCR_EUNSUPPORTED;
  }
  g_rccl.h = h;
  return FOCR_OK;
}
int check(ncclResult_t r, const char* what) {
  if (r == ncclSuccess) return FOCR_OK;
  focr_set_error("focr_comm: %s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error");
  return FOCR_ENCCL;
}
// asynchronous errors of the communicator (a peer died, a link went down: RCCL reports them out of band, the collective
// itself just never completes)
int async_error() {
  if (!g_comm || !g_rccl.CommGetAsyncError) return FOCR_OK;
  ncclResult_t st = ncclSuccess;
  ncclResult_t r = g_rccl.CommGetAsyncError(g_comm, &st);
  if (r != ncclSuccess) return check(r, "ncclCommGetAsyncError");
  if (st != ncclSuccess && st != ncclInProgress) return check(st, "communicator (asynchronous error)");
  return FOCR_OK;
}
}  // namespace

// rank 0 draws the id (FOCR_COMM_ID_BYTES = sizeof(ncclUniqueId) = 128 bytes) and hands it to the other ranks by any
// host channel (the engine broadcasts it through the process group's store)
extern "C" int focr_comm_unique_id(void* id_out) {
  FOCR_CHECK_ARG(id_out, "null pointer");
  int rc = load_rccl();
  if (rc != FOCR_OK) return rc;
  return check(g_rccl.GetUniqueId(reinterpret_cast<ncclUniqueId*>(id_out)), "ncclGetUniqueId");
}
// collective over all ranks; the calling thread's current HIP device is the rank's GPU
extern "C" int focr_comm_init(int rank, int nranks, const void* unique_id) {
  FOCR_CHECK_ARG(unique_id && nranks >= 1 && rank >= 0 && rank < nranks, "bad argument");
  if (g_comm) {
    focr_set_error("focr_comm_init: a communicator already exists (focr_comm_destroy first)");
    return FOCR_EINVAL;
  }
  int rc = load_rccl();
  if (rc != FOCR_OK) return rc;
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  rc = check(g_rccl.CommInitRank(&g_comm, nranks, id, rank), "ncclCommInitRank");
  if (rc == FOCR_OK) g_nranks = nranks;
  return rc;
}
// buf[0..n) <- sum over ranks, in place, asynchronous on `stream`.  dtype: 0 = fp32 (the gradient buffers).
extern "C" int focr_allreduce_async(void* buf, size_t n, int dtype, hipStream_t stream) {
  FOCR_CHECK_ARG(buf && n > 0, "bad argument");
  FOCR_CHECK_ARG(dtype == 0, "only fp32 (dtype 0) gradient buffers are exchanged");
  if (!g_comm) {
    focr_set_error("focr_allreduce_async: no communicator (focr_comm_init first)");
    return FOCR_EINVAL;
  }
  int rc = async_error();             // an earlier collective failed asynchronously: do not queue behind it
  if (rc != FOCR_OK) return rc;
  return check(g_rccl.AllReduce(buf, buf, n, ncclFloat32, ncclSum, g_comm, stream), "ncclAllReduce");
}
// Non-blocking health check of the communicator (FOCR_OK / FOCR_ENCCL + message).
extern "C" int focr_comm_async_error(void) { return async_error(); }
// Host-side watchdog: wait until everything queued on `stream` (the collectives' stream) has completed, polling the
// communicator for asynchronous errors; after `timeout_ms` without completion (or on an error) the communicator is
// ABORTED (ncclCommAbort: the stuck kernels are torn down, the process can exit or rebuild the group) and FOCR_ENCCL is
// returned.  This call synchronises the host with `stream`: the training step calls it only when FOCR_COMM_TIMEOUT_MS
// is set; tools/dp_selfcheck.py always does.
extern "C" int focr_comm_wait(hipStream_t stream, int timeout_ms) {
  if (!g_comm) return FOCR_OK;
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    hipError_t q = hipStreamQuery(stream);
    if (q == hipSuccess) return async_error();
    if (q != hipErrorNotReady) {
      focr_set_error("focr_comm_wait: stream error: %s", hipGetErrorString(q));
      return FOCR_EHIP;
    }
    int rc = async_error();
    const long ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
    if (rc == FOCR_OK && (timeout_ms <= 0 || ms < timeout_ms)) {
      std::this_thread::sleep_for(std::chrono::microseconds(200));
      continue;
    }
    if (rc == FOCR_OK) focr_set_error("focr_comm_wait: collective did not complete within %d ms (peer lost?)", timeout_ms);
    if (g_rccl.CommAbort) (void)g_rccl.CommAbort(g_comm);
    g_comm = nullptr;
    g_nranks = 0;
    return FOCR_ENCCL;
  }
}
extern "C" int focr_comm_nranks(void) { return g_comm ? g_nranks : 0; }
// What RCCL itself says: the rank count of the live communicator (ncclCommCount; 0 without a communicator, negative
// error code on failure) and the library's version code (ncclGetVersion, e.g. 22105; loads RCCL if necessary).
extern "C" int focr_comm_count(void) {
  if (!g_comm) return 0;
  if (!g_rccl.CommCount) return g_nranks;
  int n = 0;
  const int rc = check(g_rccl.CommCount(g_comm, &n), "ncclCommCount");
  return rc == FOCR_OK ? n : rc;
}
extern "C" int focr_comm_rccl_version(void) {
  int rc = load_rccl();
  if (rc != FOCR_OK) return rc;
  if (!g_rccl.GetVersion) return 0;
  int v = 0;
  rc = check(g_rccl.GetVersion(&v), "ncclGetVersion");
  return rc == FOCR_OK ? v : rc;
}
extern "C" int focr_comm_destroy(void) {
  if (!g_comm) return FOCR_OK;
  int rc = check(g_rccl.CommDestroy(g_comm), "ncclCommDestroy");
  g_comm = nullptr;
  g_nranks = 0;
  return rc;
}
