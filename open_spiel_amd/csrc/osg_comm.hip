// RCCL collective behind the C-ABI: the one exchange step of the path for callers
// that have no torch.distributed — C++ hosts of the reference's shape.  One communicator per
// (process, GPU); all-reduce(sum) over xGMI of the ES-MCCFR delta tables (2 x [I, Amax] fp64:
// 44 928 B for leduc_poker, latency-bound) or of a shared root's visit / reward vectors.
//
// RCCL is resolved at run time (dlopen "librccl.so.1"): the library has no link-time dependency on
// it, single-GPU users never load it, and inside a PyTorch process the loader hands back the RCCL
// PyTorch already mapped (same soname), so the two never coexist as different copies.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>

#include "osg_internal.h"

using osg::set_error;

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};

RcclApi& Rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (api.handle) break;
    }
    if (!api.handle) {
      const char* e = dlerror();
      api.error = std::string("cannot load RCCL (librccl.so.1): ") + (e ? e : "unknown error");
      return;
    }
    auto sym = [&](const char* name) -> void* {
      void* p = dlsym(api.handle, name);
      if (!p && api.error.empty()) api.error = std::string("RCCL lacks ") + name;
      return p;
    };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
  });
  return api;
}

int RcclReady(RcclApi** out) {
  RcclApi& api = Rccl();
  if (!api.error.empty()) return set_error(OSG_ERR_UNSUPPORTED, api.error);
  *out = &api;
  return OSG_OK;
}

int RcclFail(RcclApi* api, const char* what, ncclResult_t r) {
  return set_error(OSG_ERR_HIP, std::string(what) + ": " + (api->GetErrorString ? api->GetErrorString(r) : "RCCL error"));
}

}  // namespace

struct osg_comm {
  osg_ctx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0;
  int world = 1;
  // the asynchronous form (osg_allreduce_sum_f64_begin / osg_allreduce_end): the collective runs on the
  // communicator's own stream between two events, so kernels issued on the context's stream meanwhile overlap it
  hipStream_t side = nullptr;
  hipEvent_t produced = nullptr, reduced = nullptr;
  bool in_flight = false;
};

static_assert(OSG_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "osg_abi.h and rccl.h disagree on the id size");

extern "C" {

int osg_comm_unique_id(void* id_out) {
  if (!id_out) return set_error(OSG_ERR_INVALID, "osg_comm_unique_id: null argument");
  RcclApi* api = nullptr;
  if (int rc = RcclReady(&api)) return rc;
  ncclUniqueId id;
  ncclResult_t r = api->GetUniqueId(&id);
  if (r != ncclSuccess) return RcclFail(api, "ncclGetUniqueId", r);
  std::memcpy(id_out, id.internal, OSG_COMM_ID_BYTES);
  return OSG_OK;
}

int osg_comm_create(osg_ctx* ctx, int rank, int world, const void* id, osg_comm** out) {
  if (!ctx || !id || !out) return set_error(OSG_ERR_INVALID, "osg_comm_create: null argument");
  if (world < 1 || rank < 0 || rank >= world) return set_error(OSG_ERR_INVALID, "osg_comm_create: bad rank / world");
  RcclApi* api = nullptr;
  if (int rc = RcclReady(&api)) return rc;
  OSG_HIP(hipSetDevice(ctx->device));
  ncclUniqueId uid;
  std::memcpy(uid.internal, id, OSG_COMM_ID_BYTES);
  ncclComm_t comm = nullptr;
  ncclResult_t r = api->CommInitRank(&comm, world, uid, rank);
  if (r != ncclSuccess) return RcclFail(api, "ncclCommInitRank", r);
  osg_comm* c = new osg_comm;
  osg::ctx_retain(ctx);
  c->ctx = ctx;
  c->comm = comm;
  c->rank = rank;
  c->world = world;
  *out = c;
  return OSG_OK;
}

int osg_comm_destroy(osg_comm* c) {
  if (!c) return OSG_OK;
  RcclApi* api = nullptr;
  int rc = RcclReady(&api);
  if (rc == OSG_OK && c->comm) {
    hipStreamSynchronize(c->ctx->stream);
    if (c->side) hipStreamSynchronize(c->side);
    api->CommDestroy(c->comm);
  }
  if (c->produced) hipEventDestroy(c->produced);
  if (c->reduced) hipEventDestroy(c->reduced);
  if (c->side) hipStreamDestroy(c->side);
  osg::ctx_release(c->ctx);
  delete c;
  return rc;
}

int osg_comm_rank(const osg_comm* c) { return c ? c->rank : -1; }
int osg_comm_world(const osg_comm* c) { return c ? c->world : -1; }

static int AllReduceSum(osg_comm* c, void* d_buf, int64_t n, ncclDataType_t type, const char* what) {
  if (!c || (!d_buf && n > 0) || n < 0) return set_error(OSG_ERR_INVALID, std::string(what) + ": bad argument");
  if (n == 0) return OSG_OK;
  RcclApi* api = nullptr;
  if (int rc = RcclReady(&api)) return rc;
  // In place, on the context's stream: ordered after the kernels that produced the buffer and
  // before the ones that consume it, no host synchronisation.
  ncclResult_t r = api->AllReduce(d_buf, d_buf, static_cast<size_t>(n), type, ncclSum, c->comm, c->ctx->stream);
  if (r != ncclSuccess) return RcclFail(api, what, r);
  return OSG_OK;
}

int osg_allreduce_sum_f64(osg_comm* c, double* d_buf, int64_t n) {
  return AllReduceSum(c, d_buf, n, ncclFloat64, "osg_allreduce_sum_f64");
}

int osg_allreduce_sum_i32(osg_comm* c, int32_t* d_buf, int64_t n) {
  return AllReduceSum(c, d_buf, n, ncclInt32, "osg_allreduce_sum_i32");
}

int osg_allreduce_sum_f64_begin(osg_comm* c, double* d_buf, int64_t n) {
  if (!c || (!d_buf && n > 0) || n < 0) return set_error(OSG_ERR_INVALID, "osg_allreduce_sum_f64_begin: bad argument");
  if (c->in_flight) return set_error(OSG_ERR_INVALID, "osg_allreduce_sum_f64_begin: one collective in flight per communicator (call osg_allreduce_end first)");
  RcclApi* api = nullptr;
  if (int rc = RcclReady(&api)) return rc;
  OSG_HIP(hipSetDevice(c->ctx->device));
  if (!c->side) {
    OSG_HIP(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    OSG_HIP(hipEventCreateWithFlags(&c->produced, hipEventDisableTiming));
    OSG_HIP(hipEventCreateWithFlags(&c->reduced, hipEventDisableTiming));
  }
  // everything issued on the context's stream so far (the kernels that filled the buffer) comes first
  OSG_HIP(hipEventRecord(c->produced, c->ctx->stream));
  OSG_HIP(hipStreamWaitEvent(c->side, c->produced, 0));
  if (n > 0) {
    ncclResult_t r = api->AllReduce(d_buf, d_buf, static_cast<size_t>(n), ncclFloat64, ncclSum, c->comm, c->side);
    if (r != ncclSuccess) return RcclFail(api, "osg_allreduce_sum_f64_begin", r);
  }
  OSG_HIP(hipEventRecord(c->reduced, c->side));
  c->in_flight = true;
  return OSG_OK;
}

int osg_allreduce_end(osg_comm* c) {
  if (!c) return set_error(OSG_ERR_INVALID, "osg_allreduce_end: null argument");
  if (!c->in_flight) return OSG_OK;
  // no host wait: what is issued on the context's stream from here on runs after the collective
  OSG_HIP(hipStreamWaitEvent(c->ctx->stream, c->reduced, 0));
  c->in_flight = false;
  return OSG_OK;
}

}  // extern "C"
