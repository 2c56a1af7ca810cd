// RCCL collective behind the C-ABI: the one exchange step of the path for callers
// that have no torch.distributed — C++ hosts of the reference's shape.  One communicator per
// (process, GPU); all-reduce(sum) over xGMI of the ES-MCCFR delta tables (2 x [I, Amax] fp64:
// 44 928 B for leduc_poker, latency-bound) or of a shared root's visit / reward vectors.
//
// Beside it, for exactly these latency-bound messages: the one-shot all-reduce over hipIpc-mapped peer windows
// (osg_comm_oneshot_*; see the comment above k_oneshot_allreduce).  Both kinds answer osg_allreduce_sum_*.
//
// RCCL is resolved at run time (dlopen "librccl.so.1"): the library has no link-time dependency on
// it, single-GPU users never load it, and inside a PyTorch process the loader hands back the RCCL
// PyTorch already mapped (same soname), so the two never coexist as different copies.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>
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

// ---------------------------------------------------------------------------
// The one-shot all-reduce ("prefer a one-shot direct all-reduce — each rank writes its buffer
// to all 7 peers, reduces locally" for the <= 45 KB messages of this path, which a ring's 2 (N - 1) latency-bound
// steps serve badly).  Every rank owns a WINDOW in its own HBM — fine-grained device memory, exported with
// hipIpcGetMemHandle and mapped by every peer —
//     data  [2 parities][world slots][cap doubles]     slot s = what rank s contributes
//     flags [2 parities][world][kOneShotMaxBlocks] u64 sequence number of the call whose chunk has landed
// and one launch per call does everything, chunk by chunk (workgroup g owns elements [g * kChunk, (g + 1) * kChunk)):
//     push   chunk g of the local buffer into slot `rank` of EVERY window (system-scope write-through stores over
//            xGMI; the own window too), drain, release, then flag[parity][rank][g] = seq in every window;
//     wait   until flag[parity][s][g] == seq for every s (one lane polls, bounded by a wall-clock timeout);
//     reduce chunk g of the slots IN RANK ORDER — the same additions in the same order on every rank, so all
//            ranks end with bit-identical sums (RCCL's ring gives each rank a different association) — into
//            the caller's buffer.
// Workgroup g of one rank only ever waits for workgroup g of the others: no grid-wide barrier, no co-residency
// assumption beyond "a launched workgroup eventually runs".  Two parities: a rank can be one call ahead of a peer,
// never two (it cannot finish call k + 1 before every peer has pushed k + 1, i.e. finished reading call k).
// ---------------------------------------------------------------------------
constexpr int kOneShotBlock = 256;
constexpr int kOneShotChunk = 2 * kOneShotBlock;   // elements per workgroup: two per lane
constexpr int kOneShotMaxBlocks = 64;              // => messages of up to 32 768 elements (256 KiB of fp64)
constexpr int kOneShotMaxWorld = 16;

struct OneShotPeers {
  double* data[kOneShotMaxWorld];                  // window of rank s as mapped in THIS process
  unsigned long long* flags[kOneShotMaxWorld];
};

template <class T> struct Sys;
template <> struct Sys<double> {
  static __device__ __forceinline__ void store(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), static_cast<unsigned long long>(__double_as_longlong(v)),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  static __device__ __forceinline__ double load(const double* p) {
    return __longlong_as_double(static_cast<long long>(__hip_atomic_load(
        reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)));
  }
  static __device__ __forceinline__ double poison() { return __longlong_as_double(0x7FF8000000000000ll); }
};
template <> struct Sys<int32_t> {
  static __device__ __forceinline__ void store(int32_t* p, int32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  static __device__ __forceinline__ int32_t load(const int32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  static __device__ __forceinline__ int32_t poison() { return INT32_MIN; }
};

// T = double or int32_t (a slot is cap doubles = 2 cap int32).  err[0] is raised on a timeout.
template <class T>
__global__ void __launch_bounds__(kOneShotBlock)
k_oneshot_allreduce(OneShotPeers peers, T* buf, int64_t n, int rank, int world, int64_t slot_elems, unsigned long long seq,
                    unsigned long long timeout_ticks, unsigned int* err) {
  __shared__ int s_ok;
  const int g = blockIdx.x, tid = threadIdx.x;
  const int parity = static_cast<int>(seq & 1ull);
  const int64_t e0 = static_cast<int64_t>(g) * kOneShotChunk + tid, e1 = e0 + kOneShotBlock;
  // ---- push: my chunk into slot `rank` of every window ----
  T v0 = e0 < n ? buf[e0] : T(0), v1 = e1 < n ? buf[e1] : T(0);
  for (int d = 0; d < world; ++d) {
    const int peer = (rank + d) % world;  // own window first, then round the ring: ranks do not all hit one peer at once
    T* slot = reinterpret_cast<T*>(peers.data[peer]) + (static_cast<int64_t>(parity) * world + rank) * slot_elems;
    if (e0 < n) Sys<T>::store(slot + e0, v0);
    if (e1 < n) Sys<T>::store(slot + e1, v1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");    // system scope: nothing of this workgroup is still on its way
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int d = 0; d < world; ++d) {
      const int peer = (rank + d) % world;
      __hip_atomic_store(peers.flags[peer] + (static_cast<int64_t>(parity) * world + rank) * kOneShotMaxBlocks + g, seq,
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // ---- wait: chunk g of every rank has landed in MY window ----
    // The bound is on SILENCE, not on skew: the clock restarts whenever another rank's chunk arrives, so a rank that is
    // legitimately late (a NashConv evaluation, a checkpoint between two steps) is waited for as RCCL would, and only a
    // peer that stays silent for the whole bound (a dead process, a lost mapping) ends the call.
    const unsigned long long* mine = peers.flags[rank] + static_cast<int64_t>(parity) * world * kOneShotMaxBlocks + g;
    unsigned long long t0 = wall_clock64();
    int ok = 1;
    for (int s = 0; s < world && ok; ++s) {
      while (__hip_atomic_load(mine + static_cast<int64_t>(s) * kOneShotMaxBlocks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
        if (timeout_ticks != 0ull && wall_clock64() - t0 > timeout_ticks) { ok = 0; break; }
        __builtin_amdgcn_s_sleep(2);
      }
      t0 = wall_clock64();
    }
    // acquire at system scope: the slots read below were written by other agents before the flags seen above (the data
    // loads are system-scope atomics on a fine-grained window already; the fence — one cache invalidate — makes the
    // order hold for any mapping of the window)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    if (!ok) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // pinned host memory
    s_ok = ok;
  }
  __syncthreads();
  if (!s_ok) {
    // A peer never arrived.  Workgroups time out one by one, so without this the caller's buffer would end as a mix of
    // reduced chunks and local values that looks like a result: the chunk is POISONED instead (NaN / INT32_MIN), the
    // pinned error word is raised, and osg_comm_check — or the next call on the communicator — reports it.  A buffer
    // that went through a failed collective can therefore not be folded into tables unnoticed: it holds NaNs.
    if (e0 < n) buf[e0] = Sys<T>::poison();
    if (e1 < n) buf[e1] = Sys<T>::poison();
    return;
  }
  // ---- reduce: the slots in rank order (system-scope loads: never a stale cached line of an earlier call) ----
  const T* slots = reinterpret_cast<const T*>(peers.data[rank]) + static_cast<int64_t>(parity) * world * slot_elems;
  T a0 = T(0), a1 = T(0);
  for (int s = 0; s < world; ++s) {
    if (e0 < n) a0 += Sys<T>::load(slots + s * slot_elems + e0);
    if (e1 < n) a1 += Sys<T>::load(slots + s * slot_elems + e1);
  }
  if (e0 < n) buf[e0] = a0;
  if (e1 < n) buf[e1] = a1;
}

struct osg_comm {
  osg_ctx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0;
  int world = 1;
  // one-shot kind (osg_comm_oneshot_create): the local window, the peers' mappings, the call counter
  bool oneshot = false, connected = false;
  int64_t cap = 0;                        // doubles per slot
  void* window = nullptr;                 // data | flags, one fine-grained allocation
  size_t window_bytes = 0, flags_offset = 0;
  void* mapped[kOneShotMaxWorld] = {};    // peers' windows as opened here (own entry = window)
  OneShotPeers peers{};
  unsigned long long seq = 0;
  unsigned int* h_err = nullptr;          // pinned host word the kernel raises on a timeout
  unsigned long long timeout_ticks = 0;
  // the asynchronous form (osg_allreduce_sum_f64_begin / osg_allreduce_end): the collective runs on the
  // communicator's own stream between two events, so kernels issued on the context's stream meanwhile overlap it
  hipStream_t side = nullptr;
  hipEvent_t produced = nullptr, reduced = nullptr;
  bool in_flight = false;
};

static_assert(OSG_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "osg_abi.h and rccl.h disagree on the id size");

// One call of the one-shot all-reduce on `stream` (T = double / int32_t).
template <class T>
static int OneShotLaunch(osg_comm* c, T* d_buf, int64_t n, hipStream_t stream, const char* what) {
  if (!c->connected) return set_error(OSG_ERR_INVALID, std::string(what) + ": osg_comm_oneshot_connect has not been called");
  const int64_t slot_elems = c->cap * static_cast<int64_t>(sizeof(double) / sizeof(T));
  if (n > std::min<int64_t>(slot_elems, static_cast<int64_t>(kOneShotMaxBlocks) * kOneShotChunk))
    return set_error(OSG_ERR_INVALID, std::string(what) + ": message longer than the window's slot (max_doubles at creation; "
                                                          "at most 32768 elements per call)");
  // raised by an EARLIER call's kernel (system-scope store into pinned host memory: read here without a copy)
  if (__atomic_load_n(c->h_err, __ATOMIC_RELAXED) != 0)
    return set_error(OSG_ERR_HIP, std::string(what) + ": an earlier one-shot all-reduce timed out waiting for a peer "
                                                      "(OSG_ONESHOT_TIMEOUT_MS); the communicator is unusable");
  const unsigned blocks = static_cast<unsigned>((n + kOneShotChunk - 1) / kOneShotChunk);
  ++c->seq;
  k_oneshot_allreduce<T><<<dim3(blocks), dim3(kOneShotBlock), 0, stream>>>(c->peers, d_buf, n, c->rank, c->world, slot_elems,
                                                                            c->seq, c->timeout_ticks, c->h_err);
  OSG_HIP(hipGetLastError());
  return OSG_OK;
}

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
  if (c->oneshot) {
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->side) (void)hipStreamSynchronize(c->side);
    for (int s = 0; s < c->world; ++s)
      if (s != c->rank && c->mapped[s]) (void)hipIpcCloseMemHandle(c->mapped[s]);
    if (c->window) (void)hipFree(c->window);
    if (c->h_err) (void)hipHostFree(c->h_err);
    if (c->produced) (void)hipEventDestroy(c->produced);
    if (c->reduced) (void)hipEventDestroy(c->reduced);
    if (c->side) (void)hipStreamDestroy(c->side);
    osg::ctx_release(c->ctx);
    delete c;
    return OSG_OK;
  }
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

int osg_comm_oneshot_create(osg_ctx* ctx, int rank, int world, int64_t max_doubles, osg_comm** out) {
  if (!ctx || !out) return set_error(OSG_ERR_INVALID, "osg_comm_oneshot_create: null argument");
  if (world < 1 || world > kOneShotMaxWorld || rank < 0 || rank >= world)
    return set_error(OSG_ERR_INVALID, "osg_comm_oneshot_create: bad rank / world (at most 16 ranks)");
  if (max_doubles < 1 || max_doubles > static_cast<int64_t>(kOneShotMaxBlocks) * kOneShotChunk)
    return set_error(OSG_ERR_INVALID, "osg_comm_oneshot_create: max_doubles must lie in [1, 32768] (the one-shot form is for "
                                      "latency-bound messages; use the RCCL communicator beyond)");
  OSG_HIP(hipSetDevice(ctx->device));
  osg_comm* c = new osg_comm;
  c->oneshot = true;
  c->rank = rank;
  c->world = world;
  c->cap = (max_doubles + 1) & ~int64_t{1};
  c->flags_offset = (sizeof(double) * 2 * world * c->cap + 255) & ~size_t{255};
  c->window_bytes = c->flags_offset + sizeof(unsigned long long) * 2 * world * kOneShotMaxBlocks;
  // fine-grained: peers' write-through stores and this rank's system-scope loads meet in memory, not in a cache
  hipError_t e = hipExtMallocWithFlags(&c->window, c->window_bytes, hipDeviceMallocFinegrained);
  if (e == hipSuccess) e = hipMemset(c->window, 0, c->window_bytes);
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&c->h_err), sizeof(unsigned int), hipHostMallocMapped);
  if (e == hipSuccess) *c->h_err = 0;
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    if (c->window) (void)hipFree(c->window);
    if (c->h_err) (void)hipHostFree(c->h_err);
    delete c;
    return set_error(OSG_ERR_NOMEM, std::string("osg_comm_oneshot_create: ") + hipGetErrorString(e));
  }
  // How long a peer may stay SILENT before the call gives up (the clock restarts with every chunk that arrives): two
  // minutes by default — a hang detector, not a skew bound; OSG_ONESHOT_TIMEOUT_MS=0 waits for ever, as RCCL does.
  double ms = 120000.0;
  if (const char* t = std::getenv("OSG_ONESHOT_TIMEOUT_MS")) ms = std::max(0.0, std::atof(t));
  c->timeout_ticks = static_cast<unsigned long long>(ms * 1e5);   // wall_clock64 counts at 100 MHz; 0 = no bound
  osg::ctx_retain(ctx);
  c->ctx = ctx;
  *out = c;
  return OSG_OK;
}

int osg_comm_oneshot_handle(const osg_comm* c, void* handle_out) {
  if (!c || !c->oneshot || !handle_out) return set_error(OSG_ERR_INVALID, "osg_comm_oneshot_handle: not a one-shot communicator");
  static_assert(sizeof(hipIpcMemHandle_t) + 2 * sizeof(int64_t) <= OSG_ONESHOT_HANDLE_BYTES, "handle does not fit");
  std::memset(handle_out, 0, OSG_ONESHOT_HANDLE_BYTES);
  hipIpcMemHandle_t h;
  OSG_HIP(hipSetDevice(c->ctx->device));
  OSG_HIP(hipIpcGetMemHandle(&h, c->window));
  char* o = static_cast<char*>(handle_out);
  std::memcpy(o, &h, sizeof(h));
  const int64_t meta[2] = {c->cap, c->world};   // checked by the peers: everyone must have created the same geometry
  std::memcpy(o + sizeof(h), meta, sizeof(meta));
  return OSG_OK;
}

int osg_comm_oneshot_connect(osg_comm* c, const void* handles) {
  if (!c || !c->oneshot || !handles) return set_error(OSG_ERR_INVALID, "osg_comm_oneshot_connect: not a one-shot communicator");
  if (c->connected) return set_error(OSG_ERR_INVALID, "osg_comm_oneshot_connect: already connected");
  OSG_HIP(hipSetDevice(c->ctx->device));
  const char* in = static_cast<const char*>(handles);
  for (int s = 0; s < c->world; ++s) {
    const char* rec = in + static_cast<size_t>(s) * OSG_ONESHOT_HANDLE_BYTES;
    int64_t meta[2];
    std::memcpy(meta, rec + sizeof(hipIpcMemHandle_t), sizeof(meta));
    if (meta[0] != c->cap || meta[1] != c->world)
      return set_error(OSG_ERR_INVALID, "osg_comm_oneshot_connect: rank " + std::to_string(s) + " created its window with another max_doubles / world");
    if (s == c->rank) {
      c->mapped[s] = c->window;
    } else {
      hipIpcMemHandle_t h;
      std::memcpy(&h, rec, sizeof(h));
      hipError_t e = hipIpcOpenMemHandle(&c->mapped[s], h, hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess) {
        c->mapped[s] = nullptr;
        return set_error(OSG_ERR_HIP, "hipIpcOpenMemHandle (window of rank " + std::to_string(s) + "): " + hipGetErrorString(e) +
                                          " — HSA_ENABLE_IPC_MODE_LEGACY=0 is needed on hosts whose driver only offers dmabuf IPC");
      }
    }
    c->peers.data[s] = static_cast<double*>(c->mapped[s]);
    c->peers.flags[s] = reinterpret_cast<unsigned long long*>(static_cast<char*>(c->mapped[s]) + c->flags_offset);
  }
  c->connected = true;
  return OSG_OK;
}

int osg_comm_check(osg_comm* c) {
  if (!c) return set_error(OSG_ERR_INVALID, "osg_comm_check: null argument");
  OSG_HIP(hipSetDevice(c->ctx->device));
  if (c->side) OSG_HIP(hipStreamSynchronize(c->side));
  OSG_HIP(hipStreamSynchronize(c->ctx->stream));
  if (c->oneshot && __atomic_load_n(c->h_err, __ATOMIC_RELAXED) != 0)
    return set_error(OSG_ERR_HIP, "osg_comm_check: a one-shot all-reduce timed out waiting for a peer (OSG_ONESHOT_TIMEOUT_MS); the "
                                  "buffer of that call holds NaN / INT32_MIN in the chunks that were not reduced and the "
                                  "communicator is unusable");
  return OSG_OK;
}

int osg_comm_rank(const osg_comm* c) { return c ? c->rank : -1; }
int osg_comm_world(const osg_comm* c) { return c ? c->world : -1; }

static int AllReduceSum(osg_comm* c, void* d_buf, int64_t n, ncclDataType_t type, const char* what) {
  if (!c || (!d_buf && n > 0) || n < 0) return set_error(OSG_ERR_INVALID, std::string(what) + ": bad argument");
  if (n == 0) return OSG_OK;
  if (c->in_flight) return set_error(OSG_ERR_INVALID, std::string(what) + ": a collective begun with osg_allreduce_sum_f64_begin is still in flight (call osg_allreduce_end first)");
  if (c->oneshot) {
    return type == ncclFloat64 ? OneShotLaunch(c, static_cast<double*>(d_buf), n, c->ctx->stream, what)
                               : OneShotLaunch(c, static_cast<int32_t*>(d_buf), n, c->ctx->stream, what);
  }
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
  if (!c->oneshot)
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
  if (n > 0 && c->oneshot) {
    if (int rc = OneShotLaunch(c, d_buf, n, c->side, "osg_allreduce_sum_f64_begin")) return rc;
  } else if (n > 0) {
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
