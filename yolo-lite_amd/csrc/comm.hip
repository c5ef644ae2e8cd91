// yl_comm_*: the two exchanges of the sharded detect path over RCCL (xGMI), behind the C ABI (include/yl_amd.h).
//
// the path partitions by image -- weights replicated, the root hands every rank its slice of the
// uint8 batch, every rank sends back one packed (count, max_det x 6) record.  The reference has nothing here
// (utils/torch_utils.py:142-159 always ends on cuda:0), so the contract is: per-shard results identical to the one-GPU run.
// There is no reduction anywhere: scatter and gather are grouped ncclSend / ncclRecv pairs (the root's seven sends leave on
// seven xGMI links at once), and the barrier a caller needs around a timed region is a one-word gather followed by a
// one-word scatter.
//
// RCCL is bound at run time (dlopen("librccl.so.1") on the first yl_comm_* call): a single-GPU user of libyl_amd.so never
// loads it, and a process that already carries a copy (PyTorch-ROCm ships one under the same soname) gets that copy
// instead of a second one with its own device state.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "yl_internal.h"

namespace yl {
namespace {

struct Rccl {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string why;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

template <typename F>
bool bind(void* so, const char* name, F& fn, std::string& why) {
  fn = reinterpret_cast<F>(dlsym(so, name));
  if (!fn) why = std::string("librccl has no symbol ") + name;
  return fn != nullptr;
}

const Rccl* rccl() {
  std::call_once(g_rccl_once, [] {
    Rccl& r = g_rccl;
    const char* names[] = {getenv("YL_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      if (!n || !*n) continue;
      r.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (r.so) break;
      r.why = dlerror();
    }
    if (!r.so) return;
    bool ok = bind(r.so, "ncclGetUniqueId", r.GetUniqueId, r.why) && bind(r.so, "ncclCommInitRank", r.CommInitRank, r.why) &&
              bind(r.so, "ncclCommDestroy", r.CommDestroy, r.why) && bind(r.so, "ncclCommAbort", r.CommAbort, r.why) &&
              bind(r.so, "ncclGroupStart", r.GroupStart, r.why) && bind(r.so, "ncclGroupEnd", r.GroupEnd, r.why) &&
              bind(r.so, "ncclSend", r.Send, r.why) && bind(r.so, "ncclRecv", r.Recv, r.why) &&
              bind(r.so, "ncclGetErrorString", r.GetErrorString, r.why);
    if (!ok) {
      dlclose(r.so);
      r.so = nullptr;
    }
  });
  return g_rccl.so ? &g_rccl : nullptr;
}

int no_rccl() { return fail(-6, "yl_comm: RCCL is not available (" + g_rccl.why + "); the sharded path has no other transport"); }

#define YL_NCCL(r, expr)                                                                     \
  do {                                                                                       \
    ncclResult_t _e = (expr);                                                                \
    if (_e != ncclSuccess) return ::yl::fail(-5, std::string(#expr) + ": " + (r)->GetErrorString(_e)); \
  } while (0)

}  // namespace
}  // namespace yl

struct yl_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  int32_t* word = nullptr;  // world + 1 device words: the barrier's gather target and scatter source
  // Order chain.  Two operations on one communicator must EXECUTE in the same order on every rank (sends to one peer are
  // matched in execution order), and RCCL orders nothing between streams: so every operation that crosses a link waits for
  // the event its predecessor on this communicator recorded, unless that one went to the same stream.  A binder may then
  // name any stream per call (the slot's own: dist.NativeShardedRunner) as long as every rank issues the same sequence.
  // (A wait takes the event's latest record at the time of the call, so a ring of a few events is plenty; one host thread
  // per communicator.)
  hipEvent_t tail[4] = {nullptr, nullptr, nullptr, nullptr};
  uint64_t n_ops = 0;
  hipStream_t last_stream = nullptr;
};

namespace yl {
namespace {
int chain_enter(yl_comm* c, hipStream_t s) {
  if (c->world > 1 && c->n_ops && c->last_stream != s) YL_HIP(hipStreamWaitEvent(s, c->tail[(c->n_ops - 1) & 3], 0));
  return 0;
}
int chain_exit(yl_comm* c, hipStream_t s) {
  if (c->world > 1) {
    YL_HIP(hipEventRecord(c->tail[c->n_ops & 3], s));
    ++c->n_ops;
    c->last_stream = s;
  }
  return 0;
}
}  // namespace
}  // namespace yl

static_assert(sizeof(ncclUniqueId) == YL_COMM_ID_BYTES, "yl_amd.h promises an opaque id of this size");

extern "C" {

#define YL_API __attribute__((visibility("default")))

YL_API int yl_comm_get_unique_id(void* id_out) {
  const yl::Rccl* r = yl::rccl();
  if (!r) return yl::no_rccl();
  if (!id_out) return yl::fail(-1, "yl_comm_get_unique_id: null id");
  ncclUniqueId id;
  YL_NCCL(r, r->GetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return 0;
}

YL_API int yl_comm_create(const void* id_in, int rank, int world, yl_comm** out) {
  const yl::Rccl* r = yl::rccl();
  if (!r) return yl::no_rccl();
  if (!id_in || !out) return yl::fail(-1, "yl_comm_create: null argument");
  if (world < 1 || rank < 0 || rank >= world) return yl::fail(-1, "yl_comm_create: rank " + std::to_string(rank) + " of " + std::to_string(world));
  ncclUniqueId id;
  memcpy(&id, id_in, sizeof(id));
  yl_comm* c = new yl_comm;
  c->rank = rank;
  c->world = world;
  if (hipGetDevice(&c->device) != hipSuccess) {
    delete c;
    return yl::fail(-2, "yl_comm_create: no current HIP device");
  }
  ncclResult_t e = r->CommInitRank(&c->comm, world, id, rank);
  if (e != ncclSuccess) {
    delete c;
    return yl::fail(-5, std::string("ncclCommInitRank: ") + r->GetErrorString(e));
  }
  if (hipMalloc(reinterpret_cast<void**>(&c->word), sizeof(int32_t) * (size_t)(world + 1)) != hipSuccess ||
      hipMemset(c->word, 0, sizeof(int32_t) * (size_t)(world + 1)) != hipSuccess) {
    r->CommAbort(c->comm);
    delete c;
    return yl::fail(-2, "yl_comm_create: hipMalloc of the barrier words failed");
  }
  for (hipEvent_t& ev : c->tail)
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
      yl_comm_destroy(c);
      return yl::fail(-2, "yl_comm_create: hipEventCreate failed");
    }
  *out = c;
  return 0;
}

YL_API void yl_comm_destroy(yl_comm* c) {
  if (!c) return;
  const yl::Rccl* r = yl::rccl();
  if (c->word) (void)hipFree(c->word);
  for (hipEvent_t ev : c->tail)
    if (ev) (void)hipEventDestroy(ev);
  if (r && c->comm) r->CommDestroy(c->comm);
  delete c;
}

YL_API int yl_comm_rank(const yl_comm* c) { return c ? c->rank : -1; }
YL_API int yl_comm_world(const yl_comm* c) { return c ? c->world : -1; }

// root: `send` = world slices of `bytes` each, rank order; every rank (the root too) receives its slice into `recv`.
YL_API int yl_comm_scatter(yl_comm* c, const void* send, void* recv, int64_t bytes, int root, void* stream) {
  const yl::Rccl* r = yl::rccl();
  if (!r) return yl::no_rccl();
  if (!c || !recv || bytes < 0 || root < 0 || root >= c->world) return yl::fail(-1, "yl_comm_scatter: bad argument");
  if (c->rank == root && !send) return yl::fail(-1, "yl_comm_scatter: the root must pass the full buffer");
  if (bytes == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (c->world > 1) {  // (one rank: nothing crosses a link, and an empty group is not free)
    if (int rc = yl::chain_enter(c, s)) return rc;
    YL_NCCL(r, r->GroupStart());
    ncclResult_t e = ncclSuccess;
    if (c->rank == root) {
      const char* p = static_cast<const char*>(send);
      for (int k = 0; k < c->world && e == ncclSuccess; ++k)
        if (k != root) e = r->Send(p + (size_t)k * (size_t)bytes, (size_t)bytes, ncclUint8, k, c->comm, s);
    } else {
      e = r->Recv(recv, (size_t)bytes, ncclUint8, root, c->comm, s);
    }
    ncclResult_t e2 = r->GroupEnd();
    YL_NCCL(r, e);
    YL_NCCL(r, e2);
    if (int rc = yl::chain_exit(c, s)) return rc;
  }
  if (c->rank == root) {  // the root's own slice never touches a link
    const char* mine = static_cast<const char*>(send) + (size_t)root * (size_t)bytes;
    if (mine != recv) YL_HIP(hipMemcpyAsync(recv, mine, (size_t)bytes, hipMemcpyDeviceToDevice, s));
  }
  return 0;
}

// every rank sends `bytes` from `send`; the root receives world slices into `recv`, rank order (its own by a device copy).
YL_API int yl_comm_gather(yl_comm* c, const void* send, void* recv, int64_t bytes, int root, void* stream) {
  const yl::Rccl* r = yl::rccl();
  if (!r) return yl::no_rccl();
  if (!c || !send || bytes < 0 || root < 0 || root >= c->world) return yl::fail(-1, "yl_comm_gather: bad argument");
  if (c->rank == root && !recv) return yl::fail(-1, "yl_comm_gather: the root must pass the full buffer");
  if (bytes == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (c->world > 1) {
    if (int rc = yl::chain_enter(c, s)) return rc;
    YL_NCCL(r, r->GroupStart());
    ncclResult_t e = ncclSuccess;
    if (c->rank == root) {
      char* p = static_cast<char*>(recv);
      for (int k = 0; k < c->world && e == ncclSuccess; ++k)
        if (k != root) e = r->Recv(p + (size_t)k * (size_t)bytes, (size_t)bytes, ncclUint8, k, c->comm, s);
    } else {
      e = r->Send(send, (size_t)bytes, ncclUint8, root, c->comm, s);
    }
    ncclResult_t e2 = r->GroupEnd();
    YL_NCCL(r, e);
    YL_NCCL(r, e2);
    if (int rc = yl::chain_exit(c, s)) return rc;
  }
  if (c->rank == root) {
    char* mine = static_cast<char*>(recv) + (size_t)root * (size_t)bytes;
    if (mine != send) YL_HIP(hipMemcpyAsync(mine, send, (size_t)bytes, hipMemcpyDeviceToDevice, s));
  }
  return 0;
}

// Every rank's work enqueued on `stream` before the call has finished on every rank when this returns (rank 0 collects one
// word from everybody, then hands one word back; the host waits for its own stream).  No reduction kernel is involved.
YL_API int yl_comm_barrier(yl_comm* c, void* stream) {
  if (!c) return yl::fail(-1, "yl_comm_barrier: null communicator");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (c->world > 1) {
    if (int rc = yl_comm_gather(c, c->word + c->world, c->word, sizeof(int32_t), 0, stream)) return rc;
    if (int rc = yl_comm_scatter(c, c->word, c->word + c->world, sizeof(int32_t), 0, stream)) return rc;
  }
  YL_HIP(hipStreamSynchronize(s));
  return 0;
}

}  // extern "C"
