// libmscnn_dist.so: detections all-gather over RCCL / xGMI (include/mscnn_dist.h).  Host-only C++; RCCL is resolved with
// dlopen so that nothing here interposes on (or depends on) an RCCL another component of the process has mapped.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>

#include "../../include/mscnn_dist.h"

namespace {

thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

char g_transport[1024] = "";      // mscnn_dist_use_transport: an explicit library path instead of the default search

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    if (g_transport[0]) r.handle = dlopen(g_transport, RTLD_NOW | RTLD_LOCAL);
    else
      for (const char* n : names) {
        r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (r.handle) break;
      }
    if (!r.handle) return;
#define LOAD(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, sym))
    LOAD(GetUniqueId, "ncclGetUniqueId");
    LOAD(CommInitRank, "ncclCommInitRank");
    LOAD(CommDestroy, "ncclCommDestroy");
    LOAD(CommCount, "ncclCommCount");
    LOAD(CommUserRank, "ncclCommUserRank");
    LOAD(AllGather, "ncclAllGather");
    LOAD(AllReduce, "ncclAllReduce");
    LOAD(GetErrorString, "ncclGetErrorString");
#undef LOAD
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.CommCount || !r.CommUserRank || !r.AllGather || !r.AllReduce || !r.GetErrorString) {
      dlclose(r.handle);
      r.handle = nullptr;
    }
  });
  return r.handle ? &r : nullptr;
}

#define DIST_REQUIRE(cond, ...) do { if (!(cond)) { set_error(__VA_ARGS__); return 1; } } while (0)
#define DIST_HIP(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); return 2; } } while (0)
#define DIST_NCCL(expr) do { ncclResult_t e__ = (expr); if (e__ != ncclSuccess) { set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, R->GetErrorString(e__)); return 3; } } while (0)

}  // namespace

struct mscnn_dist {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;      // as the COMMUNICATOR reports them (ncclCommUserRank / ncclCommCount), checked against the arguments
  size_t pack_bytes = 0;
  void* send_dev = nullptr;       // blocking form: this rank's pack, stamped with its rank
  void* recv_dev = nullptr;       // world * pack_bytes
  void* recv_host = nullptr;      // pinned, same size
  int* flag_dev = nullptr;        // barrier payload
  // pipelined exchange (mscnn_dist_all_gather_begin / _end): two slots on the communicator's own stream
  hipStream_t xstream = nullptr;
  hipEvent_t ready = nullptr;     // compute stream -> exchange stream: the pack copy of this step has been enqueued
  struct Slot { void* send_dev = nullptr; void* recv_dev = nullptr; void* recv_host = nullptr; hipEvent_t done = nullptr; bool busy = false; } slot[2];
  int head = 0, tail = 0, inflight = 0;
};

extern "C" void mscnn_dist_set_error_text(const char* text) { set_error("%s", text); }      // (library-internal: placement.cpp)

extern "C" {

const char* mscnn_dist_last_error(void) { return g_err; }

int mscnn_dist_use_transport(const char* library_path) {
  DIST_REQUIRE(library_path && library_path[0] && std::strlen(library_path) < sizeof(g_transport), "use_transport: bad path");
  std::snprintf(g_transport, sizeof(g_transport), "%s", library_path);
  return 0;
}

int mscnn_dist_unique_id(unsigned char id_out[MSCNN_DIST_ID_BYTES]) {
  static_assert(sizeof(ncclUniqueId) == MSCNN_DIST_ID_BYTES, "ncclUniqueId size");
  Rccl* R = rccl();
  DIST_REQUIRE(R, "librccl.so.1 not found (dlopen): multi-GPU gather unavailable");
  DIST_REQUIRE(id_out, "null id");
  ncclUniqueId id;
  DIST_NCCL(R->GetUniqueId(&id));
  std::memcpy(id_out, &id, sizeof(id));
  return 0;
}

int mscnn_dist_init(const unsigned char idb[MSCNN_DIST_ID_BYTES], int rank, int world, int device, size_t pack_bytes,
                    mscnn_dist** out) {
  Rccl* R = rccl();
  DIST_REQUIRE(R, "librccl.so.1 not found (dlopen): multi-GPU gather unavailable");
  DIST_REQUIRE(idb && out && world >= 1 && rank >= 0 && rank < world && pack_bytes > 0 && pack_bytes % 16 == 0,
               "dist init: bad argument (rank %d of %d, pack %zu bytes)", rank, world, pack_bytes);
  DIST_HIP(hipSetDevice(device));
  mscnn_dist* d = new (std::nothrow) mscnn_dist();
  DIST_REQUIRE(d, "out of memory");
  d->rank = rank; d->world = world; d->device = device; d->pack_bytes = pack_bytes;
  ncclUniqueId id;
  std::memcpy(&id, idb, sizeof(id));
  ncclResult_t rc = R->CommInitRank(&d->comm, world, id, rank);
  if (rc != ncclSuccess) {
    set_error("ncclCommInitRank(rank %d of %d, device %d) -> %s", rank, world, device, R->GetErrorString(rc));
    delete d;
    return 3;
  }
  // what the collective library itself says about this communicator: the line a scaling run prints must be able to prove that RCCL
  // saw N ranks, not that the launcher was asked for N
  int cw = -1, cr = -1;
  ncclResult_t q = R->CommCount(d->comm, &cw);
  if (q == ncclSuccess) q = R->CommUserRank(d->comm, &cr);
  if (q != ncclSuccess || cw != world || cr != rank) {
    if (q != ncclSuccess) set_error("ncclCommCount / ncclCommUserRank -> %s", R->GetErrorString(q));
    else set_error("the communicator reports rank %d of %d, the caller asked for rank %d of %d", cr, cw, rank, world);
    mscnn_dist_destroy(d);
    return 3;
  }
  d->world = cw; d->rank = cr;
  hipError_t e = hipMalloc(&d->recv_dev, pack_bytes * world);
  if (e == hipSuccess) e = hipMalloc(&d->send_dev, pack_bytes);
  if (e == hipSuccess) e = hipHostMalloc(&d->recv_host, pack_bytes * world, hipHostMallocDefault);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d->flag_dev), 2 * sizeof(int));
  if (e == hipSuccess) e = hipMemset(d->flag_dev, 0, 2 * sizeof(int));
  if (e != hipSuccess) {
    set_error("dist init: buffers: %s", hipGetErrorString(e));
    mscnn_dist_destroy(d);
    return 2;
  }
  *out = d;
  return 0;
}

void mscnn_dist_destroy(mscnn_dist* d) {
  if (!d) return;
  Rccl* R = rccl();
  (void)hipSetDevice(d->device);
  if (d->xstream) (void)hipStreamSynchronize(d->xstream);
  for (auto& sl : d->slot) {
    if (sl.send_dev) (void)hipFree(sl.send_dev);
    if (sl.recv_dev) (void)hipFree(sl.recv_dev);
    if (sl.recv_host) (void)hipHostFree(sl.recv_host);
    if (sl.done) (void)hipEventDestroy(sl.done);
  }
  if (d->ready) (void)hipEventDestroy(d->ready);
  if (d->xstream) (void)hipStreamDestroy(d->xstream);
  if (d->comm && R) (void)R->CommDestroy(d->comm);
  if (d->recv_dev) (void)hipFree(d->recv_dev);
  if (d->send_dev) (void)hipFree(d->send_dev);
  if (d->recv_host) (void)hipHostFree(d->recv_host);
  if (d->flag_dev) (void)hipFree(d->flag_dev);
  delete d;
}

int mscnn_dist_rank(const mscnn_dist* d) { return d ? d->rank : -1; }
int mscnn_dist_world(const mscnn_dist* d) { return d ? d->world : 0; }

int mscnn_dist_all_gather_device(mscnn_dist* d, const void* send_dev, void* stream, const void** gathered_dev) {
  Rccl* R = rccl();
  DIST_REQUIRE(R && d && send_dev, "dist all_gather: bad argument");
  // the pack travels with its writer's rank in header word 3 (the final stage leaves 0 there): every receiver can check that slot r
  // of the gathered buffer really came from rank r
  hipStream_t cs = reinterpret_cast<hipStream_t>(stream);
  DIST_HIP(hipMemcpyAsync(d->send_dev, send_dev, d->pack_bytes, hipMemcpyDeviceToDevice, cs));
  DIST_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(static_cast<char*>(d->send_dev) + 12), d->rank, 1, cs));
  DIST_NCCL(R->AllGather(d->send_dev, d->recv_dev, d->pack_bytes, ncclChar, d->comm, cs));
  if (gathered_dev) *gathered_dev = d->recv_dev;
  return 0;
}

int mscnn_dist_all_gather(mscnn_dist* d, const void* send_dev, void* stream, const void** gathered_host) {
  const int rc = mscnn_dist_all_gather_device(d, send_dev, stream, nullptr);
  if (rc) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  DIST_HIP(hipMemcpyAsync(d->recv_host, d->recv_dev, d->pack_bytes * d->world, hipMemcpyDeviceToHost, st));
  DIST_HIP(hipStreamSynchronize(st));
  if (gathered_host) *gathered_host = d->recv_host;
  return 0;
}

// ---- pipelined form: the collective and the D2H copy of step i run on the communicator's own stream while the compute stream goes
// on with step i + 1; the host takes step i's packs one step later.  begin() copies the pack (device to device, on the compute
// stream: it may be overwritten by the next frame's final stage right after) and returns at once; end() waits for the OLDEST
// exchange in flight.  At most two may be in flight.
static int ensure_pipeline(mscnn_dist* d) {
  if (d->xstream) return 0;
  DIST_HIP(hipSetDevice(d->device));
  DIST_HIP(hipStreamCreateWithFlags(&d->xstream, hipStreamNonBlocking));
  DIST_HIP(hipEventCreateWithFlags(&d->ready, hipEventDisableTiming));
  for (auto& sl : d->slot) {
    DIST_HIP(hipMalloc(&sl.send_dev, d->pack_bytes));
    DIST_HIP(hipMalloc(&sl.recv_dev, d->pack_bytes * d->world));
    DIST_HIP(hipHostMalloc(&sl.recv_host, d->pack_bytes * d->world, hipHostMallocDefault));
    DIST_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
  }
  return 0;
}

int mscnn_dist_all_gather_begin(mscnn_dist* d, const void* send_dev, void* stream) {
  Rccl* R = rccl();
  DIST_REQUIRE(R && d && send_dev, "dist all_gather_begin: bad argument");
  DIST_REQUIRE(d->inflight < 2, "dist all_gather_begin: two exchanges already in flight (call mscnn_dist_all_gather_end)");
  const int rc = ensure_pipeline(d);
  if (rc) return rc;
  mscnn_dist::Slot& sl = d->slot[d->head];
  hipStream_t cs = reinterpret_cast<hipStream_t>(stream);
  DIST_HIP(hipMemcpyAsync(sl.send_dev, send_dev, d->pack_bytes, hipMemcpyDeviceToDevice, cs));
  DIST_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(static_cast<char*>(sl.send_dev) + 12), d->rank, 1, cs));      // (see all_gather_device)
  DIST_HIP(hipEventRecord(d->ready, cs));
  DIST_HIP(hipStreamWaitEvent(d->xstream, d->ready, 0));
  DIST_NCCL(R->AllGather(sl.send_dev, sl.recv_dev, d->pack_bytes, ncclChar, d->comm, d->xstream));
  DIST_HIP(hipMemcpyAsync(sl.recv_host, sl.recv_dev, d->pack_bytes * d->world, hipMemcpyDeviceToHost, d->xstream));
  DIST_HIP(hipEventRecord(sl.done, d->xstream));
  sl.busy = true;
  d->head ^= 1;
  ++d->inflight;
  return 0;
}

int mscnn_dist_all_gather_end(mscnn_dist* d, const void** gathered_host) {
  DIST_REQUIRE(d && gathered_host, "dist all_gather_end: bad argument");
  DIST_REQUIRE(d->inflight > 0, "dist all_gather_end: nothing in flight");
  mscnn_dist::Slot& sl = d->slot[d->tail];
  DIST_HIP(hipEventSynchronize(sl.done));
  sl.busy = false;
  d->tail ^= 1;
  --d->inflight;
  *gathered_host = sl.recv_host;      // valid until the NEXT begin(): with one exchange in flight that begin() takes this very slot
  return 0;
}

int mscnn_dist_barrier(mscnn_dist* d, void* stream) {
  Rccl* R = rccl();
  DIST_REQUIRE(R && d, "dist barrier: bad argument");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  DIST_NCCL(R->AllReduce(d->flag_dev, d->flag_dev + 1, 1, ncclInt, ncclSum, d->comm, st));
  DIST_HIP(hipStreamSynchronize(st));
  return 0;
}

}  // extern "C"
