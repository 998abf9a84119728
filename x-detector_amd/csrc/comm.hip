// Multi-GPU exchange of libxdet_hip.so: one process per GPU, images sharded by rank, ONE all-gather
// of the fixed-size padded detections per step over RCCL (xGMI inside a node).  SURVEY.md 8e; the
// reference has no distributed code at all (a single tf.estimator session at batch 1,
// light_head_rfcn_eval.py:212,466-499), so this layer is new.
//
// No PyTorch, no MPI: ranks rendezvous through a file -- rank 0 writes the 128-byte ncclUniqueId
// (write to a temporary name + rename, so a reader never sees a partial id), the others poll for
// it -- and then call ncclCommInitRank on their current HIP device.  librccl.so (573 MB) is bound
// with dlopen on the first xdet_comm_init, so single-GPU users never map it.
//
// Stream protocol of xdet_comm_allgather_detections (no host synchronisation anywhere):
//     producer streams (the nets' forward streams)      comm stream (owned by the communicator)
//       ... forward k ... bboxes_eval -> det buffers
//       record ev_in[i]            ------------------->  wait ev_in[*]
//                                                        pack_detections_kernel (score|box records)
//       wait ev_packed[*] <------------------------------ record ev_packed[k & 1]
//       ... forward k+1 (may overwrite det buffers) ...   ncclAllGather(packed -> gathered)
//                                                         record ev_done
// so the gather of step k runs under the compute of step k+1; xdet_comm_wait() makes a stream
// (or, with stream == NULL, the host) wait for ev_done.  [*] With ONE pair of det buffers the producers
// must wait for pack k before forward k+1 may overwrite them -- which joins all producer streams once per
// step.  A caller that alternates between TWO pairs (det_double_buffered != 0: forward k+1 writes the
// pair that call k-1 packed) waits for pack k-1 instead, which finished a step ago: no join at all.
#include "common.h"

#include <rccl/rccl.h>

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace xdet {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  int (*GetVersion)(int*) = nullptr;
  // optional (the watchdog degrades to a plain timeout without them)
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
};

static RcclApi g_rccl;
static std::mutex g_rccl_mu;
static bool g_rccl_overridden = false;   // bound through XDET_RCCL_LIB (+ XDET_ALLOW_RCCL_OVERRIDE=1)
static std::string g_rccl_path;           // file the collective entry points were resolved from (dladdr)

static int load_rccl() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.handle) return XDET_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  // XDET_RCCL_LIB: another build of RCCL -- or the test double of tests/fake_rccl, which lets two rank processes run
  // this file's N > 1 code on a one-GPU box (RCCL refuses two ranks on one device)
  if (const char* over = getenv("XDET_RCCL_LIB")) {
    // a stand-in must be asked for twice: a bench or a service must never run its collectives on something that is not
    // RCCL because a test's environment leaked into it
    const char* allow = getenv("XDET_ALLOW_RCCL_OVERRIDE");
    if (!allow || strcmp(allow, "1") != 0) {
      set_last_error(std::string("XDET_RCCL_LIB=") + over + " is set but XDET_ALLOW_RCCL_OVERRIDE=1 is not: refusing to bind "
                     "another library in place of librccl.so");
      return XDET_ERR_STATE;
    }
    h = dlopen(over, RTLD_NOW | RTLD_GLOBAL);
    g_rccl_overridden = true;
    if (!h) {
      set_last_error(std::string("cannot load XDET_RCCL_LIB=") + over + ": " + dlerror());
      return XDET_ERR_STATE;
    }
  }
  for (const char* n : names) {
    if (h) break;
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  }
  if (!h) {
    set_last_error(std::string("cannot load librccl.so: ") + dlerror());
    return XDET_ERR_STATE;
  }
  RcclApi a;
  a.handle = h;
#define XDET_SYM(field, name)                                                  \
  *reinterpret_cast<void**>(&a.field) = dlsym(h, name);                        \
  if (!a.field) {                                                              \
    set_last_error(std::string("librccl.so lacks ") + name);                   \
    dlclose(h);                                                                \
    return XDET_ERR_STATE;                                                     \
  }
  XDET_SYM(GetUniqueId, "ncclGetUniqueId")
  XDET_SYM(CommInitRank, "ncclCommInitRank")
  XDET_SYM(CommDestroy, "ncclCommDestroy")
  XDET_SYM(AllGather, "ncclAllGather")
  XDET_SYM(AllReduce, "ncclAllReduce")
  XDET_SYM(GetErrorString, "ncclGetErrorString")
  XDET_SYM(GetVersion, "ncclGetVersion")
#undef XDET_SYM
  *reinterpret_cast<void**>(&a.CommAbort) = dlsym(h, "ncclCommAbort");
  *reinterpret_cast<void**>(&a.CommGetAsyncError) = dlsym(h, "ncclCommGetAsyncError");
  {
    Dl_info di;
    memset(&di, 0, sizeof(di));
    if (dladdr(reinterpret_cast<void*>(a.AllGather), &di) && di.dli_fname) {
      char real[4096];
      g_rccl_path = realpath(di.dli_fname, real) ? real : di.dli_fname;
    }
  }
  g_rccl = a;
  return XDET_OK;
}

static int rccl_fail(ncclResult_t r, const char* what) {
  set_last_error(std::string("RCCL error in ") + what + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"));
  return XDET_ERR_HIP;
}
#define XDET_RCCL(expr)                                        \
  do {                                                         \
    ncclResult_t _r = (expr);                                  \
    if (_r != ncclSuccess) return rccl_fail(_r, #expr);        \
  } while (0)

// score [B,C,K] + boxes [B,C,K,4] -> one record [B,C,K,5] = (score | ymin xmin ymax xmax): the unit the
// all-gather moves (20 classes x 200 slots x 5 floats = 80 KB per image)
__global__ void pack_detections_kernel(const float* __restrict__ scores, const float* __restrict__ boxes,
                                       float* __restrict__ packed, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 b = reinterpret_cast<const float4*>(boxes)[i];
    float* o = packed + i * 5;
    o[0] = scores[i];
    o[1] = b.x; o[2] = b.y; o[3] = b.z; o[4] = b.w;
  }
}

// make the communicator's device current for the call (the caller may have switched devices since init)
struct CommDeviceGuard {
  int prev = -1, want = -1;
  explicit CommDeviceGuard(int dev) : want(dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != want) (void)hipSetDevice(want);
  }
  ~CommDeviceGuard() {
    if (prev >= 0 && prev != want) (void)hipSetDevice(prev);
  }
};

struct Comm {
  int rank = 0, world = 1, device = 0;
  double timeout_s = 300.0;       // watchdog: a host wait on the communicator's stream longer than this is a dead peer
  bool dead = false;              // set by the watchdog; every later call fails fast
  char* d_bytes = nullptr;        // device staging of xdet_comm_allgather_bytes
  size_t d_bytes_cap = 0;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev_packed[2] = {nullptr, nullptr}, ev_done = nullptr;
  std::vector<hipEvent_t> ev_in;
  double* d_scalar = nullptr;     // device scratch of the scalar collectives (barrier, max)
  // pinned host staging of the host-buffer collectives: a D2H copy into PAGEABLE memory blocks the host until the
  // stream has drained, i.e. behind a dead peer it would hang inside hipMemcpyAsync before the watchdog is ever polled
  char* h_pinned = nullptr;
  size_t h_pinned_cap = 0;
  int64_t gathers = 0;
  int pinned(size_t bytes) {
    if (bytes <= h_pinned_cap) return XDET_OK;
    if (h_pinned) (void)hipHostFree(h_pinned);
    h_pinned = nullptr;
    h_pinned_cap = 0;
    XDET_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_pinned), std::max<size_t>(bytes, 4096), hipHostMallocDefault));
    h_pinned_cap = std::max<size_t>(bytes, 4096);
    return XDET_OK;
  }

  ~Comm() {
    // a communicator the watchdog gave up on is aborted, not destroyed: ncclCommDestroy would wait for the dead peer
    // (without ncclCommAbort in the bound library the dead communicator is leaked on purpose)
    if (comm) {
      if (!dead) (void)g_rccl.CommDestroy(comm);
      else if (g_rccl.CommAbort) (void)g_rccl.CommAbort(comm);
    }
    if (h_pinned) (void)hipHostFree(h_pinned);
    if (d_bytes) (void)hipFree(d_bytes);
    for (hipEvent_t e : ev_in) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev_packed)
      if (e) (void)hipEventDestroy(e);
    if (ev_done) (void)hipEventDestroy(ev_done);
    if (d_scalar) (void)hipFree(d_scalar);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

static int write_id_file(const std::string& path, const ncclUniqueId& id) {
  const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) {
    set_last_error("comm_init: cannot create " + tmp);
    return XDET_ERR_STATE;
  }
  const size_t n = fwrite(id.internal, 1, NCCL_UNIQUE_ID_BYTES, f);
  fclose(f);
  if (n != NCCL_UNIQUE_ID_BYTES || rename(tmp.c_str(), path.c_str()) != 0) {
    (void)unlink(tmp.c_str());
    set_last_error("comm_init: cannot publish " + path);
    return XDET_ERR_STATE;
  }
  return XDET_OK;
}

static int read_id_file(const std::string& path, int timeout_s, ncclUniqueId* id) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    FILE* f = fopen(path.c_str(), "rb");
    if (f) {
      const size_t n = fread(id->internal, 1, NCCL_UNIQUE_ID_BYTES, f);
      fclose(f);
      if (n == NCCL_UNIQUE_ID_BYTES) return XDET_OK;   // rename() is atomic: a visible file is complete
    }
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s)) {
      set_last_error("comm_init: timed out waiting for rank 0's id file " + path);
      return XDET_ERR_STATE;
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
  }
}

// Host wait for `ev` on the communicator's stream with a watchdog.  hipEventSynchronize / hipStreamSynchronize would
// block forever when a peer dies inside a collective (its ring neighbour never sends); here the event is polled,
// RCCL's asynchronous error state is consulted, and after timeout_s the communicator is aborted and the call fails
// with XDET_ERR_STATE -- the launcher (xdet.launch) then sees a non-zero exit instead of a hung rank.
static int watchdog_wait(Comm* c, hipEvent_t ev, const char* what) {
  if (c->dead) {
    set_last_error(std::string(what) + ": the communicator was aborted by an earlier watchdog timeout");
    return XDET_ERR_STATE;
  }
  const auto t0 = std::chrono::steady_clock::now();
  int spins = 0;
  for (;;) {
    const hipError_t q = hipEventQuery(ev);
    if (q == hipSuccess) return XDET_OK;
    if (q != hipErrorNotReady) return hip_fail(q, "hipEventQuery(comm event)", __FILE__, __LINE__);
    std::string why;
    if (g_rccl.CommGetAsyncError && c->comm) {
      ncclResult_t ar = ncclSuccess;
      if (g_rccl.CommGetAsyncError(c->comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress)
        why = std::string("RCCL reports an asynchronous error (") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(ar) : "?") + ")";
    }
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (why.empty() && el > c->timeout_s) {
      char buf[320];
      snprintf(buf, sizeof(buf), "this wait on the communicator stream exceeded its timeout of %.0f s (waited %.0f s; a peer rank "
               "died, hangs, or legitimately needs longer: XDET_COMM_TIMEOUT_S / xdet_comm_set_timeout)", c->timeout_s, el);
      why = buf;
    }
    if (!why.empty()) {
      c->dead = true;
      if (g_rccl.CommAbort && c->comm) {
        (void)g_rccl.CommAbort(c->comm);
        c->comm = nullptr;
      }
      set_last_error(std::string(what) + ": " + why + "; rank " + std::to_string(c->rank) + " of " + std::to_string(c->world) +
                     " aborted its communicator");
      return XDET_ERR_STATE;
    }
    if (++spins < 2000) std::this_thread::yield();          // short waits stay cheap
    else std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
}

}  // namespace xdet

using namespace xdet;

extern "C" {

int xdet_comm_init(void** comm_out, int rank, int world, const char* unique_id_path, int timeout_s) {
  XDET_REQUIRE(comm_out && world >= 1 && rank >= 0 && rank < world, "comm_init: need 0 <= rank < world");
  XDET_REQUIRE(world == 1 || (unique_id_path && unique_id_path[0]), "comm_init: unique_id_path is required for world > 1");
  XDET_TRY(load_rccl());
  std::unique_ptr<Comm> c(new Comm());
  c->rank = rank;
  c->world = world;
  XDET_HIP(hipGetDevice(&c->device));
  ncclUniqueId id;
  memset(&id, 0, sizeof(id));
  if (const char* t = getenv("XDET_COMM_TIMEOUT_S")) {
    const double v = atof(t);
    if (v > 0) c->timeout_s = v;
  }
  if (rank == 0) {
    // a file left behind by an earlier launch that resolved to the same name must never be read as this launch's id:
    // remove it before the new id exists (the launcher additionally hands every launch a fresh directory)
    if (world > 1) (void)unlink(unique_id_path);
    XDET_RCCL(g_rccl.GetUniqueId(&id));
    if (world > 1) XDET_TRY(write_id_file(unique_id_path, id));
  } else {
    XDET_TRY(read_id_file(unique_id_path, timeout_s > 0 ? timeout_s : 120, &id));
  }
  XDET_RCCL(g_rccl.CommInitRank(&c->comm, world, id, rank));
  // ncclCommInitRank is collective: every rank has read the id by now, the file has served its purpose
  if (rank == 0 && world > 1) (void)unlink(unique_id_path);
  XDET_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  for (hipEvent_t& e : c->ev_packed) {
    XDET_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    XDET_HIP(hipEventRecord(e, c->stream));
  }
  XDET_HIP(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
  XDET_HIP(hipMalloc(reinterpret_cast<void**>(&c->d_scalar), 2 * sizeof(double)));
  XDET_HIP(hipEventRecord(c->ev_done, c->stream));
  *comm_out = c.release();
  return XDET_OK;
}

int xdet_comm_destroy(void* comm) {
  Comm* c = static_cast<Comm*>(comm);
  if (!c) return XDET_OK;
  CommDeviceGuard guard(c->device);
  if (c->stream && !c->dead) {
    // drain under the watchdog: a destroy behind a dead peer must not hang either
    if (hipEventRecord(c->ev_done, c->stream) == hipSuccess) (void)watchdog_wait(c, c->ev_done, "comm_destroy");
  }
  delete c;
  return XDET_OK;
}

int xdet_comm_info(void* comm, int* rank, int* world, int* device, int* rccl_version) {
  Comm* c = static_cast<Comm*>(comm);
  XDET_REQUIRE(c, "comm is NULL");
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (device) *device = c->device;
  if (rccl_version) {
    *rccl_version = 0;
    (void)g_rccl.GetVersion(rccl_version);
  }
  return XDET_OK;
}

int xdet_pack_detections(const float* det_scores, const float* det_boxes, int64_t n_slots, float* packed, void* stream) {
  XDET_REQUIRE(det_scores && det_boxes && packed && n_slots >= 0, "pack_detections: bad arguments");
  if (n_slots == 0) return XDET_OK;
  const int blocks = (int)std::min<int64_t>(cdiv(n_slots, 256), 2048);
  hipLaunchKernelGGL(pack_detections_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), det_scores,
                     det_boxes, packed, n_slots);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

int xdet_comm_allgather_detections(void* comm, const float* det_scores, const float* det_boxes, int n_images,
                                   int n_fg_classes, int topk, float* packed_local, float* gathered,
                                   void* const* producer_streams, int n_producers, int det_double_buffered) {
  Comm* c = static_cast<Comm*>(comm);
  XDET_REQUIRE(c && det_scores && det_boxes && packed_local && gathered, "allgather_detections: NULL argument");
  XDET_REQUIRE(n_images > 0 && n_fg_classes > 0 && topk > 0 && n_producers >= 0, "allgather_detections: bad sizes");
  XDET_REQUIRE(n_producers == 0 || producer_streams, "allgather_detections: producer_streams is NULL");
  if (c->dead) {
    set_last_error("allgather_detections: the communicator was aborted by the watchdog");
    return XDET_ERR_STATE;
  }
  CommDeviceGuard dev_guard(c->device);
  while ((int)c->ev_in.size() < n_producers) {
    hipEvent_t e;
    XDET_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->ev_in.push_back(e);
  }
  for (int i = 0; i < n_producers; ++i) {
    hipStream_t ps = reinterpret_cast<hipStream_t>(producer_streams[i]);
    XDET_HIP(hipEventRecord(c->ev_in[i], ps));
    XDET_HIP(hipStreamWaitEvent(c->stream, c->ev_in[i], 0));
  }
  const int64_t slots = (int64_t)n_images * n_fg_classes * topk;
  XDET_TRY(xdet_pack_detections(det_scores, det_boxes, slots, packed_local, c->stream));
  const int cur = (int)(c->gathers & 1);
  XDET_HIP(hipEventRecord(c->ev_packed[cur], c->stream));
  // whatever the producers enqueue from now on (the next forward) may overwrite det buffers: order it behind
  // the pack that last read the pair it will write -- this call's pack with one pair, the previous call's
  // with two alternating pairs
  const hipEvent_t guard = c->ev_packed[det_double_buffered ? cur ^ 1 : cur];
  for (int i = 0; i < n_producers; ++i)
    XDET_HIP(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(producer_streams[i]), guard, 0));
  XDET_RCCL(g_rccl.AllGather(packed_local, gathered, (size_t)slots * 5, ncclFloat, c->comm, c->stream));
  XDET_HIP(hipEventRecord(c->ev_done, c->stream));
  ++c->gathers;
  return XDET_OK;
}

int xdet_comm_wait(void* comm, void* stream) {
  Comm* c = static_cast<Comm*>(comm);
  XDET_REQUIRE(c, "comm is NULL");
  CommDeviceGuard guard(c->device);
  if (stream) XDET_HIP(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), c->ev_done, 0));
  else return watchdog_wait(c, c->ev_done, "comm_wait");
  return XDET_OK;
}

int xdet_comm_allreduce_max(void* comm, double* value_host) {
  Comm* c = static_cast<Comm*>(comm);
  XDET_REQUIRE(c && value_host, "allreduce_max: NULL argument");
  if (c->dead) {
    set_last_error("allreduce_max: the communicator was aborted by the watchdog");
    return XDET_ERR_STATE;
  }
  CommDeviceGuard guard(c->device);
  // drain first: the staging buffer may still be the target of an earlier call's copy
  XDET_HIP(hipEventRecord(c->ev_done, c->stream));
  XDET_TRY(watchdog_wait(c, c->ev_done, "allreduce_max"));
  XDET_TRY(c->pinned(2 * sizeof(double)));
  double* hp = reinterpret_cast<double*>(c->h_pinned);
  hp[0] = *value_host;
  XDET_HIP(hipMemcpyAsync(c->d_scalar, hp, sizeof(double), hipMemcpyHostToDevice, c->stream));
  XDET_RCCL(g_rccl.AllReduce(c->d_scalar, c->d_scalar + 1, 1, ncclDouble, ncclMax, c->comm, c->stream));
  XDET_HIP(hipMemcpyAsync(hp + 1, c->d_scalar + 1, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  XDET_HIP(hipEventRecord(c->ev_done, c->stream));
  XDET_TRY(watchdog_wait(c, c->ev_done, "allreduce_max"));
  *value_host = hp[1];
  return XDET_OK;
}

// Small host-buffer all-gather (rank records, per-rank rates): send `bytes` from every rank, receive world * bytes in
// rank order.  Staged through device memory and moved by ncclAllGather on the communicator's stream, i.e. over the
// same transport as the detections -- what arrives proves which ranks took part.
int xdet_comm_allgather_bytes(void* comm, const void* send_host, void* recv_host, size_t bytes) {
  Comm* c = static_cast<Comm*>(comm);
  XDET_REQUIRE(c && send_host && recv_host && bytes > 0 && bytes <= (1u << 20), "allgather_bytes: bad arguments (1 MiB per rank at most)");
  if (c->dead) {
    set_last_error("allgather_bytes: the communicator was aborted by the watchdog");
    return XDET_ERR_STATE;
  }
  CommDeviceGuard guard(c->device);
  const size_t need = bytes * (size_t)(c->world + 1);
  if (need > c->d_bytes_cap) {
    // (drain under the watchdog: an earlier collective behind a dead peer must not hang the reallocation either)
    XDET_HIP(hipEventRecord(c->ev_done, c->stream));
    XDET_TRY(watchdog_wait(c, c->ev_done, "allgather_bytes"));
    if (c->d_bytes) (void)hipFree(c->d_bytes);
    c->d_bytes = nullptr;
    c->d_bytes_cap = 0;
    XDET_HIP(hipMalloc(reinterpret_cast<void**>(&c->d_bytes), need));
    c->d_bytes_cap = need;
  }
  // through pinned staging: both copies are then truly asynchronous and the host only ever waits in the watchdog
  XDET_HIP(hipEventRecord(c->ev_done, c->stream));
  XDET_TRY(watchdog_wait(c, c->ev_done, "allgather_bytes"));
  XDET_TRY(c->pinned(need));
  memcpy(c->h_pinned, send_host, bytes);
  XDET_HIP(hipMemcpyAsync(c->d_bytes, c->h_pinned, bytes, hipMemcpyHostToDevice, c->stream));
  XDET_RCCL(g_rccl.AllGather(c->d_bytes, c->d_bytes + bytes, bytes, ncclChar, c->comm, c->stream));
  XDET_HIP(hipMemcpyAsync(c->h_pinned + bytes, c->d_bytes + bytes, bytes * (size_t)c->world, hipMemcpyDeviceToHost, c->stream));
  XDET_HIP(hipEventRecord(c->ev_done, c->stream));
  XDET_TRY(watchdog_wait(c, c->ev_done, "allgather_bytes"));
  memcpy(recv_host, c->h_pinned + bytes, bytes * (size_t)c->world);
  return XDET_OK;
}

int xdet_comm_library(char* path_buf, int buflen, int* overridden) {
  XDET_REQUIRE(path_buf && buflen > 0, "comm_library: need a buffer");
  XDET_TRY(load_rccl());
  snprintf(path_buf, (size_t)buflen, "%s", g_rccl_path.c_str());
  if (overridden) *overridden = g_rccl_overridden ? 1 : 0;
  return XDET_OK;
}

int xdet_comm_set_timeout(void* comm, double seconds) {
  Comm* c = static_cast<Comm*>(comm);
  XDET_REQUIRE(c && seconds > 0, "comm_set_timeout: need a communicator and a positive timeout");
  c->timeout_s = seconds;
  return XDET_OK;
}

int xdet_comm_barrier(void* comm) {
  double v = 0.0;
  return xdet_comm_allreduce_max(comm, &v);
}

}  // extern "C"
