// TEST DOUBLE, not product code: the eight RCCL entry points csrc/comm.hip binds, implemented over a POSIX
// shared-memory segment so that TWO (or more) rank processes can run the real N > 1 code of libxdet_hip.so -- id-file
// rendezvous, ncclCommInitRank, the pack + all-gather event protocol, the scalar collectives, the byte all-gather, the
// watchdog -- on a box with ONE GPU (RCCL itself refuses two ranks on one device).  Loaded through XDET_RCCL_LIB by
// tests/test_gpu_two_ranks.py only.  What it cannot show: that RCCL's own transports work between GPUs.
//
// Semantics kept: every collective is ENQUEUED on the caller's stream and returns; data moves in stream order
// (hipMemcpyAsync device -> this rank's slot of the pinned, registered segment; a host function that waits until every
// rank has arrived; hipMemcpyAsync of all slots -> device; a second arrival barrier before the slots are reused).  A rank
// whose peer died therefore hangs ON THE STREAM, exactly like a real collective -- which is what the communicator's
// watchdog has to catch.  ncclCommAbort releases the waiting host functions.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

extern "C" {

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5, ncclRemoteError = 6, ncclInProgress = 7 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5,
               ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat = 7, ncclFloat64 = 8, ncclDouble = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;

static const size_t kSlot = 64u << 20;          // bytes per rank and collective

struct Shared {
  std::atomic<int> arrived[2];                  // two alternating arrival counters
  std::atomic<long long> generation[2];
  std::atomic<int> attached;
};
struct Comm {
  int rank, nranks;
  Shared* sh;
  char* data;                                   // nranks slots of kSlot bytes behind the header
  size_t bytes;
  char name[64];
  std::atomic<int> aborted;
  long long gen[2];
};
typedef Comm* ncclComm_t;

static size_t dtype_size(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
  }
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error (fake RCCL)" : "fake RCCL error"; }
ncclResult_t ncclGetVersion(int* v) { *v = 99999; return ncclSuccess; }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "/xdet_fake_rccl_%ld_%lld", (long)getpid(),
           (long long)std::chrono::steady_clock::now().time_since_epoch().count());
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  Comm* c = new Comm();
  c->rank = rank; c->nranks = nranks; c->aborted = 0; c->gen[0] = c->gen[1] = 0;
  snprintf(c->name, sizeof(c->name), "%s", id.internal[0] ? id.internal : "/xdet_fake_rccl_solo");
  c->bytes = 4096 + (size_t)nranks * kSlot;
  int fd = -1;
  if (rank == 0) {
    shm_unlink(c->name);
    fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) return ncclSystemError;
  } else {
    for (int i = 0; i < 3000 && fd < 0; ++i) {          // up to 60 s for rank 0 to create the segment
      fd = shm_open(c->name, O_RDWR, 0600);
      struct stat st;
      if (fd >= 0 && (fstat(fd, &st) != 0 || (size_t)st.st_size < c->bytes)) { close(fd); fd = -1; }
      if (fd < 0) std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
    if (fd < 0) return ncclSystemError;
  }
  void* p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return ncclSystemError;
  c->sh = reinterpret_cast<Shared*>(p);
  c->data = reinterpret_cast<char*>(p) + 4096;
  if (hipHostRegister(p, c->bytes, hipHostRegisterDefault) != hipSuccess) return ncclUnhandledCudaError;
  c->sh->attached.fetch_add(1);
  for (int i = 0; c->sh->attached.load() < nranks; ++i) {  // "collective" init: wait for every rank
    if (i > 6000) return ncclSystemError;
    std::this_thread::sleep_for(std::chrono::milliseconds(10));
  }
  *out = c;
  return ncclSuccess;
}

static void release(Comm* c) {
  c->aborted = 1;
  std::this_thread::sleep_for(std::chrono::milliseconds(50));   // let spinning host functions see the flag
  (void)hipHostUnregister(c->sh);
  munmap(c->sh, c->bytes);
  if (c->rank == 0) shm_unlink(c->name);
  delete c;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { if (c) release(c); return ncclSuccess; }
ncclResult_t ncclCommAbort(ncclComm_t c) { if (c) release(c); return ncclSuccess; }
ncclResult_t ncclCommGetAsyncError(ncclComm_t, ncclResult_t* r) { *r = ncclSuccess; return ncclSuccess; }

struct Arrive { Comm* c; int which; long long gen; };
static void arrive_and_wait(void* arg) {          // host function, runs in stream order; must not call HIP
  Arrive* a = static_cast<Arrive*>(arg);
  Comm* c = a->c;
  Shared* sh = c->sh;
  if (sh->arrived[a->which].fetch_add(1) + 1 == c->nranks) {
    sh->arrived[a->which].store(0);
    sh->generation[a->which].fetch_add(1);
  }
  while (sh->generation[a->which].load() <= a->gen && !c->aborted.load())
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  delete a;
}
static ncclResult_t barrier_on_stream(Comm* c, int which, hipStream_t s) {
  Arrive* a = new Arrive{c, which, c->gen[which]++};
  return hipLaunchHostFunc(s, arrive_and_wait, a) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t s) {
  const size_t n = count * dtype_size(t);
  if (n > kSlot) return ncclInvalidArgument;
  if (hipMemcpyAsync(c->data + (size_t)c->rank * kSlot, send, n, hipMemcpyDeviceToHost, s) != hipSuccess) return ncclUnhandledCudaError;
  if (barrier_on_stream(c, 0, s) != ncclSuccess) return ncclUnhandledCudaError;
  for (int r = 0; r < c->nranks; ++r)
    if (hipMemcpyAsync(static_cast<char*>(recv) + (size_t)r * n, c->data + (size_t)r * kSlot, n, hipMemcpyHostToDevice, s) != hipSuccess)
      return ncclUnhandledCudaError;
  return barrier_on_stream(c, 1, s);
}

struct Reduce { Comm* c; size_t count; ncclRedOp_t op; double* out; };
static void reduce_doubles(void* arg) {
  Reduce* r = static_cast<Reduce*>(arg);
  for (size_t i = 0; i < r->count; ++i) {
    double v = reinterpret_cast<double*>(r->c->data)[i];
    for (int k = 1; k < r->c->nranks; ++k) {
      const double x = reinterpret_cast<double*>(r->c->data + (size_t)k * kSlot)[i];
      v = r->op == ncclMax ? (x > v ? x : v) : r->op == ncclMin ? (x < v ? x : v) : r->op == ncclProd ? v * x : v + x;
    }
    r->out[i] = v;
  }
  delete r;
}
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t s) {
  if (t != ncclDouble || count * 8 > kSlot / 2) return ncclInvalidArgument;       // all the product uses
  if (hipMemcpyAsync(c->data + (size_t)c->rank * kSlot, send, count * 8, hipMemcpyDeviceToHost, s) != hipSuccess) return ncclUnhandledCudaError;
  if (barrier_on_stream(c, 0, s) != ncclSuccess) return ncclUnhandledCudaError;
  double* scratch = reinterpret_cast<double*>(c->data + (size_t)c->rank * kSlot + kSlot / 2);   // this rank's private half
  if (hipLaunchHostFunc(s, reduce_doubles, new Reduce{c, count, op, scratch}) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpyAsync(recv, scratch, count * 8, hipMemcpyHostToDevice, s) != hipSuccess) return ncclUnhandledCudaError;
  return barrier_on_stream(c, 1, s);
}

}  // extern "C"
