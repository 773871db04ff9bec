// rccl_loopback -- TEST INFRASTRUCTURE, not part of the product: the RCCL entry points libmoptix.so binds (nine + the two optional ones)
// (csrc/moptix_api.hip RcclApi), implemented over POSIX shared memory + hipMemcpy, so that the N > 1 branches of
// moptix_gather_tiles / moptix_reduce_frame (pack -> send; grouped receives -> unpack; reduce) run as N processes on a ONE-GPU box.
// RCCL itself refuses a communicator whose ranks share a device ("Duplicate GPU detected", init.cc), and no multi-GPU box has been
// available to this project; what this exercises is everything on OUR side of the ncclXxx calls, not RCCL or xGMI.
// Selected by MOPTIX_RCCL_LIB=<path to this library>.
//
// Protocol: one shared segment per unique id, one mailbox per ordered (src, dst) pair: the sender waits for `full == 0`, copies a
// chunk device -> mailbox, sets `full`; the receiver waits for `full`, copies mailbox -> device, clears it.  Every wait has a
// deadline (ncclSystemError after 60 s): a missing peer fails the call instead of hanging the box.
//
// MOPTIX_LOOPBACK_STUCK=1 models what RCCL does when a peer never joins a collective: ncclRecv returns at once after putting a
// kernel on the caller's stream that spins until the communicator is aborted (ncclCommAbort raises a flag in host-mapped memory;
// the kernel also gives up by itself after 30 s so that no test can hang the GPU).  libmoptix.so's deadline (comm_wait: poll the
// stream, ncclCommGetAsyncError, ncclCommAbort) is tested against it (tests/test_gpu_rccl_loopback.py).
//
// MOPTIX_LOOPBACK_STUCK=2 models the OTHER place a missing call shows: on the host.  With a blocking communicator RCCL would sit inside
// ncclGroupEnd / ncclSend while the links to the peer come up; libmoptix.so therefore makes its communicators non-blocking
// (ncclCommInitRankConfig, config.blocking = 0), where the call returns ncclInProgress and ncclCommGetAsyncError keeps saying so.
// Here: on a non-blocking communicator ncclGroupEnd / ncclSend / ncclReduce return ncclInProgress and the communicator's state stays
// ncclInProgress until it is aborted -- libmoptix.so's comm_settle has to give up at its deadline.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

namespace {
constexpr int kMaxRanks = 8;
constexpr size_t kChunk = 1 << 20;      // bytes per mailbox
struct Mailbox { std::atomic<uint32_t> full; uint32_t bytes; char pad[56]; char data[kChunk]; };
struct Segment { std::atomic<uint32_t> arrived; char pad[60]; Mailbox box[kMaxRanks][kMaxRanks]; };
struct Comm { int rank, n; Segment* seg; char name[64]; uint32_t* abortHost; uint32_t* abortDev; int blocking; int inProgress; };
Comm* g_lastGrouped = nullptr;      // the communicator of the calls since ncclGroupStart (this transport's groups are one thread, one communicator)
bool host_stuck(const Comm* c) { const char* e = getenv("MOPTIX_LOOPBACK_STUCK"); return e && e[0] == '2' && c && !c->blocking; }

__global__ void k_stuck(const uint32_t* abortFlag, unsigned long long maxTicks) {      // a collective's kernel whose peer never arrives
  const unsigned long long t0 = wall_clock64();                                         // 100 MHz
  while (__hip_atomic_load(abortFlag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u && wall_clock64() - t0 < maxTicks) __builtin_amdgcn_s_sleep(127);
}

bool wait_for(std::atomic<uint32_t>& flag, uint32_t want) {
  const auto t0 = std::chrono::steady_clock::now();
  while (flag.load(std::memory_order_acquire) != want) {
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) return false;
    std::this_thread::sleep_for(std::chrono::microseconds(20));
  }
  return true;
}
ncclResult_t send_bytes(Comm* c, const void* dev, size_t bytes, int peer, hipStream_t stream) {
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;      // what the stream has queued (the pack kernel) is the payload
  Mailbox& m = c->seg->box[c->rank][peer];
  for (size_t off = 0; off < bytes || (bytes == 0 && off == 0); off += kChunk) {
    const size_t n = bytes - off < kChunk ? bytes - off : kChunk;
    if (!wait_for(m.full, 0)) return ncclSystemError;
    if (n && hipMemcpy(m.data, (const char*)dev + off, n, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    m.bytes = (uint32_t)n;
    m.full.store(1, std::memory_order_release);
    if (bytes == 0) break;
  }
  return ncclSuccess;
}
// into device memory, or (host != nullptr) added to a host accumulator of floats
ncclResult_t recv_bytes(Comm* c, void* dev, float* hostAdd, size_t bytes, int peer, hipStream_t stream) {
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  Mailbox& m = c->seg->box[peer][c->rank];
  for (size_t off = 0; off < bytes || (bytes == 0 && off == 0); off += kChunk) {
    if (!wait_for(m.full, 1)) return ncclSystemError;
    const size_t n = m.bytes;
    if (n != (bytes - off < kChunk ? bytes - off : kChunk)) return ncclInvalidArgument;      // the two sides disagree on the count
    if (hostAdd) { const float* src = (const float*)m.data; float* dst = hostAdd + off / 4; for (size_t i = 0; i < n / 4; i++) dst[i] += src[i]; }
    else if (n && hipMemcpy((char*)dev + off, m.data, n, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    m.full.store(0, std::memory_order_release);
    if (bytes == 0) break;
  }
  return ncclSuccess;
}
size_t type_bytes(ncclDataType_t t) { return t == ncclFloat ? 4 : 0; }
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof(*id));
  unsigned long long r = (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count() ^ ((unsigned long long)getpid() << 32);
  snprintf(id->internal, sizeof(id->internal), "/moptix_loopback_%d_%llx", (int)getpid(), r);
  const int fd = shm_open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);      // the creator is rank 0's process; the others open it
  if (fd < 0) return ncclSystemError;
  if (ftruncate(fd, sizeof(Segment)) != 0) { close(fd); return ncclSystemError; }      // zero-filled, pages appear when touched
  close(fd);
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  const int fd = shm_open(id.internal, O_RDWR, 0600);
  if (fd < 0) return ncclSystemError;
  void* p = mmap(nullptr, sizeof(Segment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return ncclSystemError;
  Comm* c = new Comm; c->rank = rank; c->n = nranks; c->seg = (Segment*)p;
  strncpy(c->name, id.internal, sizeof(c->name) - 1); c->name[sizeof(c->name) - 1] = 0;
  c->abortHost = nullptr; c->abortDev = nullptr; c->blocking = 1; c->inProgress = 0;
  if (hipHostMalloc((void**)&c->abortHost, 64, hipHostMallocMapped) == hipSuccess) {
    *c->abortHost = 0;
    if (hipHostGetDevicePointer((void**)&c->abortDev, c->abortHost, 0) != hipSuccess) c->abortDev = nullptr;
  }
  c->seg->arrived.fetch_add(1);
  const auto t0 = std::chrono::steady_clock::now();      // like ncclCommInitRank: returns once every rank has arrived
  while ((int)c->seg->arrived.load() < nranks) {
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) { munmap(p, sizeof(Segment)); delete c; return ncclSystemError; }
    std::this_thread::sleep_for(std::chrono::microseconds(100));
  }
  if (rank == 0) shm_unlink(c->name);      // every rank has mapped it: the name can go now, so a rank that dies later leaks nothing in /dev/shm
  *comm = (ncclComm_t)c;
  return ncclSuccess;
}
ncclResult_t ncclCommInitRankConfig(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank, ncclConfig_t* config) {
  const ncclResult_t r = ncclCommInitRank(comm, nranks, id, rank);
  if (r == ncclSuccess && config) ((Comm*)*comm)->blocking = config->blocking != 0;
  return r;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  Comm* c = (Comm*)comm;
  if (!c) return ncclInvalidArgument;
  munmap(c->seg, sizeof(Segment));
  delete c;                                  // (the 64 bytes of the abort flag stay mapped: a kernel may still be reading them)
  return ncclSuccess;
}
ncclResult_t ncclCommAbort(ncclComm_t comm) {
  Comm* c = (Comm*)comm;
  if (!c) return ncclInvalidArgument;
  if (c->abortHost) __atomic_store_n(c->abortHost, 1u, __ATOMIC_RELEASE);      // kernels of this communicator leave
  return ncclCommDestroy(comm);
}
ncclResult_t ncclCommGetAsyncError(ncclComm_t comm, ncclResult_t* asyncError) {
  if (!comm || !asyncError) return ncclInvalidArgument;
  *asyncError = ((Comm*)comm)->inProgress ? ncclInProgress : ncclSuccess;
  return ncclSuccess;
}
const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclSystemError: return "loopback: a peer did not arrive within 60 s (or shared memory failed)";
    case ncclInvalidArgument: return "loopback: invalid argument / the two sides disagree on a count";
    case ncclUnhandledCudaError: return "loopback: HIP call failed";
    default: return "loopback: error";
  }
}
ncclResult_t ncclGroupStart() { g_lastGrouped = nullptr; return ncclSuccess; }      // calls run where they are made: receives never depend on this rank's sends
ncclResult_t ncclGroupEnd() {
  Comm* c = g_lastGrouped; g_lastGrouped = nullptr;
  if (host_stuck(c)) { c->inProgress = 1; return ncclInProgress; }      // the peer's side of the connections never comes
  return ncclSuccess;
}
ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
  Comm* c = (Comm*)comm;
  if (!c || !type_bytes(type) || peer < 0 || peer >= c->n || peer == c->rank) return ncclInvalidArgument;
  return send_bytes(c, sendbuff, count * type_bytes(type), peer, stream);
}
ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
  Comm* c = (Comm*)comm;
  if (!c || !type_bytes(type) || peer < 0 || peer >= c->n || peer == c->rank) return ncclInvalidArgument;
  g_lastGrouped = c;
  if (host_stuck(c)) return ncclSuccess;       // queued in the group; ncclGroupEnd reports the state
  if (getenv("MOPTIX_LOOPBACK_STUCK")) {      // the peer never sends: what the caller gets from RCCL then is a kernel that does not end
    if (!c->abortDev) return ncclSystemError;
    k_stuck<<<1, 64, 0, stream>>>(c->abortDev, 3000000000ull);
    return hipGetLastError() == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
  }
  return recv_bytes(c, recvbuff, nullptr, count * type_bytes(type), peer, stream);
}
ncclResult_t ncclReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t type, ncclRedOp_t op, int root, ncclComm_t comm, hipStream_t stream) {
  Comm* c = (Comm*)comm;
  if (!c || type != ncclFloat || op != ncclSum || root < 0 || root >= c->n) return ncclInvalidArgument;
  if (c->rank != root) return send_bytes(c, sendbuff, count * 4, root, stream);
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  std::vector<float> acc(count);
  if (count && hipMemcpy(acc.data(), sendbuff, count * 4, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  for (int r = 0; r < c->n; r++) {      // rank order: a fixed summation order (RCCL's differs; the sample split's bound covers any order)
    if (r == root) continue;
    const ncclResult_t e = recv_bytes(c, nullptr, acc.data(), count * 4, r, stream);
    if (e != ncclSuccess) return e;
  }
  if (count && hipMemcpy(recvbuff, acc.data(), count * 4, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}

}  // extern "C"
