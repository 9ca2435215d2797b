// fake_rccl.cpp -- a stand-in for librccl.so (TEST INFRASTRUCTURE, loaded through FASTMC_RCCL_LIB): the ten nccl* entry points
// libfastmc.so binds (fast_amd/csrc/fastmc.hip: load_rccl), for ONE process.  It lets the communicator code of the library --
// clique set-up, grouped all-gather / all-reduce on the handles' streams, the queued forms, abort and destroy -- run with a world
// of 2 ... 8 "ranks" on the single GPU a test box has (FASTMC_TEST_VIRTUAL_RANKS=1), and without any GPU for the calls that need
// none (unique id, error strings).  Semantics kept: calls inside ncclGroupStart / ncclGroupEnd are collected and executed at the
// outermost ncclGroupEnd, every rank of a clique must have made the same call (else ncclInvalidUsage), in-place all-reduce,
// results ordered behind the work already on each rank's stream.  Not kept: asynchrony (ncclGroupEnd returns when the data is in
// place) and anything across processes (ncclCommInitRank accepts a world of one).
//
//   g++ -O1 -shared -fPIC -o libfake_rccl.so fake_rccl.cpp -ldl        (no HIP headers: the runtime is looked up in the process)
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <vector>

extern "C" {
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5, ncclRemoteError = 6, ncclInProgress = 7 } ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef struct FakeComm* ncclComm_t;
typedef int ncclDataType_t;   // ncclUint64 = 5, ncclDouble = 8 (rccl.h)
typedef int ncclRedOp_t;      // ncclSum = 0
typedef void* hipStream_t_;
}

namespace {
struct Clique {
  int n = 0;
  bool aborted = false;
  int live = 0;
};
struct Call {
  int kind;                   // 0 all-gather, 1 all-reduce
  const void* send;
  void* recv;
  size_t count;
  int dtype, op;
  hipStream_t_ stream;
};
}  // namespace
struct FakeComm {
  Clique* clique;
  int rank;
  int device;
  std::vector<Call> pending;
};

namespace {
std::mutex g_mu;
int g_depth = 0;
std::vector<FakeComm*> g_touched;          // communicators with pending calls of the open group
long g_calls[4] = {0, 0, 0, 0};            // all-gathers, all-reduces, groups executed, aborts (fake_rccl_counters)

// the HIP runtime already in the process (libfastmc.so links it); nullptr on a box without it
typedef int (*hipMemcpy_t)(void*, const void*, size_t, int);
typedef int (*hipStreamSynchronize_t)(void*);
typedef int (*hipSetDevice_t)(int);
hipMemcpy_t p_memcpy = nullptr;
hipStreamSynchronize_t p_sync = nullptr;
hipSetDevice_t p_setdev = nullptr;
bool hip_ready() {
  if (p_memcpy) return true;
  void* lib = dlopen("libamdhip64.so", RTLD_NOW | RTLD_NOLOAD);
  if (!lib) lib = dlopen("libamdhip64.so.7", RTLD_NOW | RTLD_NOLOAD);
  if (!lib) lib = dlopen("libamdhip64.so.6", RTLD_NOW | RTLD_NOLOAD);
  if (!lib) lib = dlopen("libamdhip64.so", RTLD_NOW);
  if (!lib) return false;
  p_memcpy = (hipMemcpy_t)dlsym(lib, "hipMemcpy");
  p_sync = (hipStreamSynchronize_t)dlsym(lib, "hipStreamSynchronize");
  p_setdev = (hipSetDevice_t)dlsym(lib, "hipSetDevice");
  return p_memcpy && p_sync && p_setdev;
}
size_t dtype_bytes(int t) { return (t == 5 || t == 8 || t == 4) ? 8 : ((t == 2 || t == 3 || t == 7) ? 4 : 1); }

// every rank of `cl` (in `comms`, by rank) has exactly one pending call: execute it
ncclResult_t execute(std::vector<FakeComm*>& comms) {
  const int n = (int)comms.size();
  const Call c0 = comms[0]->pending[0];
  for (int r = 0; r < n; ++r) {
    const Call& c = comms[r]->pending[0];
    if (c.kind != c0.kind || c.count != c0.count || c.dtype != c0.dtype || c.op != c0.op) return ncclInvalidUsage;
  }
  if (!hip_ready()) return ncclSystemError;
  // behind the work already enqueued on every rank's stream
  for (int r = 0; r < n; ++r) { p_setdev(comms[r]->device); if (p_sync(comms[r]->pending[0].stream)) return ncclUnhandledCudaError; }
  const size_t bytes = c0.count * dtype_bytes(c0.dtype);
  if (c0.kind == 0) {
    // the send buffers first (a rank's receive buffer may alias nothing of its own send buffer, but keep the order honest)
    std::vector<std::vector<char>> send(n, std::vector<char>(bytes));
    for (int r = 0; r < n; ++r) if (p_memcpy(send[r].data(), comms[r]->pending[0].send, bytes, 2 /* D2H */)) return ncclUnhandledCudaError;
    for (int r = 0; r < n; ++r)
      for (int q = 0; q < n; ++q)
        if (p_memcpy((char*)comms[r]->pending[0].recv + (size_t)q * bytes, send[q].data(), bytes, 1 /* H2D */)) return ncclUnhandledCudaError;
    ++g_calls[0];
  } else {
    if (c0.dtype != 5 || c0.op != 0) return ncclInvalidArgument;        // uint64 sums are all the library asks for
    std::vector<uint64_t> acc(c0.count, 0), tmp(c0.count);
    for (int r = 0; r < n; ++r) {
      if (p_memcpy(tmp.data(), comms[r]->pending[0].send, bytes, 2)) return ncclUnhandledCudaError;
      for (size_t i = 0; i < c0.count; ++i) acc[i] += tmp[i];
    }
    for (int r = 0; r < n; ++r) if (p_memcpy(comms[r]->pending[0].recv, acc.data(), bytes, 1)) return ncclUnhandledCudaError;
    ++g_calls[1];
  }
  for (int r = 0; r < n; ++r) comms[r]->pending.erase(comms[r]->pending.begin());
  return ncclSuccess;
}

ncclResult_t flush_locked() {
  // group the touched communicators by clique; a clique runs when all of its ranks have a call pending
  ncclResult_t rc = ncclSuccess;
  std::vector<FakeComm*> touched;
  touched.swap(g_touched);
  for (;;) {
    FakeComm* first = nullptr;
    for (FakeComm* c : touched) if (!c->pending.empty()) { first = c; break; }
    if (!first) break;
    Clique* cl = first->clique;
    std::vector<FakeComm*> comms(cl->n, nullptr);
    for (FakeComm* c : touched) if (c->clique == cl && !c->pending.empty()) comms[c->rank] = c;
    bool complete = true;
    for (FakeComm* c : comms) complete = complete && c != nullptr;
    if (!complete || cl->aborted) {
      // a rank is missing from the group: in real RCCL this call would wait for the peer for ever; here it is a usage error
      for (FakeComm* c : touched) if (c->clique == cl) c->pending.clear();
      rc = cl->aborted ? ncclInvalidUsage : ncclInvalidUsage;
      continue;
    }
    const ncclResult_t r = execute(comms);
    if (r != ncclSuccess) { for (FakeComm* c : comms) c->pending.clear(); rc = r; }
  }
  ++g_calls[2];
  return rc;
}

ncclResult_t enqueue(FakeComm* comm, const Call& c) {
  if (!comm || !comm->clique || comm->clique->aborted) return ncclInvalidArgument;
  std::lock_guard<std::mutex> g(g_mu);
  comm->pending.push_back(c);
  bool seen = false;
  for (FakeComm* t : g_touched) seen = seen || t == comm;
  if (!seen) g_touched.push_back(comm);
  if (g_depth == 0) return flush_locked();       // outside a group: a world of one executes at once, anything else is an error
  return ncclSuccess;
}
}  // namespace

extern "C" {
// what tells libfastmc that this is the tests' stand-in, not an RCCL build: only then may several ranks share a device
int fake_rccl_marker = 1;
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  static uint64_t serial = 0x0123456789abcdefULL;
  std::lock_guard<std::mutex> g(g_mu);
  for (int i = 0; i < 16; ++i) { serial = serial * 6364136223846793005ULL + 1442695040888963407ULL; memcpy(id->internal + 8 * i, &serial, 8); }
  memcpy(id->internal, "FAKERCCL", 8);
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  if (memcmp(id.internal, "FAKERCCL", 8) != 0) return ncclInvalidArgument;
  if (nranks != 1) return ncclInvalidUsage;          // one process: no peers to meet
  Clique* cl = new Clique();
  cl->n = 1; cl->live = 1;
  *comm = new FakeComm{cl, 0, 0, {}};
  return ncclSuccess;
}
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
  if (!comms || ndev < 1) return ncclInvalidArgument;
  Clique* cl = new Clique();
  cl->n = ndev; cl->live = ndev;
  for (int i = 0; i < ndev; ++i) comms[i] = new FakeComm{cl, i, devlist ? devlist[i] : i, {}};
  return ncclSuccess;
}
static ncclResult_t drop(ncclComm_t comm, bool abort) {
  if (!comm) return ncclInvalidArgument;
  std::lock_guard<std::mutex> g(g_mu);
  if (abort) { comm->clique->aborted = true; ++g_calls[3]; }
  for (size_t i = 0; i < g_touched.size(); ++i) if (g_touched[i] == comm) { g_touched.erase(g_touched.begin() + i); break; }
  if (--comm->clique->live == 0) delete comm->clique;
  delete comm;
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { return drop(comm, false); }
ncclResult_t ncclCommAbort(ncclComm_t comm) { return drop(comm, true); }
ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t_ stream) {
  if (!sendbuff || !recvbuff || sendcount == 0) return ncclInvalidArgument;
  return enqueue(comm, Call{0, sendbuff, recvbuff, sendcount, datatype, 0, stream});
}
ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t_ stream) {
  if (!sendbuff || !recvbuff || count == 0) return ncclInvalidArgument;
  return enqueue(comm, Call{1, sendbuff, recvbuff, count, datatype, op, stream});
}
ncclResult_t ncclGroupStart(void) {
  std::lock_guard<std::mutex> g(g_mu);
  ++g_depth;
  return ncclSuccess;
}
ncclResult_t ncclGroupEnd(void) {
  std::lock_guard<std::mutex> g(g_mu);
  if (g_depth <= 0) return ncclInvalidUsage;
  if (--g_depth > 0) return ncclSuccess;
  return flush_locked();
}
const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "unhandled cuda error (fake rccl: a HIP call failed)";
    case ncclSystemError: return "unhandled system error (fake rccl: no HIP runtime in the process)";
    case ncclInvalidArgument: return "invalid argument";
    case ncclInvalidUsage: return "invalid usage (fake rccl: ranks of a clique made different calls, a rank was missing from the group, or peers in other processes)";
    default: return "fake rccl error";
  }
}
// test hook: counts of what ran (all-gathers, all-reduces, groups, aborts)
void fake_rccl_counters(long out[4]) {
  std::lock_guard<std::mutex> g(g_mu);
  for (int i = 0; i < 4; ++i) out[i] = g_calls[i];
}
}
