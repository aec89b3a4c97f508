// siga_amd/csrc/sigax_comm.cpp -- the one exchange step of the multi-GPU path behind the C-ABI: a
// variable-length gather of the fixed-size edge records to one rank's GPU over RCCL -- ncclAllGather of the per-rank
// counts, then grouped ncclSend / ncclRecv of the records (xGMI between the MI355X of one node) -- for callers that run one
// process per GPU and want the records of a step on ONE device (bench.py --gpus N; a service that post-processes on GPU 0).
// Replaces the serial hits -> ASQG pass of src/overlap_builder.cpp:466-483 as far as it concerns moving records between
// devices; the rank order of the gathered records is the read order, so the ED order of a one-GPU run is preserved.
//
// RCCL is bound at run time (dlopen): libsigax.so carries no link-time dependency on it, the one-GPU paths never touch it,
// and a process that already holds an RCCL (PyTorch ships its own librccl.so) gets THAT one instead of a second copy.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/sigax.h"

int sigax_fail(int code, const char* fmt, ...);  // sigax_api.cpp: sets the thread-local error text, returns code

namespace {
struct Rccl {
  void* h = nullptr;
  bool tried = false, ok = false;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string why;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

bool rccl_load() {
  std::lock_guard<std::mutex> lock(g_rccl_mu);
  if (g_rccl.tried) return g_rccl.ok;
  g_rccl.tried = true;
  // an RCCL this process has already (PyTorch's), else the ROCm one
  void* h = nullptr;
  if (dlsym(RTLD_DEFAULT, "ncclCommInitRank") != nullptr) h = RTLD_DEFAULT;
  const char* names[] = {"librccl.so.1", "librccl.so"};
  for (int pass = 0; pass < 2 && !h; ++pass)
    for (const char* n : names) {
      h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (h) break;
    }
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    g_rccl.why = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : "?");
    return false;
  }
  g_rccl.h = h;
  auto sym = [&](const char* n) { return dlsym(h, n); };
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
  g_rccl.AllGather = (decltype(g_rccl.AllGather))sym("ncclAllGather");
  g_rccl.Send = (decltype(g_rccl.Send))sym("ncclSend");
  g_rccl.Recv = (decltype(g_rccl.Recv))sym("ncclRecv");
  g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
  g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
  g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllGather && g_rccl.Send && g_rccl.Recv &&
              g_rccl.GroupStart && g_rccl.GroupEnd && g_rccl.GetErrorString;
  if (!g_rccl.ok) g_rccl.why = "librccl.so lacks one of the nccl* entry points";
  return g_rccl.ok;
}
}  // namespace

struct sigax_comm {
  int device, rank, world;
  ncclComm_t comm;
  unsigned long long* d_counts;  // [world + 1]: this rank's count at [world], everyone's at [0, world)
  unsigned long long* h_counts;  // pinned mirror
};

#define RCCL_TRY(expr)                                                                                      \
  do {                                                                                                      \
    ncclResult_t r_ = (expr);                                                                               \
    if (r_ != ncclSuccess) return sigax_fail(SIGAX_E_DEVICE, "%s: %s", #expr, g_rccl.GetErrorString(r_)); \
  } while (0)
#define HIP_TRY2(expr)                                                                                  \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess) return sigax_fail(SIGAX_E_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

static_assert(SIGAX_COMM_ID_BYTES == sizeof(ncclUniqueId), "sigax_comm ids are ncclUniqueId");

extern "C" int sigax_comm_unique_id(uint8_t id[SIGAX_COMM_ID_BYTES]) {
  if (!id) return sigax_fail(SIGAX_E_ARG, "NULL argument");
  if (!rccl_load()) return sigax_fail(SIGAX_E_DEVICE, "RCCL unavailable: %s", g_rccl.why.c_str());
  ncclUniqueId u;
  RCCL_TRY(g_rccl.GetUniqueId(&u));
  memcpy(id, &u, sizeof(u));
  return SIGAX_OK;
}

extern "C" int sigax_comm_create(int device, int rank, int world, const uint8_t id[SIGAX_COMM_ID_BYTES], sigax_comm** out) {
  if (!out || !id || world < 1 || rank < 0 || rank >= world) return sigax_fail(SIGAX_E_ARG, "bad argument");
  *out = nullptr;
  if (!rccl_load()) return sigax_fail(SIGAX_E_DEVICE, "RCCL unavailable: %s", g_rccl.why.c_str());
  HIP_TRY2(hipSetDevice(device));
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  sigax_comm* c = new sigax_comm();
  c->device = device;
  c->rank = rank;
  c->world = world;
  c->comm = nullptr;
  c->d_counts = nullptr;
  c->h_counts = nullptr;
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, u, rank);
  if (r != ncclSuccess) {
    delete c;
    return sigax_fail(SIGAX_E_DEVICE, "ncclCommInitRank(rank %d of %d on device %d): %s", rank, world, device, g_rccl.GetErrorString(r));
  }
  hipError_t e = hipMalloc((void**)&c->d_counts, ((size_t)world + 1) * 8);
  if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_counts, ((size_t)world + 1) * 8, hipHostMallocDefault);
  if (e != hipSuccess) {
    sigax_comm_destroy(c);
    return sigax_fail(SIGAX_E_DEVICE, "buffers of the communicator: %s", hipGetErrorString(e));
  }
  *out = c;
  return SIGAX_OK;
}

extern "C" void sigax_comm_destroy(sigax_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->comm) (void)g_rccl.CommDestroy(c->comm);
  if (c->d_counts) (void)hipFree(c->d_counts);
  if (c->h_counts) (void)hipHostFree(c->h_counts);
  delete c;
}

// Every rank calls sigax_gather_counts with the number of records it holds: counts[0 .. world) (host) holds every rank's
// count on return (one host wait on `stream`: the root sizes its buffer and its receives from them).  Then every rank calls
// sigax_gather_edges with those counts: the record transfer is enqueued on `stream`; on `root`, d_out (room for the sum of
// the counts) receives rank 0's records, then rank 1's, ...; other ranks pass d_out = NULL.
extern "C" int sigax_gather_counts(sigax_comm* c, uint64_t n_local, uint64_t* counts, void* stream) {
  if (!c || !counts) return sigax_fail(SIGAX_E_ARG, "bad argument");
  HIP_TRY2(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  const int W = c->world;
  c->h_counts[W] = n_local;
  HIP_TRY2(hipMemcpyAsync(c->d_counts + W, c->h_counts + W, 8, hipMemcpyHostToDevice, st));
  RCCL_TRY(g_rccl.AllGather(c->d_counts + W, c->d_counts, 1, ncclUint64, c->comm, st));
  HIP_TRY2(hipMemcpyAsync(c->h_counts, c->d_counts, (size_t)W * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY2(hipStreamSynchronize(st));
  for (int r = 0; r < W; ++r) counts[r] = c->h_counts[r];
  return SIGAX_OK;
}

extern "C" int sigax_gather_edges(sigax_comm* c, const sigax_edge* d_local, const uint64_t* counts, int root, sigax_edge* d_out, void* stream) {
  if (!c || !counts || root < 0 || root >= c->world) return sigax_fail(SIGAX_E_ARG, "bad argument");
  const int W = c->world;
  const uint64_t n_local = counts[c->rank];
  uint64_t total = 0;
  for (int r = 0; r < W; ++r) total += counts[r];
  // (argument errors are the same on every rank or concern this rank alone before anything is posted)
  if ((n_local && !d_local) || (c->rank == root && total && !d_out)) return sigax_fail(SIGAX_E_ARG, "NULL record buffer");
  HIP_TRY2(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  // records travel as u32 x 4 (16-byte sigax_edge); the root's own share is a device copy
  RCCL_TRY(g_rccl.GroupStart());
  ncclResult_t r0 = ncclSuccess;
  if (c->rank == root) {
    uint64_t at = 0;
    for (int r = 0; r < W && r0 == ncclSuccess; ++r) {
      if (r != root && counts[r]) r0 = g_rccl.Recv(d_out + at, (size_t)counts[r] * 4, ncclUint32, r, c->comm, st);
      at += counts[r];
    }
  } else if (n_local) {
    r0 = g_rccl.Send(d_local, (size_t)n_local * 4, ncclUint32, root, c->comm, st);
  }
  ncclResult_t r1 = g_rccl.GroupEnd();
  if (r0 != ncclSuccess) return sigax_fail(SIGAX_E_DEVICE, "ncclSend/ncclRecv: %s", g_rccl.GetErrorString(r0));
  if (r1 != ncclSuccess) return sigax_fail(SIGAX_E_DEVICE, "ncclGroupEnd: %s", g_rccl.GetErrorString(r1));
  if (c->rank == root && n_local) {
    uint64_t at = 0;
    for (int r = 0; r < root; ++r) at += counts[r];
    HIP_TRY2(hipMemcpyAsync(d_out + at, d_local, (size_t)n_local * sizeof(sigax_edge), hipMemcpyDeviceToDevice, st));
  }
  return SIGAX_OK;
}
