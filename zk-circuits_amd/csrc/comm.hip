// The one exchange of the sharded path: an all-gather of fixed-size proof blobs between the ranks of a node, over RCCL / xGMI, behind the
// C-ABI (include/qpgpu.h qpgpu_comm_*), so that a Rust host reaches it like qpgpu_prove — no Python, no torch (the
// reference's counterpart is the rayon join at wormhole/aggregator/src/circuits/tree.rs:92-103, a shared-memory hand-over of proofs).
//
// librccl is opened at run time (dlopen), from the directory of the HIP runtime this library is linked against: a process that also
// holds PyTorch has a second HIP runtime with its own RCCL, and streams / device pointers must stay with the runtime that made them.
// Blobs are ≤ 150 KB each (76 MB for 512 leaf proofs): latency-bound, one ncclAllGather per call, staged through device buffers the
// communicator keeps (host pointers in and out, as everywhere at this boundary).
#include "../../include/qpgpu.h"
#include "kernels.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#include <string>

namespace qp { extern thread_local std::string g_last_error; int set_err(int code, const std::string& m); }
using namespace qp;

namespace {
struct Rccl {
  void* lib = nullptr; std::string path, err;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  int (*GetVersion)(int*) = nullptr;
};
Rccl& rccl() {
  static Rccl r; static std::once_flag once;
  std::call_once(once, [] {
    std::string dir;
    if (const char* e = getenv("QPGPU_RCCL_LIB")) { r.path = e; }
    else {
      Dl_info info;                                     // the directory of libamdhip64.so as linked into this library
      if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname) { dir = info.dli_fname; const size_t k = dir.rfind('/'); dir = k == std::string::npos ? "" : dir.substr(0, k + 1); }
      r.path = dir + "librccl.so.1";
    }
    r.lib = dlopen(r.path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!r.lib && !getenv("QPGPU_RCCL_LIB")) for (const std::string& alt : {dir + "librccl.so", std::string("librccl.so.1")}) {
      if (r.lib) break;
      r.path = alt; r.lib = dlopen(alt.c_str(), RTLD_NOW | RTLD_LOCAL);
    }
    if (!r.lib) { const char* e = dlerror(); r.err = std::string("cannot open ") + r.path + ": " + (e ? e : "?"); return; }
    auto sym = [&](const char* n) { void* p = dlsym(r.lib, n); if (!p && r.err.empty()) r.err = std::string("librccl lacks ") + n; return p; };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    r.GetVersion = (decltype(r.GetVersion))dlsym(r.lib, "ncclGetVersion");
  });
  return r;
}
int rccl_ready() { Rccl& r = rccl(); if (!r.err.empty()) return set_err(QPGPU_ERR_NO_DEVICE, "RCCL unavailable: " + r.err); return 0; }
template <class F> int guard(F&& f) {
  try { return f(); }
  catch (const std::bad_alloc&) { return set_err(QPGPU_ERR_OOM, "host memory allocation failed"); }
  catch (...) { return set_err(QPGPU_ERR_HIP, "internal error"); }
}
#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return set_err(e_ == hipErrorOutOfMemory ? QPGPU_ERR_OOM : QPGPU_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
#define NCCL_TRY(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return set_err(QPGPU_ERR_HIP, std::string(#x) + ": " + rccl().GetErrorString(r_)); } while (0)
}

struct qpgpu_comm {
  int device = 0, rank = 0, world = 1;
  ncclComm_t comm = nullptr; hipStream_t stream = nullptr;
  char* d_send = nullptr; char* d_recv = nullptr; size_t cap_send = 0, cap_recv = 0;
  std::mutex mu;
};

extern "C" {

int qpgpu_comm_unique_id(uint8_t id[QPGPU_COMM_ID_BYTES]) { return guard([&]() -> int {
  static_assert(QPGPU_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id is ncclUniqueId");
  if (!id) return set_err(QPGPU_ERR_ARG, "null argument");
  if (int rc = rccl_ready()) return rc;
  ncclUniqueId u; NCCL_TRY(rccl().GetUniqueId(&u));
  memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES); return 0;
}); }

int qpgpu_comm_create(qpgpu_ctx* ctx, int rank, int world, const uint8_t id[QPGPU_COMM_ID_BYTES], qpgpu_comm** out) { return guard([&]() -> int {
  if (!ctx || !id || !out) return set_err(QPGPU_ERR_ARG, "null argument");
  if (world < 1 || rank < 0 || rank >= world) return set_err(QPGPU_ERR_ARG, "rank " + std::to_string(rank) + " outside a world of " + std::to_string(world));
  if (int rc = rccl_ready()) return rc;
  HIP_TRY(hipSetDevice(ctx->device));
  qpgpu_comm* c = new qpgpu_comm; c->device = ctx->device; c->rank = rank; c->world = world;
  struct Undo { qpgpu_comm* c; ~Undo() { if (c) qpgpu_comm_destroy(c); } } undo{c};
  HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  ncclUniqueId u; memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
  NCCL_TRY(rccl().CommInitRank(&c->comm, world, u, rank));          // blocks until all `world` ranks have called it with the same id
  undo.c = nullptr; *out = c; return 0;
}); }

void qpgpu_comm_destroy(qpgpu_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm && rccl().CommDestroy) (void)rccl().CommDestroy(c->comm);
  if (c->d_send) (void)hipFree(c->d_send);
  if (c->d_recv) (void)hipFree(c->d_recv);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int qpgpu_comm_rank(const qpgpu_comm* c) { return c ? c->rank : -1; }
int qpgpu_comm_world(const qpgpu_comm* c) { return c ? c->world : 0; }

int qpgpu_gather_proofs(qpgpu_comm* c, const uint8_t* local, size_t count, size_t proof_size, uint8_t* out) { return guard([&]() -> int {
  if (!c || !out || (!local && count)) return set_err(QPGPU_ERR_ARG, "null argument");
  const size_t bytes = count * proof_size;
  if (bytes == 0) return 0;
  if (proof_size && bytes / proof_size != count) return set_err(QPGPU_ERR_ARG, "size overflow");
  std::lock_guard<std::mutex> lk(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  const size_t pad = (bytes + 255) & ~size_t(255);                    // every rank's block starts 256-byte aligned in the staging buffer
  if (c->cap_send < pad) { if (c->d_send) (void)hipFree(c->d_send); c->d_send = nullptr; c->cap_send = 0; HIP_TRY(hipMalloc((void**)&c->d_send, pad)); c->cap_send = pad; }
  if (c->cap_recv < pad * (size_t)c->world) { if (c->d_recv) (void)hipFree(c->d_recv); c->d_recv = nullptr; c->cap_recv = 0; HIP_TRY(hipMalloc((void**)&c->d_recv, pad * (size_t)c->world)); c->cap_recv = pad * (size_t)c->world; }
  HIP_TRY(hipMemcpyAsync(c->d_send, local, bytes, hipMemcpyHostToDevice, c->stream));
  NCCL_TRY(rccl().AllGather(c->d_send, c->d_recv, pad, ncclUint8, c->comm, c->stream));
  HIP_TRY(hipMemcpy2DAsync(out, bytes, c->d_recv, pad, bytes, (size_t)c->world, hipMemcpyDeviceToHost, c->stream));   // drops the padding
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}); }

int qpgpu_comm_rccl_version(void) { int v = 0; if (rccl_ready() == 0 && rccl().GetVersion) (void)rccl().GetVersion(&v); return v; }

} // extern "C"
