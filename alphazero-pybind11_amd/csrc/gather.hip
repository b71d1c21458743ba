// The path's one exchange step, natively: finished self-play samples go to rank 0 over RCCL / xGMI -
// what the reference's hist_saver does with a queue inside one process (game_runner.py:729-747).  Self-play shards by game slot with
// no communication during the search; once per drain every rank contributes its new PlayHistory rows:
//   azmi_gather_counts   one ncclAllGather of the ranks' row counts (8 bytes each)
//   azmi_gather_rows     the rows themselves, UNPADDED: one grouped ncclSend / ncclRecv per array and sending rank (rank 0's own rows
//                        by a device copy); rank 0's extra memory is exactly the gathered rows (the torch.distributed version padded
//                        every rank to the largest count: world x n_max rows)
// librccl is loaded when the first communicator is made (dlopen): a one-GPU process never touches it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/azmi.h"
#include "engine_host.h"

namespace {
struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string err;
};
Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) { r.err = std::string("librccl not found: ") + (dlerror() ? dlerror() : ""); return; }
#define AZMI_SYM(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, #sym)); if (!r.field) { r.err = "librccl lacks " #sym; return; }
    AZMI_SYM(GetUniqueId, ncclGetUniqueId) AZMI_SYM(CommInitRank, ncclCommInitRank) AZMI_SYM(CommDestroy, ncclCommDestroy)
    AZMI_SYM(AllGather, ncclAllGather) AZMI_SYM(Send, ncclSend) AZMI_SYM(Recv, ncclRecv)
    AZMI_SYM(GroupStart, ncclGroupStart) AZMI_SYM(GroupEnd, ncclGroupEnd) AZMI_SYM(GetErrorString, ncclGetErrorString)
#undef AZMI_SYM
  });
  return &r;
}
#define AZMI_NCCL_TRY(expr) do { const ncclResult_t r_ = (expr); if (r_ != ncclSuccess) \
  return azmi_host_fail(AZMI_ERR_NO_DEVICE, "%s: %s", #expr, rccl()->GetErrorString(r_)); } while (0)
// inside a ncclGroupStart block: the group is closed before the error leaves (a return without ncclGroupEnd would leave this thread
// in group mode, and every later RCCL call of the process would be queued into a group nobody ends)
#define AZMI_NCCL_TRY_IN_GROUP(expr) do { const ncclResult_t r_ = (expr); if (r_ != ncclSuccess) { (void)rccl()->GroupEnd(); \
  return azmi_host_fail(AZMI_ERR_NO_DEVICE, "%s: %s", #expr, rccl()->GetErrorString(r_)); } } while (0)
}  // namespace

struct azmi_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  unsigned long long* counts_dev = nullptr;      // [world + 1]: the all-gather's receive buffer, then this rank's own count
};

extern "C" {

int azmi_comm_available(void) {
  Rccl* r = rccl();
  if (!r->lib || !r->err.empty()) return azmi_host_fail(AZMI_ERR_NO_DEVICE, "%s", r->err.c_str());
  return AZMI_OK;
}

int azmi_comm_unique_id(void* out128) {
  if (!out128) return azmi_host_fail(AZMI_ERR_INVALID, "null argument");
  Rccl* r = rccl();
  if (!r->lib || !r->err.empty()) return azmi_host_fail(AZMI_ERR_NO_DEVICE, "%s", r->err.c_str());
  static_assert(sizeof(ncclUniqueId) == 128, "the id travels as 128 bytes");
  ncclUniqueId id;
  AZMI_NCCL_TRY(r->GetUniqueId(&id));
  memcpy(out128, &id, sizeof(id));
  return AZMI_OK;
}

int azmi_comm_create(const void* id128, int rank, int world, int device, azmi_comm** out) {
  if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return azmi_host_fail(AZMI_ERR_INVALID, "azmi_comm_create: bad arguments");
  Rccl* r = rccl();
  if (!r->lib || !r->err.empty()) return azmi_host_fail(AZMI_ERR_NO_DEVICE, "%s", r->err.c_str());
  AZMI_HIP_TRY(hipSetDevice(device));
  auto c = new azmi_comm();
  c->rank = rank; c->world = world; c->device = device;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  const ncclResult_t rc = r->CommInitRank(&c->comm, world, id, rank);
  if (rc != ncclSuccess) { delete c; return azmi_host_fail(AZMI_ERR_NO_DEVICE, "ncclCommInitRank: %s", r->GetErrorString(rc)); }
  if (hipMalloc(reinterpret_cast<void**>(&c->counts_dev), sizeof(unsigned long long) * (world + 1)) != hipSuccess) {
    (void)r->CommDestroy(c->comm); delete c;
    return azmi_host_fail(AZMI_ERR_OOM, "hipMalloc(count buffer) failed");
  }
  *out = c;
  return AZMI_OK;
}

void azmi_comm_destroy(azmi_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->counts_dev) (void)hipFree(c->counts_dev);
  if (c->comm) (void)rccl()->CommDestroy(c->comm);
  delete c;
}

int azmi_gather_counts(azmi_comm* c, uint64_t n_local, uint64_t* out_counts, void* stream) {
  if (!c || !out_counts) return azmi_host_fail(AZMI_ERR_INVALID, "null argument");
  AZMI_HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned long long mine = n_local;
  AZMI_HIP_TRY(hipMemcpyAsync(c->counts_dev + c->world, &mine, sizeof(mine), hipMemcpyHostToDevice, st));
  AZMI_NCCL_TRY(rccl()->AllGather(c->counts_dev + c->world, c->counts_dev, 1, ncclUint64, c->comm, st));
  std::vector<unsigned long long> h(c->world);
  AZMI_HIP_TRY(hipMemcpyAsync(h.data(), c->counts_dev, sizeof(unsigned long long) * c->world, hipMemcpyDeviceToHost, st));
  AZMI_HIP_TRY(hipStreamSynchronize(st));
  for (int r = 0; r < c->world; ++r) out_counts[r] = h[r];
  return AZMI_OK;
}

int azmi_gather_rows(azmi_comm* c, const void* const* src, const uint64_t* row_bytes, uint32_t num_parts, const uint64_t* counts,
                     void* const* dst, void* stream) {
  if (!c || !src || !row_bytes || !counts || (c->rank == 0 && !dst)) return azmi_host_fail(AZMI_ERR_INVALID, "null argument");
  AZMI_HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  Rccl* r = rccl();
  const uint64_t n_local = counts[c->rank];
  for (uint32_t p = 0; p < num_parts; ++p) {
    const uint64_t rb = row_bytes[p];
    if (c->rank == 0) {
      uint8_t* d = static_cast<uint8_t*>(dst[p]);
      if (n_local) AZMI_HIP_TRY(hipMemcpyAsync(d, src[p], n_local * rb, hipMemcpyDeviceToDevice, st));      // my own rows: a device copy
      uint64_t off = n_local * rb;
      AZMI_NCCL_TRY(r->GroupStart());
      for (int q = 1; q < c->world; ++q) {
        if (counts[q]) AZMI_NCCL_TRY_IN_GROUP(r->Recv(d + off, counts[q] * rb, ncclUint8, q, c->comm, st));
        off += counts[q] * rb;
      }
      AZMI_NCCL_TRY(r->GroupEnd());
    } else if (n_local) {
      AZMI_NCCL_TRY(r->GroupStart());
      AZMI_NCCL_TRY_IN_GROUP(r->Send(src[p], n_local * rb, ncclUint8, 0, c->comm, st));
      AZMI_NCCL_TRY(r->GroupEnd());
    }
  }
  return AZMI_OK;
}

}  // extern "C"
