// Gradient exchange of the data-parallel step over RCCL (xGMI inside the node), as plain C-ABI entry points: what the reference gets from
// torch.nn.parallel.DistributedDataParallel (func/train.py:771-778) and torch.distributed.init_process_group('nccl') (common/utils.py:145-148).
// One process per GPU; a communicator is created from a 128-byte unique id that rank 0 draws and the host ships to the other ranks by any side
// channel (file, TCP store, MPI).  Every call enqueues on the caller's stream and returns; nothing is allocated besides the communicator itself.
//
// RCCL is bound at FIRST USE (dlopen of librccl.so.1), not at link time: a process that never exchanges gradients -- the 1-GPU path, the CPU-side
// symbol tests -- does not load it, and inside a PyTorch process the soname resolves to the librccl.so.1 torch has already loaded (one RCCL per process).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <mutex>
#include "../../include/avt_hip.h"

void avt_set_error(const char* fmt, ...);

namespace {
struct Rccl {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*);
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*CommCount)(const ncclComm_t, int*);
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*);
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  const char* (*GetErrorString)(ncclResult_t);
};

const Rccl* rccl() {
  static std::mutex mu;
  static Rccl r{};
  static int state = 0;          // 0 = not tried, 1 = bound, -1 = failed
  std::lock_guard<std::mutex> lock(mu);
  if (state == 0) {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    state = -1;
    if (!h) { avt_set_error("avt_comm: cannot load librccl.so.1: %s", dlerror()); return nullptr; }
#define AVT_BIND(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name)); if (!r.field) { avt_set_error("avt_comm: librccl has no %s", name); return nullptr; }
    AVT_BIND(GetUniqueId, "ncclGetUniqueId") AVT_BIND(CommInitRank, "ncclCommInitRank") AVT_BIND(CommDestroy, "ncclCommDestroy")
    AVT_BIND(CommCount, "ncclCommCount") AVT_BIND(CommUserRank, "ncclCommUserRank") AVT_BIND(AllReduce, "ncclAllReduce")
    AVT_BIND(ReduceScatter, "ncclReduceScatter") AVT_BIND(AllGather, "ncclAllGather") AVT_BIND(Broadcast, "ncclBroadcast")
    AVT_BIND(GetErrorString, "ncclGetErrorString")
#undef AVT_BIND
    state = 1;
  }
  if (state != 1) { avt_set_error("avt_comm: librccl.so.1 could not be bound earlier in this process"); return nullptr; }
  return &r;
}

struct Comm { ncclComm_t c; int nranks, rank, device; };

#define AVT_CHECK(cond, ...) do { if (!(cond)) { avt_set_error(__VA_ARGS__); return -1; } } while (0)
#define AVT_NCCL(call, what) do { ncclResult_t rc_ = (call); if (rc_ != ncclSuccess) { avt_set_error("%s: RCCL error %d (%s)", what, (int)rc_, R->GetErrorString(rc_)); return (int)rc_; } } while (0)

bool dtype_of(int dtype, ncclDataType_t& t, size_t& bytes) {
  if (dtype == 0) { t = ncclFloat32; bytes = 4; return true; }
  if (dtype == 1) { t = ncclBfloat16; bytes = 2; return true; }
  return false;
}
}  // namespace

extern "C" int avt_comm_unique_id(void* id128) {
  AVT_CHECK(id128, "avt_comm_unique_id: null argument");
  const Rccl* R = rccl();
  if (!R) return -1;
  static_assert(sizeof(ncclUniqueId) == AVT_COMM_ID_BYTES, "AVT_COMM_ID_BYTES");
  AVT_NCCL(R->GetUniqueId(reinterpret_cast<ncclUniqueId*>(id128)), "avt_comm_unique_id");
  return 0;
}

extern "C" int avt_comm_init_rank(void** comm, int nranks, int rank, int device, const void* id128) {
  AVT_CHECK(comm && id128, "avt_comm_init_rank: null argument");
  AVT_CHECK(nranks >= 1 && rank >= 0 && rank < nranks, "avt_comm_init_rank: bad rank %d of %d", rank, nranks);
  int ndev = 0;
  AVT_CHECK(hipGetDeviceCount(&ndev) == hipSuccess && device >= 0 && device < ndev, "avt_comm_init_rank: device %d of %d", device, ndev);
  const Rccl* R = rccl();
  if (!R) return -1;
  int prev = 0;
  (void)hipGetDevice(&prev);
  AVT_CHECK(hipSetDevice(device) == hipSuccess, "avt_comm_init_rank: hipSetDevice(%d) failed", device);
  ncclUniqueId id;
  __builtin_memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  const ncclResult_t rc = R->CommInitRank(&c, nranks, id, rank);
  (void)hipSetDevice(prev);
  if (rc != ncclSuccess) { avt_set_error("avt_comm_init_rank: RCCL error %d (%s)", (int)rc, R->GetErrorString(rc)); return (int)rc; }
  *comm = new Comm{c, nranks, rank, device};
  return 0;
}

extern "C" int avt_comm_destroy(void* comm) {
  if (!comm) return 0;
  const Rccl* R = rccl();
  if (!R) return -1;
  Comm* cm = static_cast<Comm*>(comm);
  const ncclResult_t rc = R->CommDestroy(cm->c);
  delete cm;
  if (rc != ncclSuccess) { avt_set_error("avt_comm_destroy: RCCL error %d (%s)", (int)rc, R->GetErrorString(rc)); return (int)rc; }
  return 0;
}

extern "C" int avt_comm_size(void* comm, int* nranks, int* rank) {
  AVT_CHECK(comm, "avt_comm_size: null communicator");
  const Rccl* R = rccl();
  if (!R) return -1;
  Comm* cm = static_cast<Comm*>(comm);
  int n = 0, r = 0;                                 // asked of RCCL, not of the handle: what the communicator really connected
  AVT_NCCL(R->CommCount(cm->c, &n), "avt_comm_size");
  AVT_NCCL(R->CommUserRank(cm->c, &r), "avt_comm_size");
  if (nranks) *nranks = n;
  if (rank) *rank = r;
  return 0;
}

extern "C" int avt_allreduce_bucket(void* comm, void* buf, size_t n, int dtype, void* stream) {
  AVT_CHECK(comm && buf && n > 0, "avt_allreduce_bucket: null argument");
  ncclDataType_t t; size_t eb;
  AVT_CHECK(dtype_of(dtype, t, eb), "avt_allreduce_bucket: dtype must be 0 (fp32) or 1 (bf16), got %d", dtype);
  const Rccl* R = rccl();
  if (!R) return -1;
  AVT_NCCL(R->AllReduce(buf, buf, n, t, ncclSum, static_cast<Comm*>(comm)->c, (hipStream_t)stream), "avt_allreduce_bucket");
  return 0;
}

extern "C" int avt_reduce_scatter_bucket(void* comm, void* buf, size_t n, int dtype, void* stream) {
  AVT_CHECK(comm && buf && n > 0, "avt_reduce_scatter_bucket: null argument");
  ncclDataType_t t; size_t eb;
  AVT_CHECK(dtype_of(dtype, t, eb), "avt_reduce_scatter_bucket: dtype must be 0 (fp32) or 1 (bf16), got %d", dtype);
  Comm* cm = static_cast<Comm*>(comm);
  AVT_CHECK(n % (size_t)cm->nranks == 0 && ((n / cm->nranks) * eb) % 16 == 0, "avt_reduce_scatter_bucket: %zu elements do not split into %d 16-byte aligned shards", n, cm->nranks);
  const Rccl* R = rccl();
  if (!R) return -1;
  const size_t shard = n / (size_t)cm->nranks;
  AVT_NCCL(R->ReduceScatter(buf, static_cast<char*>(buf) + (size_t)cm->rank * shard * eb, shard, t, ncclSum, cm->c, (hipStream_t)stream), "avt_reduce_scatter_bucket");
  return 0;
}

extern "C" int avt_allgather_bucket(void* comm, void* buf, size_t n, int dtype, void* stream) {
  AVT_CHECK(comm && buf && n > 0, "avt_allgather_bucket: null argument");
  ncclDataType_t t; size_t eb;
  AVT_CHECK(dtype_of(dtype, t, eb), "avt_allgather_bucket: dtype must be 0 (fp32) or 1 (bf16), got %d", dtype);
  Comm* cm = static_cast<Comm*>(comm);
  AVT_CHECK(n % (size_t)cm->nranks == 0 && ((n / cm->nranks) * eb) % 16 == 0, "avt_allgather_bucket: %zu elements do not split into %d 16-byte aligned shards", n, cm->nranks);
  const Rccl* R = rccl();
  if (!R) return -1;
  const size_t shard = n / (size_t)cm->nranks;
  AVT_NCCL(R->AllGather(static_cast<char*>(buf) + (size_t)cm->rank * shard * eb, buf, shard, t, cm->c, (hipStream_t)stream), "avt_allgather_bucket");
  return 0;
}

extern "C" int avt_broadcast_bucket(void* comm, void* buf, size_t n, int dtype, int root, void* stream) {
  AVT_CHECK(comm && buf && n > 0, "avt_broadcast_bucket: null argument");
  ncclDataType_t t; size_t eb;
  AVT_CHECK(dtype_of(dtype, t, eb), "avt_broadcast_bucket: dtype must be 0 (fp32) or 1 (bf16), got %d", dtype);
  Comm* cm = static_cast<Comm*>(comm);
  AVT_CHECK(root >= 0 && root < cm->nranks, "avt_broadcast_bucket: root %d of %d ranks", root, cm->nranks);
  const Rccl* R = rccl();
  if (!R) return -1;
  AVT_NCCL(R->Broadcast(buf, buf, n, t, root, cm->c, (hipStream_t)stream), "avt_broadcast_bucket");
  return 0;
}
