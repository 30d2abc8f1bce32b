// libbvhip: the collectives of the data-parallel step as plain C entry points over RCCL (xGMI), for hosts
// that are not Python - the list of SURVEY.md 8b: init / all_gather / reduce_scatter / all_reduce_bucket /
// destroy.  They are the calls big_vision_amd/dp.py issues through torch.distributed:
//   all_gather of the text embeddings            (_deprecated_contrastive.py:67-77,122)
//   reduce_scatter of their gradients            (the transpose JAX AD inserts: psum_scatter)
//   bucketed all_reduce(SUM) of the flat fp32 gradient buffer (pmean of grads, :343-344; sharding.py:83-101)
//   reduce_scatter / all_gather of the flat buffer for the "fsdp" placement (sharding.py:104-139)
// RCCL is bound at RUN time (dlopen + dlsym of librccl.so, no link-time dependency): a process that already
// holds an RCCL (PyTorch ships one) gets THAT instance - two RCCLs in one process is the failure mode this
// avoids - and libbvhip.so itself loads on hosts without RCCL.  The unique id is created by rank 0
// (bv_comm_unique_id) and carried to the other ranks by the host (file, environment, MPI ...).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string.h>

#include <mutex>

#include "bv_common.h"
#include "bvhip_internal.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclReduceScatter) ReduceScatter = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
};
Rccl g_rccl;
std::mutex g_mu;

int load_rccl() {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_rccl.handle) return BV_OK;
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);          // an RCCL that is already in the process first
    if (h) break;
  }
  for (int i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    bv_set_error("bv_comm: cannot load librccl.so (%s)", dlerror());
    return BV_ERR_UNSUPPORTED;
  }
#define BV_SYM(field, sym)                                                      \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, #sym));      \
  if (!g_rccl.field) {                                                          \
    bv_set_error("bv_comm: librccl.so has no %s", #sym);                        \
    return BV_ERR_UNSUPPORTED;                                                  \
  }
  BV_SYM(GetVersion, ncclGetVersion)
  BV_SYM(GetUniqueId, ncclGetUniqueId)
  BV_SYM(CommInitRank, ncclCommInitRank)
  BV_SYM(CommDestroy, ncclCommDestroy)
  BV_SYM(GetErrorString, ncclGetErrorString)
  BV_SYM(AllReduce, ncclAllReduce)
  BV_SYM(ReduceScatter, ncclReduceScatter)
  BV_SYM(AllGather, ncclAllGather)
#undef BV_SYM
  g_rccl.handle = h;
  return BV_OK;
}

int check(ncclResult_t r, const char* what) {
  if (r == ncclSuccess) return BV_OK;
  bv_set_error("%s: RCCL error %d (%s)", what, (int)r, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
  return BV_ERR_HIP;
}

bool dtype_of(int dtype, ncclDataType_t* out, size_t* size) {
  if (dtype == BV_COMM_F32) { *out = ncclFloat32; *size = 4; return true; }
  if (dtype == BV_COMM_BF16) { *out = ncclBfloat16; *size = 2; return true; }
  if (dtype == BV_COMM_F64) { *out = ncclFloat64; *size = 8; return true; }
  return false;
}

}  // namespace

static_assert(sizeof(ncclUniqueId) == BV_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");

extern "C" int bv_comm_version(int* version) {
  int rc = load_rccl();
  if (rc) return rc;
  BV_REQUIRE(version != nullptr, "bv_comm_version: null output");
  return check(g_rccl.GetVersion(version), "ncclGetVersion");
}

extern "C" int bv_comm_unique_id(void* id_out) {
  int rc = load_rccl();
  if (rc) return rc;
  BV_REQUIRE(id_out != nullptr, "bv_comm_unique_id: null output");
  ncclUniqueId id;
  rc = check(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
  if (rc) return rc;
  memcpy(id_out, &id, sizeof(id));
  return BV_OK;
}

extern "C" int bv_comm_init(const void* id, int rank, int world, void** comm_out) {
  int rc = load_rccl();
  if (rc) return rc;
  BV_REQUIRE(id && comm_out && world >= 1 && rank >= 0 && rank < world, "bv_comm_init: bad arguments (rank %d of %d)", rank, world);
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclComm_t c = nullptr;
  rc = check(g_rccl.CommInitRank(&c, world, uid, rank), "ncclCommInitRank");   // the calling thread's current HIP device
  if (rc) return rc;
  *comm_out = c;
  return BV_OK;
}

extern "C" int bv_comm_destroy(void* comm) {
  if (!comm) return BV_OK;
  int rc = load_rccl();
  if (rc) return rc;
  return check(g_rccl.CommDestroy((ncclComm_t)comm), "ncclCommDestroy");
}

extern "C" int bv_comm_all_gather(void* comm, const void* send, void* recv, long count_per_rank, int dtype,
                                  void* stream) {
  int rc = load_rccl();
  if (rc) return rc;
  ncclDataType_t dt; size_t sz;
  BV_REQUIRE(comm && send && recv && count_per_rank > 0 && dtype_of(dtype, &dt, &sz), "bv_comm_all_gather: bad arguments");
  return check(g_rccl.AllGather(send, recv, (size_t)count_per_rank, dt, (ncclComm_t)comm, (hipStream_t)stream), "ncclAllGather");
}

extern "C" int bv_comm_reduce_scatter(void* comm, const void* send, void* recv, long count_per_rank, int dtype,
                                      void* stream) {
  int rc = load_rccl();
  if (rc) return rc;
  ncclDataType_t dt; size_t sz;
  BV_REQUIRE(comm && send && recv && count_per_rank > 0 && dtype_of(dtype, &dt, &sz), "bv_comm_reduce_scatter: bad arguments");
  return check(g_rccl.ReduceScatter(send, recv, (size_t)count_per_rank, dt, ncclSum, (ncclComm_t)comm, (hipStream_t)stream),
               "ncclReduceScatter");
}

// In-place SUM over the ranks of buf[0 .. count) in buckets of bucket_elems elements (0 = one message): xGMI is
// point-to-point, RCCL picks ring / tree / direct per message; large buckets amortise the launch, per-block
// ranges (28 MB for B/16) are what overlaps the backward (dp.GradSync).
extern "C" int bv_comm_all_reduce_bucket(void* comm, void* buf, long count, long bucket_elems, int dtype, void* stream) {
  int rc = load_rccl();
  if (rc) return rc;
  ncclDataType_t dt; size_t sz;
  BV_REQUIRE(comm && buf && count > 0 && bucket_elems >= 0 && dtype_of(dtype, &dt, &sz), "bv_comm_all_reduce_bucket: bad arguments");
  const long step = bucket_elems > 0 ? bucket_elems : count;
  for (long off = 0; off < count; off += step) {
    const long n = count - off < step ? count - off : step;
    char* p = (char*)buf + (size_t)off * sz;
    rc = check(g_rccl.AllReduce(p, p, (size_t)n, dt, ncclSum, (ncclComm_t)comm, (hipStream_t)stream), "ncclAllReduce");
    if (rc) return rc;
  }
  return BV_OK;
}
