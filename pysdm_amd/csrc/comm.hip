// comm.hip -- what crosses processes, issued by the library itself: RCCL (over xGMI between the
// GPUs of a node) all-reduces on the context's stream, where a sharded step would otherwise call the
// host's `exchange` callback (sdm_hip.h).  The host keeps the bootstrap only: one process asks for
// a unique id (sdm_comm_unique_id), hands it to the others by whatever it has (torch.distributed's
// store, MPI, a file), every process calls sdm_comm_init - from then on no host code runs inside
// the sub-step loop.
// RCCL is resolved at run time (dlopen of the library the process has loaded already - under
// PyTorch that is torch's own librccl.so - else the system's): libsdm_hip.so does not link against
// it, loads without it, and the C-ABI example stays free of it.
#include <dlfcn.h>

#include "common.h"

namespace {
struct UniqueId { char bytes[SDM_COMM_ID_BYTES]; };  // == ncclUniqueId (rccl.h:40-43)
enum { RCCL_SUM = 0, RCCL_MIN = 3, RCCL_INT64 = 4, RCCL_FLOAT64 = 8 };  // rccl.h:448-467

struct Rccl {
  void *lib = nullptr;
  int (*GetUniqueId)(UniqueId *) = nullptr;
  int (*CommInitRank)(void **, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;

int rccl_load() {
  if (g_rccl.lib) return SDM_OK;
  const char *names[] = {"librccl.so.1", "librccl.so"};
  void *lib = nullptr;
  for (const char *name : names)  // the copy this process runs already, if any
    if ((lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD))) break;
  for (int k = 0; !lib && k < 2; ++k) lib = dlopen(names[k], RTLD_NOW | RTLD_LOCAL);
  if (!lib) {
    sdm_set_error("RCCL is not available: %s", dlerror());
    return SDM_E_HIP;
  }
  Rccl r;
  r.lib = lib;
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(lib, "ncclCommInitRank");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(lib, "ncclCommDestroy");
  r.AllReduce = (decltype(r.AllReduce))dlsym(lib, "ncclAllReduce");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(lib, "ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce) {
    sdm_set_error("the RCCL library lacks an expected symbol");
    return SDM_E_HIP;
  }
  g_rccl = r;
  return SDM_OK;
}

int rccl_try(int code, const char *what) {
  if (code == 0) return SDM_OK;
  sdm_set_error("%s failed: %s", what,
                g_rccl.GetErrorString ? g_rccl.GetErrorString(code) : "RCCL error");
  return SDM_E_HIP;
}
}  // namespace

extern "C" int sdm_comm_unique_id(uint8_t *id) {
  ARG_TRY(id != nullptr);
  const int rc = rccl_load();
  if (rc) return rc;
  UniqueId u;
  const int e = rccl_try(g_rccl.GetUniqueId(&u), "ncclGetUniqueId");
  if (e) return e;
  memcpy(id, u.bytes, SDM_COMM_ID_BYTES);
  return SDM_OK;
}

extern "C" int sdm_comm_init(sdm_ctx *ctx, const uint8_t *id, int rank, int world) {
  ARG_TRY(ctx && id && world >= 1 && rank >= 0 && rank < world);
  int rc = rccl_load();
  if (rc) return rc;
  if (ctx->comm && ctx->comm_owned) (void)g_rccl.CommDestroy(ctx->comm);
  ctx->comm = nullptr;
  HIP_TRY(hipSetDevice(ctx->device));
  UniqueId u;
  memcpy(u.bytes, id, SDM_COMM_ID_BYTES);
  void *comm = nullptr;
  rc = rccl_try(g_rccl.CommInitRank(&comm, world, u, rank), "ncclCommInitRank");
  if (rc) return rc;
  ctx->comm = comm;
  ctx->comm_owned = true;
  ctx->comm_rank = rank;
  ctx->comm_world = world;
  return SDM_OK;
}

extern "C" int sdm_shard_set_comm(sdm_ctx *ctx, void *rccl_comm) {
  ARG_TRY(ctx != nullptr);
  if (rccl_comm) {
    const int rc = rccl_load();
    if (rc) return rc;
  }
  if (ctx->comm && ctx->comm_owned && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(ctx->comm);
  ctx->comm = rccl_comm;
  ctx->comm_owned = false;
  ctx->comm_rank = ctx->comm_world = -1;  // (the caller's communicator: not ours to ask)
  return SDM_OK;
}

extern "C" int sdm_comm_destroy(sdm_ctx *ctx) {
  ARG_TRY(ctx != nullptr);
  if (ctx->comm && ctx->comm_owned && g_rccl.CommDestroy) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)g_rccl.CommDestroy(ctx->comm);
  }
  ctx->comm = nullptr;
  ctx->comm_owned = false;
  return SDM_OK;
}

// one exchange of a sharded step: in-place reduction over the processes, ordered on ctx->stream
// behind the kernels that filled the buffer and ahead of those that read it
int sdm_exchange(sdm_ctx *ctx, sdm_exchange_fn callback, void *user, int what, void *buffer,
                 int64_t count) {
  ++ctx->stats[SDM_STAT_EXCHANGES];
  ctx->stats[SDM_STAT_EXCHANGE_BYTES] += 8 * count;
  if (ctx->comm) {
    const int type = what == SDM_XCHG_SUM_I64 ? RCCL_INT64 : RCCL_FLOAT64;
    const int op = what == SDM_XCHG_MIN_F64 ? RCCL_MIN : RCCL_SUM;
    return rccl_try(g_rccl.AllReduce(buffer, buffer, (size_t)count, type, op, ctx->comm,
                                     ctx->stream), "ncclAllReduce");
  }
  if (!callback) {
    sdm_set_error("sharded mode: neither a communicator (sdm_comm_init) nor an exchange callback");
    return SDM_E_ARG;
  }
  if (callback(user, what, buffer, count) != 0) {
    sdm_set_error("sharded mode: the exchange callback failed");
    return SDM_E_HIP;
  }
  return SDM_OK;
}
