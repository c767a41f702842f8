// The collectives of the data-parallel step on the C side of the boundary (rk_allreduce_bucket): thin
// wrappers of RCCL that enqueue IN ORDER on the caller's HIP stream -- the all-reduce / reduce-scatter + all-gather
// of the gradient buckets, the MAX all-reduce of the blocks' item stamps, the grouped point-to-point exchange of the
// owned-row update.  The reference has no multi-device code (model.py:397-402 is its whole backward + update); this is
// what "users sharded over the GPUs of a node with an all-reduce of the weight gradients over xGMI" needs at the ABI.
//
// RCCL is bound at RUN time (dlopen + dlsym), not at link time: a process must hold ONE librccl -- the one PyTorch
// has loaded when the caller is a PyTorch process (the `librccl` argument names it; default: whatever "librccl.so"
// resolves to, an already loaded one first) -- and the library must load on a box without RCCL at all (single GPU).
#include <dlfcn.h>
#include <stdint.h>
#include <string.h>

#include "common.h"

namespace {

typedef struct { char internal[128]; } nccl_uid_t;
typedef void *nccl_comm_t;
enum { NCCL_SUM = 0, NCCL_MAX = 2, NCCL_INT32 = 2, NCCL_FLOAT32 = 7 };

struct Api {
  void *handle = nullptr;
  int (*GetUniqueId)(nccl_uid_t *) = nullptr;
  int (*CommInitRank)(nccl_comm_t *, int, nccl_uid_t, int) = nullptr;
  int (*CommDestroy)(nccl_comm_t) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*ReduceScatter)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*Send)(const void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
Api g_api;
char g_path[1024] = "";

bool load_api() {
  if (g_api.handle) return true;
  void *h = nullptr;
  if (g_path[0]) h = dlopen(g_path, RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);     // the one the process already holds
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) { rk_set_error("rk_comm: cannot load librccl.so (%s)", dlerror()); return false; }
#define SYM(field, name)                                                                      \
  do {                                                                                        \
    *(void **)(&g_api.field) = dlsym(h, name);                                                \
    if (!g_api.field) { rk_set_error("rk_comm: librccl.so lacks %s", name); dlclose(h); g_api = Api(); return false; } \
  } while (0)
  SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
  SYM(AllReduce, "ncclAllReduce"); SYM(ReduceScatter, "ncclReduceScatter"); SYM(AllGather, "ncclAllGather");
  SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv"); SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  g_api.handle = h;
  return true;
}

inline int fail(int rc, const char *what) {
  rk_set_error("%s: %s", what, g_api.GetErrorString ? g_api.GetErrorString(rc) : "RCCL error");
  return -1;
}
#define NCCL_TRY(call, what)                 \
  do {                                       \
    const int rc__ = (call);                 \
    if (rc__ != 0) return fail(rc__, what);  \
  } while (0)

inline int dt_of(int dtype) { return dtype == RK_COMM_I32 ? NCCL_INT32 : NCCL_FLOAT32; }
inline int op_of(int op) { return op == RK_COMM_MAX ? NCCL_MAX : NCCL_SUM; }

}  // namespace

static bool name_library(const char *path) {
  if (path == nullptr || g_api.handle != nullptr) return true;    // (already loaded: the process holds one RCCL)
  if (strlen(path) >= sizeof(g_path)) { rk_set_error("rk_comm: library path too long"); return false; }
  strcpy(g_path, path);
  return true;
}

extern "C" int rk_comm_unique_id(void *id128, const char *librccl) {
  RK_REQUIRE(id128 != nullptr, "null id");
  if (!name_library(librccl) || !load_api()) return -1;
  nccl_uid_t id;
  NCCL_TRY(g_api.GetUniqueId(&id), "ncclGetUniqueId");
  memcpy(id128, &id, sizeof(id));
  return 0;
}

extern "C" void *rk_comm_init(const void *id128, int32_t world, int32_t rank, const char *librccl) {
  if (id128 == nullptr || world < 1 || rank < 0 || rank >= world) { rk_set_error("rk_comm_init: bad arguments"); return nullptr; }
  if (!name_library(librccl) || !load_api()) return nullptr;
  nccl_uid_t id;
  memcpy(&id, id128, sizeof(id));
  nccl_comm_t comm = nullptr;
  const int rc = g_api.CommInitRank(&comm, world, id, rank);
  if (rc != 0) { fail(rc, "ncclCommInitRank"); return nullptr; }
  return comm;
}

extern "C" void rk_comm_destroy(void *comm) {
  if (comm && g_api.CommDestroy) (void)g_api.CommDestroy((nccl_comm_t)comm);
}

extern "C" int rk_allreduce_bucket(void *comm, void *const *bufs, const int64_t *counts, int32_t n, int32_t dtype,
                                   int32_t op, void *stream) {
  RK_REQUIRE(comm != nullptr && g_api.handle != nullptr, "no communicator");
  RK_REQUIRE(n >= 0 && (n == 0 || (bufs && counts)), "null buckets");
  int live = 0;
  for (int i = 0; i < n; ++i) live += counts[i] > 0;
  if (live == 0) return 0;
  if (live > 1) NCCL_TRY(g_api.GroupStart(), "ncclGroupStart");
  int rc = 0;
  for (int i = 0; i < n && rc == 0; ++i)
    if (counts[i] > 0)
      rc = g_api.AllReduce(bufs[i], bufs[i], (size_t)counts[i], dt_of(dtype), op_of(op), (nccl_comm_t)comm, (hipStream_t)stream);
  if (live > 1) {
    const int rc2 = g_api.GroupEnd();
    if (rc == 0) rc = rc2;
  }
  if (rc != 0) return fail(rc, "ncclAllReduce (group)");
  return 0;
}

extern "C" int rk_reduce_scatter(void *comm, const void *send, void *recv, int64_t recv_count, int32_t dtype,
                                 void *stream) {
  RK_REQUIRE(comm != nullptr && g_api.handle != nullptr, "no communicator");
  if (recv_count == 0) return 0;
  NCCL_TRY(g_api.ReduceScatter(send, recv, (size_t)recv_count, dt_of(dtype), NCCL_SUM, (nccl_comm_t)comm, (hipStream_t)stream),
           "ncclReduceScatter");
  return 0;
}

extern "C" int rk_all_gather(void *comm, const void *send, void *recv, int64_t send_count, int32_t dtype, void *stream) {
  RK_REQUIRE(comm != nullptr && g_api.handle != nullptr, "no communicator");
  if (send_count == 0) return 0;
  NCCL_TRY(g_api.AllGather(send, recv, (size_t)send_count, dt_of(dtype), (nccl_comm_t)comm, (hipStream_t)stream), "ncclAllGather");
  return 0;
}

extern "C" int rk_exchange(void *comm, int32_t world, void *const *sends, const int64_t *send_counts, void *const *recvs,
                           const int64_t *recv_counts, int32_t dtype, void *stream) {
  RK_REQUIRE(comm != nullptr && g_api.handle != nullptr, "no communicator");
  RK_REQUIRE(world >= 1 && sends && send_counts && recvs && recv_counts, "null peers");
  NCCL_TRY(g_api.GroupStart(), "ncclGroupStart");
  int rc = 0;
  for (int q = 0; q < world && rc == 0; ++q) {
    if (send_counts[q] > 0)
      rc = g_api.Send(sends[q], (size_t)send_counts[q], dt_of(dtype), q, (nccl_comm_t)comm, (hipStream_t)stream);
    if (rc == 0 && recv_counts[q] > 0)
      rc = g_api.Recv(recvs[q], (size_t)recv_counts[q], dt_of(dtype), q, (nccl_comm_t)comm, (hipStream_t)stream);
  }
  const int rc2 = g_api.GroupEnd();
  if (rc == 0) rc = rc2;
  if (rc != 0) return fail(rc, "ncclSend / ncclRecv (group)");
  return 0;
}
