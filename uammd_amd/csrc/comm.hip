// Multi-GPU communication behind the C ABI: RCCL over xGMI, one process per GPU (SURVEY 8b "uammd_comm_*", 8e).
//
// New design — the reference is single GPU (no NCCL / MPI anywhere in /root/reference/src).  A C++14 UAMMD program drives N GPUs by
// creating one communicator per process and calling these entry points between the library's own kernels:
//   path A (LJ slabs):  positions of the one-cut-off halo to the two z neighbours (point to point, one xGMI link each),
//                       migrating particles likewise; forces need no reduction (full per-particle forces, common.cuh:10-34);
//   path B (FCM slabs): halo planes of the spread / velocity grids to the two neighbours, one all-to-all transpose per 3-D FFT
//                       direction (each rank sends 1/world of its planes to every peer: one block per link);
//   Lanczos:            scalar all-reduces of the recurrence's dot products.
// RCCL is loaded with dlopen on first use (librccl.so.1): programs that stay on one GPU never touch it, and a process that
// already holds a copy (PyTorch ships one) shares it instead of loading a second set of nccl* symbols.
#include "device_common.hpp"
#include "../../include/uammd_hip.h"

#include <dlfcn.h>
#include <cstring>

namespace uammd_hip {

// the subset of the NCCL API used here (RCCL keeps NCCL's names and ABI: rccl.h)
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5,
               ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;

struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int *) = nullptr;  // (optional)
};

static Rccl g_rccl;

static int rccl_load() {
  if (g_rccl.lib) return 0;
  void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!lib) { set_last_error("uammd_comm: cannot load librccl.so.1 (%s)", dlerror()); return -1; }
#define UH_SYM(field, name)                                                                             \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(lib, name));                           \
  if (!g_rccl.field) { set_last_error("uammd_comm: librccl has no symbol %s", name); dlclose(lib); return -1; }
  UH_SYM(GetUniqueId, "ncclGetUniqueId")
  UH_SYM(CommInitRank, "ncclCommInitRank")
  UH_SYM(CommDestroy, "ncclCommDestroy")
  UH_SYM(Send, "ncclSend")
  UH_SYM(Recv, "ncclRecv")
  UH_SYM(AllReduce, "ncclAllReduce")
  UH_SYM(GroupStart, "ncclGroupStart")
  UH_SYM(GroupEnd, "ncclGroupEnd")
  UH_SYM(GetErrorString, "ncclGetErrorString")
#undef UH_SYM
  g_rccl.GetVersion = reinterpret_cast<decltype(g_rccl.GetVersion)>(dlsym(lib, "ncclGetVersion"));
  g_rccl.lib = lib;
  return 0;
}

#define UH_NCCL(expr)                                                                                   \
  do {                                                                                                  \
    ncclResult_t r_ = (expr);                                                                           \
    if (r_ != ncclSuccess) {                                                                            \
      set_last_error("%s failed: %s", #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); \
      return -20 - (int)r_;                                                                             \
    }                                                                                                   \
  } while (0)

struct Comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  int *d_counts = nullptr;   // 4 ints of device scratch for uammd_comm_exchange_counts
  int *h_counts = nullptr;   // 4 ints of pinned host memory: the landing place of uammd_comm_exchange_counts_device's one read
  // UAMMD_COMM_SELF_THROUGH_RCCL=1: the all-to-all sends the rank's own block through RCCL too (how it was until round 5: A/B runs)
  bool selfThroughRccl = getenv("UAMMD_COMM_SELF_THROUGH_RCCL") && atoi(getenv("UAMMD_COMM_SELF_THROUGH_RCCL")) != 0;
};

}  // namespace uammd_hip

using namespace uammd_hip;

extern "C" {

int uammd_comm_unique_id(char id[128]) {
  if (!id) { set_last_error("uammd_comm_unique_id: null argument"); return -1; }
  if (int e = rccl_load()) return e;
  ncclUniqueId u;
  UH_NCCL(g_rccl.GetUniqueId(&u));
  std::memcpy(id, u.internal, 128);
  return 0;
}

int uammd_comm_rccl_version(int *version) {
  if (!version) { set_last_error("uammd_comm_rccl_version: null argument"); return -1; }
  if (int e = rccl_load()) return e;
  if (!g_rccl.GetVersion) { set_last_error("uammd_comm_rccl_version: librccl has no ncclGetVersion"); return -1; }
  UH_NCCL(g_rccl.GetVersion(version));
  return 0;
}

int uammd_comm_init(uammd_comm **out, int rank, int world, const char id[128]) {
  if (!out || !id || world < 1 || rank < 0 || rank >= world) { set_last_error("uammd_comm_init: bad arguments"); return -1; }
  if (int e = rccl_load()) return e;
  Comm *c = new Comm();
  c->rank = rank;
  c->world = world;
  ncclUniqueId u;
  std::memcpy(u.internal, id, 128);
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, u, rank);
  if (r != ncclSuccess) {
    set_last_error("ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
    delete c;
    return -20 - (int)r;
  }
  if (hipMalloc((void **)&c->d_counts, 4 * sizeof(int)) != hipSuccess || hipHostMalloc((void **)&c->h_counts, 4 * sizeof(int)) != hipSuccess) {
    set_last_error("uammd_comm_init: hipMalloc failed");
    (void)g_rccl.CommDestroy(c->comm);
    if (c->d_counts) (void)hipFree(c->d_counts);
    delete c;
    return -2;
  }
  *out = reinterpret_cast<uammd_comm *>(c);
  return 0;
}

int uammd_comm_destroy(uammd_comm *h) {
  if (!h) return 0;
  Comm *c = reinterpret_cast<Comm *>(h);
  if (c->comm) (void)g_rccl.CommDestroy(c->comm);
  if (c->d_counts) (void)hipFree(c->d_counts);
  if (c->h_counts) (void)hipHostFree(c->h_counts);
  delete c;
  return 0;
}

int uammd_comm_rank(const uammd_comm *h) { return h ? reinterpret_cast<const Comm *>(h)->rank : -1; }
int uammd_comm_world(const uammd_comm *h) { return h ? reinterpret_cast<const Comm *>(h)->world : -1; }

// Rows to rank + 1 ("up") and rank - 1 ("down") of the periodic ring; rows from rank - 1 and rank + 1.  One grouped call: each
// message rides its own xGMI link.  With world = 1 the rank is its own neighbour on both sides (what goes up arrives from below).
int uammd_comm_halo_exchange(uammd_comm *h, const float *d_sendUp, int nUp, const float *d_sendDown, int nDown, float *d_recvFromDown,
                             int nFromDown, float *d_recvFromUp, int nFromUp, int floatsPerRow, void *stream) {
  if (!h || floatsPerRow <= 0 || nUp < 0 || nDown < 0 || nFromDown < 0 || nFromUp < 0) { set_last_error("uammd_comm_halo_exchange: bad arguments"); return -1; }
  Comm *c = reinterpret_cast<Comm *>(h);
  hipStream_t st = (hipStream_t)stream;
  const int up = (c->rank + 1) % c->world, down = (c->rank + c->world - 1) % c->world;
  const size_t w = (size_t)floatsPerRow;
  UH_NCCL(g_rccl.GroupStart());
  if (nUp) UH_NCCL(g_rccl.Send(d_sendUp, w * nUp, ncclFloat32, up, c->comm, st));
  if (nFromDown) UH_NCCL(g_rccl.Recv(d_recvFromDown, w * nFromDown, ncclFloat32, down, c->comm, st));
  if (nDown) UH_NCCL(g_rccl.Send(d_sendDown, w * nDown, ncclFloat32, down, c->comm, st));
  if (nFromUp) UH_NCCL(g_rccl.Recv(d_recvFromUp, w * nFromUp, ncclFloat32, up, c->comm, st));
  UH_NCCL(g_rccl.GroupEnd());
  return 0;
}

// The two message sizes a refresh needs, to the two neighbours only: counts[0] goes up, counts[1] goes down; on return
// fromDown / fromUp hold what this rank will receive.  Synchronises the stream (the sizes are needed on the host).
int uammd_comm_exchange_counts(uammd_comm *h, const int toUpDown[2], int fromDownUp[2], void *stream) {
  if (!h || !toUpDown || !fromDownUp) { set_last_error("uammd_comm_exchange_counts: null argument"); return -1; }
  Comm *c = reinterpret_cast<Comm *>(h);
  hipStream_t st = (hipStream_t)stream;
  const int up = (c->rank + 1) % c->world, down = (c->rank + c->world - 1) % c->world;
  UH_CHECK(hipMemcpyAsync(c->d_counts, toUpDown, 2 * sizeof(int), hipMemcpyHostToDevice, st));
  UH_NCCL(g_rccl.GroupStart());
  UH_NCCL(g_rccl.Send(c->d_counts, 1, ncclInt32, up, c->comm, st));
  UH_NCCL(g_rccl.Recv(c->d_counts + 2, 1, ncclInt32, down, c->comm, st));
  UH_NCCL(g_rccl.Send(c->d_counts + 1, 1, ncclInt32, down, c->comm, st));
  UH_NCCL(g_rccl.Recv(c->d_counts + 3, 1, ncclInt32, up, c->comm, st));
  UH_NCCL(g_rccl.GroupEnd());
  UH_CHECK(hipMemcpyAsync(fromDownUp, c->d_counts + 2, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
  UH_CHECK(hipStreamSynchronize(st));
  return 0;
}

// The same exchange with the two sizes still on the device (where the selection kernel left them: uammd_slab_select's counts): they go to
// the neighbours from there, and ONE read returns all four numbers {toUp, toDown, fromDown, fromUp} — one stream synchronisation per
// phase of a refresh instead of two (the read of the own sizes, then the exchange's).
int uammd_comm_exchange_counts_device(uammd_comm *h, const int *d_toUpDown, int all4[4], void *stream) {
  if (!h || !d_toUpDown || !all4) { set_last_error("uammd_comm_exchange_counts_device: null argument"); return -1; }
  Comm *c = reinterpret_cast<Comm *>(h);
  hipStream_t st = (hipStream_t)stream;
  const int up = (c->rank + 1) % c->world, down = (c->rank + c->world - 1) % c->world;
  UH_NCCL(g_rccl.GroupStart());
  UH_NCCL(g_rccl.Send(d_toUpDown, 1, ncclInt32, up, c->comm, st));
  UH_NCCL(g_rccl.Recv(c->d_counts + 2, 1, ncclInt32, down, c->comm, st));
  UH_NCCL(g_rccl.Send(d_toUpDown + 1, 1, ncclInt32, down, c->comm, st));
  UH_NCCL(g_rccl.Recv(c->d_counts + 3, 1, ncclInt32, up, c->comm, st));
  UH_NCCL(g_rccl.GroupEnd());
  UH_CHECK(hipMemcpyAsync(c->d_counts, d_toUpDown, 2 * sizeof(int), hipMemcpyDeviceToDevice, st));
  UH_CHECK(hipMemcpyAsync(c->h_counts, c->d_counts, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
  UH_CHECK(hipStreamSynchronize(st));
  for (int k = 0; k < 4; ++k) all4[k] = c->h_counts[k];
  return 0;
}

// The transpose of the slab FFT: block p of d_send (bytesPerPeer bytes at offset p * bytesPerPeer) goes to rank p, block p of d_recv
// comes from rank p.  Grouped send / recv pairs: xGMI is point to point, every block travels on its own link.
int uammd_comm_alltoall(uammd_comm *h, const void *d_send, void *d_recv, size_t bytesPerPeer, void *stream) {
  if (!h || !d_send || !d_recv) { set_last_error("uammd_comm_alltoall: null argument"); return -1; }
  Comm *c = reinterpret_cast<Comm *>(h);
  hipStream_t st = (hipStream_t)stream;
  if (bytesPerPeer == 0) return 0;
  // the rank's OWN block (1 / world of the transpose) does not go through the network layer: a send to oneself is a kernel and a staging
  // copy there (18.7 + 5 us for 25 MB at a world of one); here it is one device-to-device copy on the same stream
  if (!c->selfThroughRccl)
    UH_CHECK(hipMemcpyAsync((char *)d_recv + (size_t)c->rank * bytesPerPeer, (const char *)d_send + (size_t)c->rank * bytesPerPeer, bytesPerPeer,
                            hipMemcpyDeviceToDevice, st));
  if (c->world == 1 && !c->selfThroughRccl) return 0;
  UH_NCCL(g_rccl.GroupStart());
  for (int p = 0; p < c->world; ++p) {
    if (p == c->rank && !c->selfThroughRccl) continue;
    UH_NCCL(g_rccl.Send((const char *)d_send + (size_t)p * bytesPerPeer, bytesPerPeer, ncclInt8, p, c->comm, st));
    UH_NCCL(g_rccl.Recv((char *)d_recv + (size_t)p * bytesPerPeer, bytesPerPeer, ncclInt8, p, c->comm, st));
  }
  UH_NCCL(g_rccl.GroupEnd());
  return 0;
}

// in-place sum over the ranks (energy / virial totals, the Lanczos recurrence's scalars)
int uammd_comm_allreduce_sum(uammd_comm *h, float *d_buf, int n, void *stream) {
  if (!h || !d_buf || n < 0) { set_last_error("uammd_comm_allreduce_sum: bad arguments"); return -1; }
  if (n == 0) return 0;
  Comm *c = reinterpret_cast<Comm *>(h);
  UH_NCCL(g_rccl.AllReduce(d_buf, d_buf, (size_t)n, ncclFloat32, ncclSum, c->comm, (hipStream_t)stream));
  return 0;
}

}  // extern "C"
