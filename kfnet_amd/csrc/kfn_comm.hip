// kfn_comm.hip -- the rank -> rank hand-off of the recurrent Kalman state over RCCL.
//
// No reference counterpart: zlthinker/KFNet is single-process, single-device (the only device
// placement in the tree is KFNet/train.py:375).  This is the one exchange of the frame-sharded
// configuration: rank r sends its [H,W,4] fp32 state (76.8 KB at 60x80)
// to rank r+1 with ncclSend/ncclRecv, stream-ordered on the stream the scan kernel runs on, so
// the receive is ordered before the consumer's kfn_kalman_scan and the send after the
// producer's.  There is no collective on the data path.
//
// librccl is bound at run time (dlopen "librccl.so.1"): a process that already loaded RCCL
// (PyTorch ships one with the same SONAME) shares that copy, and single-GPU hosts never need
// the library at all.
#include "kfn_common.h"
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>

struct kfn_comm {
  ncclComm_t comm;
  int rank, nranks, device;
};

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  char why[256] = {0};
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.handle) break;
  }
  if (!g_rccl.handle) {
    snprintf(g_rccl.why, sizeof(g_rccl.why), "dlopen(librccl.so.1): %s", dlerror());
    return;
  }
  auto sym = [&](const char* name) -> void* {
    void* p = dlsym(g_rccl.handle, name);
    if (!p && !g_rccl.why[0]) snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl: missing symbol %s", name);
    return p;
  };
  g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
  g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
  g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
  g_rccl.CommUserRank = reinterpret_cast<decltype(g_rccl.CommUserRank)>(sym("ncclCommUserRank"));
  g_rccl.CommCount = reinterpret_cast<decltype(g_rccl.CommCount)>(sym("ncclCommCount"));
  g_rccl.CommCuDevice = reinterpret_cast<decltype(g_rccl.CommCuDevice)>(sym("ncclCommCuDevice"));
  g_rccl.Send = reinterpret_cast<decltype(g_rccl.Send)>(sym("ncclSend"));
  g_rccl.Recv = reinterpret_cast<decltype(g_rccl.Recv)>(sym("ncclRecv"));
  g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
}

int need_rccl(const char* who) {
  std::call_once(g_rccl_once, load_rccl);
  if (g_rccl.why[0]) return kfn::fail(KFN_ERR_UNSUPPORTED, "%s: RCCL unavailable (%s)", who, g_rccl.why);
  return KFN_OK;
}

int check_nccl(ncclResult_t r, const char* what) {
  if (r == ncclSuccess) return KFN_OK;
  return kfn::fail(KFN_ERR_HIP, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
}

int check_state_args(const kfn_comm* c, int peer, const void* state, int H, int W, const char* who) {
  KFN_REQUIRE(c != nullptr, "%s: null communicator", who);
  KFN_REQUIRE(state != nullptr, "%s: null state pointer", who);
  KFN_REQUIRE(H > 0 && W > 0, "%s: bad grid %dx%d", who, H, W);
  KFN_REQUIRE(peer >= 0 && peer < c->nranks && peer != c->rank, "%s: bad peer %d (rank %d of %d)", who, peer,
              c->rank, c->nranks);
  return KFN_OK;
}

}  // namespace

// Can this process bind librccl (dlopen + every symbol)?  No communicator, no bootstrap socket, no device access: the
// probe every rank runs before the collective part of a link set-up (kfnet_amd/dist.py::make_link).
extern "C" int kfn_comm_available(void) { return need_rccl("kfn_comm_available"); }

extern "C" int kfn_comm_unique_id(void* id, size_t bytes) {
  KFN_REQUIRE(id != nullptr && bytes == KFN_COMM_ID_BYTES, "kfn_comm_unique_id: id must be a %d-byte buffer",
              KFN_COMM_ID_BYTES);
  static_assert(KFN_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
  int rc = need_rccl("kfn_comm_unique_id");
  if (rc != KFN_OK) return rc;
  ncclUniqueId uid;
  rc = check_nccl(g_rccl.GetUniqueId(&uid), "ncclGetUniqueId");
  if (rc != KFN_OK) return rc;
  memcpy(id, uid.internal, KFN_COMM_ID_BYTES);
  return KFN_OK;
}

extern "C" int kfn_comm_init(kfn_comm** comm, int rank, int nranks, const void* unique_id, int device) {
  KFN_REQUIRE(comm != nullptr, "kfn_comm_init: null out pointer");
  *comm = nullptr;
  KFN_REQUIRE(unique_id != nullptr, "kfn_comm_init: null unique id");
  KFN_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "kfn_comm_init: bad rank %d of %d", rank, nranks);
  KFN_REQUIRE(device >= 0, "kfn_comm_init: bad device %d", device);
  int rc = need_rccl("kfn_comm_init");
  if (rc != KFN_OK) return rc;
  KFN_HIP(hipSetDevice(device));   // the communicator binds to the calling thread's current device
  ncclUniqueId uid;
  memcpy(uid.internal, unique_id, KFN_COMM_ID_BYTES);
  ncclComm_t c = nullptr;
  rc = check_nccl(g_rccl.CommInitRank(&c, nranks, uid, rank), "ncclCommInitRank");
  if (rc != KFN_OK) return rc;
  kfn_comm* out = new kfn_comm{c, rank, nranks, device};
  *comm = out;
  return KFN_OK;
}

extern "C" int kfn_comm_destroy(kfn_comm* comm) {
  if (comm == nullptr) return KFN_OK;
  int rc = KFN_OK;
  if (comm->comm != nullptr && g_rccl.CommDestroy) rc = check_nccl(g_rccl.CommDestroy(comm->comm), "ncclCommDestroy");
  delete comm;
  return rc;
}

// RCCL's OWN view of the communicator (ncclCommUserRank / ncclCommCount / ncclCommCuDevice), not an echo of what
// kfn_comm_init was told: bench.py's `rccl_ranks` is this.  A communicator whose answers differ from the arguments
// it was created with is reported as an error.
extern "C" int kfn_comm_rank(const kfn_comm* comm, int* rank, int* nranks) {
  KFN_REQUIRE(comm != nullptr, "kfn_comm_rank: null communicator");
  KFN_REQUIRE(comm->comm != nullptr && g_rccl.CommUserRank && g_rccl.CommCount, "kfn_comm_rank: no RCCL communicator");
  int r = -1, n = -1, dev = -1;
  int rc = check_nccl(g_rccl.CommUserRank(comm->comm, &r), "ncclCommUserRank");
  if (rc != KFN_OK) return rc;
  rc = check_nccl(g_rccl.CommCount(comm->comm, &n), "ncclCommCount");
  if (rc != KFN_OK) return rc;
  if (g_rccl.CommCuDevice) {
    rc = check_nccl(g_rccl.CommCuDevice(comm->comm, &dev), "ncclCommCuDevice");
    if (rc != KFN_OK) return rc;
    if (dev != comm->device)
      return kfn::fail(KFN_ERR_HIP, "kfn_comm_rank: RCCL reports device %d, the communicator was created on %d", dev, comm->device);
  }
  if (r != comm->rank || n != comm->nranks)
    return kfn::fail(KFN_ERR_HIP, "kfn_comm_rank: RCCL reports rank %d of %d, the communicator was created as %d of %d", r, n,
                     comm->rank, comm->nranks);
  if (rank) *rank = r;
  if (nranks) *nranks = n;
  return KFN_OK;
}

extern "C" int kfn_send_state(kfn_comm* comm, int peer, const float* state, int H, int W, void* stream) {
  int rc = check_state_args(comm, peer, state, H, W, "kfn_send_state");
  if (rc != KFN_OK) return rc;
  return check_nccl(g_rccl.Send(state, (size_t)H * W * 4, ncclFloat32, peer, comm->comm,
                                reinterpret_cast<hipStream_t>(stream)), "ncclSend");
}

extern "C" int kfn_recv_state(kfn_comm* comm, int peer, float* state, int H, int W, void* stream) {
  int rc = check_state_args(comm, peer, state, H, W, "kfn_recv_state");
  if (rc != KFN_OK) return rc;
  return check_nccl(g_rccl.Recv(state, (size_t)H * W * 4, ncclFloat32, peer, comm->comm,
                                reinterpret_cast<hipStream_t>(stream)), "ncclRecv");
}
