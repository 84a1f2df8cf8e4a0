// comm.hip — the multi-GPU exchange step of the frontend path behind the C-ABI.
// The reference is one process per node with no collective (its only "communication" is the DDS topic
// /frontend/keyframe, frontend.cpp:200,783); sharding frames (or pyramid levels) over the GPUs of a node adds
// exactly ONE exchange: an all-gather of fixed-size per-rank blocks over RCCL/xGMI.  A C++ host reaches it here;
// dvslam_amd/dist.py is the gloo test double of the same block layout.
//
// RCCL is resolved with dlopen at first use: single-GPU callers never need librccl, and a process that already
// carries an RCCL (PyTorch bundles one under the same SONAME librccl.so.1) shares that copy instead of loading a second.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include "common.h"

using namespace dvs;

namespace {

// the handful of RCCL entry points used (rccl.h: ncclGetUniqueId, ncclCommInitRank, ncclAllGather, ncclCommDestroy, ...)
struct NcclUniqueId { char internal[DVS_COMM_ID_BYTES]; };
typedef struct ncclComm* NcclComm;
enum { kNcclSuccess = 0, kNcclUint8 = 1 };  // ncclResult_t::ncclSuccess, ncclDataType_t::ncclUint8 (= ncclChar + 1)
struct Rccl {
  void* so = nullptr;
  int (*GetVersion)(int*) = nullptr;
  int (*GetUniqueId)(NcclUniqueId*) = nullptr;
  int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
  int (*CommDestroy)(NcclComm) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, NcclComm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;

dvs_status load_rccl() {
  if (g_rccl.so) return DVS_OK;
  const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
  void* so = nullptr;
  for (const char* n : names)
    if ((so = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!so) { set_error("librccl.so.1 not found (%s): the multi-GPU exchange needs RCCL", dlerror()); return DVS_ERR_UNSUPPORTED; }
  Rccl r;
  r.so = so;
  r.GetVersion = (int (*)(int*))dlsym(so, "ncclGetVersion");
  r.GetUniqueId = (int (*)(NcclUniqueId*))dlsym(so, "ncclGetUniqueId");
  r.CommInitRank = (int (*)(NcclComm*, int, NcclUniqueId, int))dlsym(so, "ncclCommInitRank");
  r.CommDestroy = (int (*)(NcclComm))dlsym(so, "ncclCommDestroy");
  r.AllGather = (int (*)(const void*, void*, size_t, int, NcclComm, hipStream_t))dlsym(so, "ncclAllGather");
  r.GetErrorString = (const char* (*)(int))dlsym(so, "ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather) {
    set_error("librccl.so.1 lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy");
    return DVS_ERR_UNSUPPORTED;
  }
  g_rccl = r;
  return DVS_OK;
}

#define DVS_NCCL(call)                                                                                                   \
  do {                                                                                                                   \
    int r_ = (call);                                                                                                     \
    if (r_ != kNcclSuccess) {                                                                                            \
      set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); \
      return DVS_ERR_HIP;                                                                                                \
    }                                                                                                                    \
  } while (0)

// this rank's boundary block {descriptors[cap x 32], n, padding}: 16-byte copies of the descriptor rows of the last frame
__global__ void __launch_bounds__(256) k_pack_boundary(const uint4* __restrict__ desc, const int* __restrict__ n, uint4* __restrict__ block,
                                                       int rows16, int blk16) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < rows16) block[i] = desc[i];
  else if (i < blk16) block[i] = i == rows16 ? make_uint4((uint32_t)n[0], 0u, 0u, 0u) : make_uint4(0u, 0u, 0u, 0u);
}

// Loopback group (dvs_comm_create_loopback): `world` logical ranks of ONE process on ONE device, each driven by its own host thread
// with its own streams.  The all-gather keeps the collective's contract — every rank contributes one block and receives all — with a
// host rendezvous in place of RCCL's: each rank records an event behind its send block, all ranks meet (a timed barrier: a rank
// that never arrives fails the others instead of hanging them), then every rank's stream waits for the peers' events and pulls
// their blocks with device-to-device copies.  It exists so that the multi-rank branches of dvs_exchange_boundary (rank > 0: the
// previous rank's block of this call; rank 0: the last rank's block of the previous call) run on a one-GPU box.
struct LoopGroup {
  int world = 0, device = 0, refs = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  unsigned long generation = 0;
  bool broken = false;
  const void* send[DVS_COMM_MAX_LOOPBACK] = {};
  hipEvent_t ev_sent[DVS_COMM_MAX_LOOPBACK] = {};     // rank's send block is complete (its stream)
  hipEvent_t ev_pulled[DVS_COMM_MAX_LOOPBACK] = {};   // rank has pulled every peer's block of its latest call (its stream)
  std::atomic<bool> pulled_once{false};   // set by every rank behind the second rendezvous of the first gather
  // every rank arrives; false when the group is broken (a peer timed out or failed)
  bool barrier() {
    std::unique_lock<std::mutex> lk(mu);
    if (broken) return false;
    const unsigned long gen = generation;
    if (++arrived == world) { arrived = 0; generation++; cv.notify_all(); return true; }
    if (!cv.wait_for(lk, std::chrono::seconds(30), [&] { return generation != gen || broken; })) { broken = true; cv.notify_all(); }
    return !broken;
  }
};

}  // namespace

struct dvs_comm {
  int device = 0, rank = 0, world = 1;
  NcclComm comm = nullptr;
  LoopGroup* loop = nullptr;   // loopback communicator: no RCCL
  // host-transport communicator (dvs_comm_create_host): blocks and gather buffers live in HOST memory, the all-gather is the caller's
  // callback (MPI_Allgather, a torch.distributed group, sockets).  No device is touched: the rank logic of dvs_exchange_boundary —
  // buffer rotation, slot of this rank, predecessor selection with the wrap-around to the previous call — is the same code for every
  // transport, and this one lets two OS processes without a GPU run it (tests/test_adapters_and_dist.py)
  dvs_host_all_gather_fn host_gather = nullptr;
  void* host_user = nullptr;
  uint8_t* gather[3] = {nullptr, nullptr, nullptr};  // [world][block] x 3: a call's result points into this call's and the previous
                                                     // call's buffer, and stays valid while the next call gathers into the third
  long calls = 0;
  size_t block = 0;
  int turn = 0;
  // a caller may issue successive calls on different streams (the lane schedule of dvs_pipeline): a call that finds another stream
  // than its predecessor's orders itself behind that call's gather — rank 0's result points into the PREVIOUS call's buffer
  // (the first change of stream is covered by a host wait, from then on every call records an event: single-stream callers pay nothing)
  hipEvent_t ev_last = nullptr;
  hipStream_t last_stream = nullptr;
  bool has_last = false, multi_stream = false;
};

namespace {

// loopback: a rank's send block may be pulled by peers until their latest pulls have run — order the block's next overwrite behind them
dvs_status loop_before_send(dvs_comm* c, hipStream_t st) {
  LoopGroup* G = c->loop;
  if (!G->pulled_once) return DVS_OK;
  for (int p = 0; p < G->world; p++)
    if (p != c->rank) DVS_HIP(hipStreamWaitEvent(st, G->ev_pulled[p], 0));
  return DVS_OK;
}

dvs_status loop_all_gather(dvs_comm* c, const void* send, void* recv, size_t bytes, hipStream_t st) {
  LoopGroup* G = c->loop;
  G->send[c->rank] = send;
  DVS_HIP(hipEventRecord(G->ev_sent[c->rank], st));
  if (!G->barrier()) { set_error("loopback all-gather: a rank of the group did not arrive (rank %d waited)", c->rank); return DVS_ERR_HIP; }
  for (int p = 0; p < G->world; p++) {
    uint8_t* dst = (uint8_t*)recv + (size_t)p * bytes;
    if (p == c->rank) {
      if (dst != send) DVS_HIP(hipMemcpyAsync(dst, send, bytes, hipMemcpyDeviceToDevice, st));
      continue;
    }
    DVS_HIP(hipStreamWaitEvent(st, G->ev_sent[p], 0));
    DVS_HIP(hipMemcpyAsync(dst, G->send[p], bytes, hipMemcpyDeviceToDevice, st));
  }
  DVS_HIP(hipEventRecord(G->ev_pulled[c->rank], st));
  // nobody re-records its events or moves its send pointer before every rank has enqueued its pulls
  if (!G->barrier()) { set_error("loopback all-gather: a rank of the group did not arrive (rank %d waited)", c->rank); return DVS_ERR_HIP; }
  G->pulled_once = true;
  return DVS_OK;
}

dvs_status comm_all_gather(dvs_comm* c, const void* send, void* recv, size_t bytes, hipStream_t st) {
  if (c->host_gather) {
    const int r = c->host_gather(c->host_user, send, recv, bytes);
    if (r != 0) { set_error("host all-gather callback failed with %d (rank %d of %d)", r, c->rank, c->world); return DVS_ERR_HIP; }
    return DVS_OK;
  }
  if (c->loop) return loop_all_gather(c, send, recv, bytes, st);
  DVS_NCCL(g_rccl.AllGather(send, recv, bytes, kNcclUint8, c->comm, st));
  return DVS_OK;
}

}  // namespace

extern "C" {

size_t dvs_boundary_block_bytes(int32_t cap) { return cap < 0 ? 0 : ((size_t)cap * 32 + 4 + 63) / 64 * 64; }

dvs_status dvs_comm_get_unique_id(uint8_t* id) {
  DVS_ARG(id);
  DVS_TRY(load_rccl());
  NcclUniqueId u;
  DVS_NCCL(g_rccl.GetUniqueId(&u));
  memcpy(id, u.internal, DVS_COMM_ID_BYTES);
  return DVS_OK;
}

dvs_status dvs_comm_create(int32_t device, int32_t rank, int32_t world, const uint8_t* id, dvs_comm** out) {
  DVS_ARG(out && id && world >= 1 && rank >= 0 && rank < world);
  *out = nullptr;
  DVS_TRY(check_device(device));
  DVS_TRY(load_rccl());
  dvs_comm* c = new (std::nothrow) dvs_comm();
  if (!c) { set_error("out of host memory"); return DVS_ERR_HIP; }
  c->device = device; c->rank = rank; c->world = world;
  NcclUniqueId u;
  memcpy(u.internal, id, DVS_COMM_ID_BYTES);
  const int r = g_rccl.CommInitRank(&c->comm, world, u, rank);
  if (r != kNcclSuccess) {
    set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    delete c;
    return DVS_ERR_HIP;
  }
  *out = c;
  return DVS_OK;
}

dvs_status dvs_comm_create_host(int32_t rank, int32_t world, dvs_host_all_gather_fn all_gather, void* user, dvs_comm** out) {
  DVS_ARG(out && all_gather && world >= 1 && rank >= 0 && rank < world);
  *out = nullptr;
  dvs_comm* c = new (std::nothrow) dvs_comm();
  if (!c) { set_error("out of host memory"); return DVS_ERR_HIP; }
  c->device = -1; c->rank = rank; c->world = world; c->host_gather = all_gather; c->host_user = user;
  *out = c;
  return DVS_OK;
}

int32_t dvs_comm_is_host(const dvs_comm* c) { return c && c->host_gather ? 1 : 0; }

dvs_status dvs_comm_reset_sequence(dvs_comm* c) {
  DVS_ARG(c);
  // the next dvs_exchange_boundary is a FIRST call again: rank 0 gets no predecessor.  The caller has drained its streams (a reader of
  // the previous gathers may not be pending); the buffers and their rotation stay.
  c->calls = 0;
  c->has_last = false;
  return DVS_OK;
}

dvs_status dvs_comm_create_loopback(int32_t device, int32_t world, dvs_comm** out) {
  DVS_ARG(out && world >= 1 && world <= DVS_COMM_MAX_LOOPBACK);
  for (int r = 0; r < world; r++) out[r] = nullptr;
  DVS_TRY(check_device(device));
  LoopGroup* G = new (std::nothrow) LoopGroup();
  if (!G) { set_error("out of host memory"); return DVS_ERR_HIP; }
  G->world = world; G->device = device;
  bool ok = true;
  for (int r = 0; r < world; r++)
    ok = ok && hipEventCreateWithFlags(&G->ev_sent[r], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&G->ev_pulled[r], hipEventDisableTiming) == hipSuccess;
  for (int r = 0; ok && r < world; r++) {
    dvs_comm* c = new (std::nothrow) dvs_comm();
    if (!c) { ok = false; break; }
    c->device = device; c->rank = r; c->world = world; c->loop = G;
    G->refs++;
    out[r] = c;
  }
  if (!ok) {
    for (int r = 0; r < world; r++) { delete out[r]; out[r] = nullptr; }
    for (int r = 0; r < world; r++) { if (G->ev_sent[r]) (void)hipEventDestroy(G->ev_sent[r]); if (G->ev_pulled[r]) (void)hipEventDestroy(G->ev_pulled[r]); }
    delete G;
    set_error("loopback communicator: event creation failed");
    return DVS_ERR_HIP;
  }
  return DVS_OK;
}

void dvs_comm_destroy(dvs_comm* c) {
  if (!c) return;
  if (c->host_gather) {
    for (uint8_t* p : c->gather) free(p);
    delete c;
    return;
  }
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  if (c->loop) {
    LoopGroup* G = c->loop;
    bool last;
    { std::lock_guard<std::mutex> lk(G->mu); G->broken = true; last = --G->refs == 0; }   // a group that lost a rank cannot gather any more
    G->cv.notify_all();
    if (last) {
      for (int r = 0; r < G->world; r++) { (void)hipEventDestroy(G->ev_sent[r]); (void)hipEventDestroy(G->ev_pulled[r]); }
      delete G;
    }
  }
  if (c->comm) (void)g_rccl.CommDestroy(c->comm);
  if (c->ev_last) (void)hipEventDestroy(c->ev_last);
  for (uint8_t* p : c->gather) if (p) (void)hipFree(p);
  delete c;
}

int32_t dvs_comm_rank(const dvs_comm* c) { return c ? c->rank : -1; }
int32_t dvs_comm_world(const dvs_comm* c) { return c ? c->world : 0; }
int32_t dvs_comm_rccl_version(void) {
  int v = 0;
  if (load_rccl() != DVS_OK || !g_rccl.GetVersion || g_rccl.GetVersion(&v) != kNcclSuccess) return 0;
  return v;
}

dvs_status dvs_comm_all_gather(dvs_comm* c, void* stream, const void* d_send, void* d_recv, size_t bytes_per_rank) {
  DVS_ARG(c && d_send && d_recv);
  if (bytes_per_rank == 0) return DVS_OK;
  if (c->host_gather) return comm_all_gather(c, d_send, d_recv, bytes_per_rank, nullptr);
  DVS_HIP(hipSetDevice(c->device));
  if (c->loop) DVS_TRY(loop_before_send(c, (hipStream_t)stream));   // (the caller's send block was written before this call: see the header)
  return comm_all_gather(c, d_send, d_recv, bytes_per_rank, (hipStream_t)stream);
}

dvs_status dvs_exchange_boundary(dvs_comm* c, void* stream, const uint8_t* d_desc_last, const int32_t* d_n_last, int32_t cap,
                                 const uint8_t** d_prev_desc, const int32_t** d_prev_n) {
  DVS_ARG(c && d_desc_last && d_n_last && cap > 0 && d_prev_desc && d_prev_n);
  const bool host = c->host_gather != nullptr;
  DVS_ARG(host || ((uintptr_t)d_desc_last) % 16 == 0);
  if (!host) DVS_HIP(hipSetDevice(c->device));
  const size_t blk = dvs_boundary_block_bytes(cap);
  if (c->block != blk) {  // (re)allocate the three gather buffers once per capacity: nothing is allocated per step
    if (host) {
      for (uint8_t*& p : c->gather) { free(p); p = nullptr; }
      for (uint8_t*& p : c->gather)
        if (!(p = (uint8_t*)calloc(blk, (size_t)c->world))) { set_error("out of host memory (%zu bytes of gather buffer)", blk * (size_t)c->world); return DVS_ERR_HIP; }
    } else {
      DVS_HIP(hipDeviceSynchronize());
      for (uint8_t*& p : c->gather) { if (p) DVS_HIP(hipFree(p)); p = nullptr; }
      for (uint8_t*& p : c->gather) DVS_HIP(hipMalloc((void**)&p, blk * (size_t)c->world));
    }
    c->block = blk; c->calls = 0; c->turn = 0;
  }
  uint8_t* g = c->gather[c->turn];
  const uint8_t* gprev = c->calls > 0 ? c->gather[(c->turn + 2) % 3] : nullptr;
  c->turn = (c->turn + 1) % 3;
  c->calls++;
  hipStream_t st = (hipStream_t)stream;
  if (!host && c->has_last && c->last_stream != st) {
    if (c->multi_stream) DVS_HIP(hipStreamWaitEvent(st, c->ev_last, 0));
    else {
      DVS_HIP(hipStreamSynchronize(c->last_stream));
      DVS_HIP(hipEventCreateWithFlags(&c->ev_last, hipEventDisableTiming));
      c->multi_stream = true;
    }
  }
  if (c->loop) DVS_TRY(loop_before_send(c, st));
  uint8_t* mine = g + (size_t)c->rank * blk;
  const int rows16 = cap * 2, blk16 = (int)(blk / 16);
  if (host) {   // the block k_pack_boundary writes: descriptor rows, the count, zero padding
    memcpy(mine, d_desc_last, (size_t)cap * 32);
    memset(mine + (size_t)cap * 32, 0, blk - (size_t)cap * 32);
    memcpy(mine + (size_t)cap * 32, d_n_last, 4);
  } else {
    hipLaunchKernelGGL(k_pack_boundary, dim3((blk16 + 255) / 256), dim3(256), 0, st, (const uint4*)d_desc_last, d_n_last, (uint4*)mine, rows16, blk16);
    DVS_HIP(hipGetLastError());
  }
  // in place: this rank's block already sits at its slot of the receive buffer
  DVS_TRY(comm_all_gather(c, mine, g, blk, st));
  if (!host && c->multi_stream) DVS_HIP(hipEventRecord(c->ev_last, st));
  c->last_stream = st; c->has_last = true;
  // predecessor of this rank's FIRST frame of the batch: the previous rank's last frame of the SAME batch — or, for rank 0, the
  // last rank's last frame of the PREVIOUS batch (the previous call's gather; nothing on the first call)
  const uint8_t* pb = c->rank > 0 ? g + (size_t)(c->rank - 1) * blk : (gprev ? gprev + (size_t)(c->world - 1) * blk : nullptr);
  *d_prev_desc = pb;
  *d_prev_n = pb ? (const int32_t*)(pb + (size_t)cap * 32) : nullptr;
  return DVS_OK;
}

}  // extern "C"
