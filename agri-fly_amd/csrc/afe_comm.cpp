// afe_comm.cpp -- the shared-world exchange of the C ABI: positions of every
// shard, 12 B per vehicle, gathered onto every GPU at query cadence.
//
// Two host shapes, both through the same engine objects:
//   * one process per GPU (the layout bench.py and torch.distributed use): afe_comm wraps an RCCL
//     communicator; afe_gather_positions = pack + ncclAllGather on the engine's stream.  RCCL is
//     loaded at run time (dlopen "librccl.so.1"), so a host that never shards needs no RCCL.
//   * one process driving several GPUs -- the reference's own process model, a single loop over a
//     std::vector of vehicles (AIFS_ROS/hiperlab_rostools/src/Simulator/main.cpp:323-325):
//     afe_group owns one engine per device and exchanges with peer-to-peer copies over xGMI
//     (direct, one hop per pair: the all-gather is 1.5 MB per link at 1 M vehicles on 8 GPUs).
//     The same device may be listed more than once ("logical shards"), which is how the sharding
//     is tested bit for bit on a single GPU.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "afe_host.h"
#include "afe_render.h"   // engine_stream_device

namespace afe {
// afe_engine.cpp
int engine_pack_to_scratch(afe_engine *e, float **scratch);   // positions -> planar fp32 [3][n] scratch, on the engine's stream
void engine_shard(const afe_engine *e, int64_t *first_global, int64_t *n);
}  // namespace afe
using namespace afe;

namespace {

struct Rccl {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string why;
};

Rccl *rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r.handle ? &r : nullptr;
  tried = true;
  // by SONAME: a process that already loaded RCCL (torch ships its own copy) gets that one back
  void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!h) { r.why = std::string("RCCL not loadable: ") + dlerror(); return nullptr; }
  bool ok = true;
  auto sym = [&](const char *name) { void *p = dlsym(h, name); if (!p) { ok = false; r.why = std::string("RCCL lacks ") + name; } return p; };
  r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
  r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
  r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
  r.CommUserRank = (decltype(r.CommUserRank))sym("ncclCommUserRank");
  r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
  r.Broadcast = (decltype(r.Broadcast))sym("ncclBroadcast");
  r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
  r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
  r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
  if (!ok) { dlclose(h); return nullptr; }
  r.handle = h;
  return &r;
}

}  // namespace

struct afe_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, n_ranks = 1, device = 0;
  std::string err;
};

static_assert(AFE_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "afe_comm id size");

extern "C" int afe_comm_unique_id(uint8_t id[AFE_COMM_ID_BYTES]) {
  if (!id) return AFE_ERR_INVALID_ARG;
  Rccl *r = rccl();
  if (!r) return AFE_ERR_COMM;
  ncclUniqueId u;
  if (r->GetUniqueId(&u) != ncclSuccess) return AFE_ERR_COMM;
  std::memcpy(id, u.internal, AFE_COMM_ID_BYTES);
  return AFE_OK;
}

extern "C" int afe_comm_create(afe_comm **out, const uint8_t id[AFE_COMM_ID_BYTES], int rank, int n_ranks, int device) {
  if (!out || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return AFE_ERR_INVALID_ARG;
  *out = nullptr;
  Rccl *r = rccl();
  if (!r) return AFE_ERR_COMM;
  if (device < 0 && hipGetDevice(&device) != hipSuccess) return AFE_ERR_NO_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return AFE_ERR_NO_DEVICE;
  ncclUniqueId u;
  std::memcpy(u.internal, id, AFE_COMM_ID_BYTES);
  afe_comm *c = new afe_comm();
  c->rank = rank; c->n_ranks = n_ranks; c->device = device;
  const ncclResult_t rc = r->CommInitRank(&c->comm, n_ranks, u, rank);
  if (rc != ncclSuccess) {
    std::fprintf(stderr, "agrifly_engine: ncclCommInitRank failed: %s\n", r->GetErrorString(rc));
    delete c;
    return AFE_ERR_COMM;
  }
  *out = c;
  return AFE_OK;
}

extern "C" int afe_comm_info(const afe_comm *c, int *rank, int *n_ranks) {
  if (!c) return AFE_ERR_INVALID_ARG;
  Rccl *r = rccl();
  if (!r) return AFE_ERR_COMM;
  int cnt = 0, me = 0;   // as the communicator itself reports them
  if (r->CommCount(c->comm, &cnt) != ncclSuccess || r->CommUserRank(c->comm, &me) != ncclSuccess) return AFE_ERR_COMM;
  if (rank) *rank = me;
  if (n_ranks) *n_ranks = cnt;
  return AFE_OK;
}

extern "C" int afe_comm_destroy(afe_comm *c) {
  if (!c) return AFE_ERR_INVALID_ARG;
  Rccl *r = rccl();
  if (r && c->comm) { (void)hipSetDevice(c->device); (void)r->CommDestroy(c->comm); }
  delete c;
  return AFE_OK;
}

extern "C" const char *afe_comm_last_error(const afe_comm *c) {
  if (!c) { Rccl *r = rccl(); (void)r; static std::string s; s = "no communicator"; return s.c_str(); }
  return c->err.c_str();
}

// The exchange itself, whatever moves the bytes: which rank's block goes where in the gathered buffer (counts, offsets,
// one all-gather per component when the shards are equal, one broadcast per rank and component when they are not).
// Pure host logic over opaque buffers -- afe_gather_positions runs it with RCCL on device memory, a host with its own
// transport (MPI, torch.distributed) runs it with that, and the CPU test suite runs it over gloo on host memory:
// the same code with only the transport swapped.
extern "C" int afe_gather_exchange(const afe_gather_transport *t, int rank, int n_ranks, const int64_t *counts, int64_t n_local,
                                   const float *packed_xyz, float *xyz_all) {
  if (!t || !t->all_gather || !t->broadcast || n_ranks < 1 || rank < 0 || rank >= n_ranks || n_local <= 0 || !packed_xyz || !xyz_all)
    return AFE_ERR_INVALID_ARG;
  bool equal = true;
  int64_t n_all = n_local * n_ranks;
  std::vector<int64_t> firsts((size_t)n_ranks, 0);
  if (counts) {
    if (counts[rank] != n_local) return AFE_ERR_INVALID_ARG;
    n_all = 0;
    for (int k = 0; k < n_ranks; k++) {
      if (counts[k] <= 0) return AFE_ERR_INVALID_ARG;
      firsts[(size_t)k] = n_all;
      n_all += counts[k];
      equal = equal && counts[k] == n_local;
    }
  }
  int rc = t->group_start ? t->group_start(t->ctx) : 0;
  for (int comp = 0; comp < 3 && rc == 0; comp++) {
    if (equal) {
      rc = t->all_gather(t->ctx, packed_xyz + comp * n_local, xyz_all + comp * n_all, n_local);
    } else {
      for (int k = 0; k < n_ranks && rc == 0; k++)
        rc = t->broadcast(t->ctx, packed_xyz + comp * n_local, xyz_all + comp * n_all + firsts[(size_t)k], counts[k], k);
    }
  }
  const int erc = t->group_end ? t->group_end(t->ctx) : 0;
  if (rc == 0) rc = erc;
  return rc == 0 ? AFE_OK : AFE_ERR_COMM;
}

namespace {
struct RcclCtx { Rccl *r; afe_comm *c; hipStream_t st; ncclResult_t last; };
int rccl_all_gather(void *ctx, const float *send, float *recv, int64_t count) {
  RcclCtx *x = (RcclCtx *)ctx;
  x->last = x->r->AllGather(send, recv, (size_t)count, ncclFloat, x->c->comm, x->st);
  return x->last == ncclSuccess ? 0 : 1;
}
int rccl_broadcast(void *ctx, const float *send, float *recv, int64_t count, int root) {
  RcclCtx *x = (RcclCtx *)ctx;
  x->last = x->r->Broadcast(send, recv, (size_t)count, ncclFloat, root, x->c->comm, x->st);
  return x->last == ncclSuccess ? 0 : 1;
}
int rccl_group_start(void *ctx) { RcclCtx *x = (RcclCtx *)ctx; x->last = x->r->GroupStart(); return x->last == ncclSuccess ? 0 : 1; }
int rccl_group_end(void *ctx) {
  RcclCtx *x = (RcclCtx *)ctx;
  const ncclResult_t e = x->r->GroupEnd();
  if (e != ncclSuccess) x->last = e;
  return e == ncclSuccess ? 0 : 1;
}
}  // namespace

// counts[r] = vehicles of rank r (NULL: every rank holds as many as this one).  Output: planar
// fp32 [3][n_all] on the device, vehicles in global order.
extern "C" int afe_gather_positions(afe_engine *e, afe_comm *c, const int64_t *counts, float *dev_xyz_all) {
  if (!e || !c || !dev_xyz_all) return AFE_ERR_INVALID_ARG;
  Rccl *r = rccl();
  if (!r) return AFE_ERR_COMM;
  void *stream_v = nullptr;
  int device = 0;
  engine_stream_device(e, &stream_v, &device);
  if (device != c->device) { c->err = "engine and communicator live on different devices"; return AFE_ERR_INVALID_ARG; }
  int64_t first = 0, n = 0;
  engine_shard(e, &first, &n);
  if (counts && counts[c->rank] != n) { c->err = "counts[rank] differs from the engine's vehicle count"; return AFE_ERR_INVALID_ARG; }
  // one rank: the gathered buffer IS the packed block -- straight into it, no copy through the communicator (three
  // device-to-device copies of 4 MB at 2^20 vehicles, 16 us of every query cycle)
  if (c->n_ranks == 1) return afe_pack_positions(e, dev_xyz_all);
  float *scratch = nullptr;
  int rc = engine_pack_to_scratch(e, &scratch);
  if (rc) return rc;
  RcclCtx x = {r, c, (hipStream_t)stream_v, ncclSuccess};
  const afe_gather_transport t = {&x, rccl_all_gather, rccl_broadcast, rccl_group_start, rccl_group_end};
  rc = afe_gather_exchange(&t, c->rank, c->n_ranks, counts, n, scratch, dev_xyz_all);
  if (rc == AFE_ERR_COMM) c->err = std::string("RCCL all-gather: ") + r->GetErrorString(x.last);
  else if (rc) c->err = "bad counts";
  return rc;
}

// ---------------------------------------------------------------------------
// one process, several devices

struct afe_group {
  std::vector<afe_engine *> engines;
  std::vector<int> devices;
  std::vector<int64_t> first, count;
  std::vector<float *> all_xyz;       // per shard: planar [3][n_total] on that shard's device
  std::vector<hipEvent_t> packed;     // per shard: its scratch is ready
  std::vector<hipEvent_t> pulled;     // per shard: it has pulled every block it needs
  int64_t n_total = 0;
  bool peer_ok = true;                // every pair of distinct devices reads the other's memory directly
  bool staged = false;                // the gather copies row by row with hipMemcpyPeerAsync (no peer access, or asked for)
  std::string err;
};

extern "C" int afe_group_create(afe_group **out, int64_t n_vehicles, int precision, const int *devices, int n_devices) {
  if (!out || n_vehicles <= 0 || !devices || n_devices < 1 || n_devices > n_vehicles) return AFE_ERR_INVALID_ARG;
  *out = nullptr;
  afe_group *g = new afe_group();
  g->n_total = n_vehicles;
  const int64_t base = n_vehicles / n_devices, rem = n_vehicles % n_devices;   // contiguous blocks, sizes differ by <= 1
  int64_t first = 0;
  for (int k = 0; k < n_devices; k++) {
    const int64_t cnt = base + (k < rem ? 1 : 0);
    afe_engine *e = nullptr;
    const int rc = afe_create(&e, cnt, precision, devices[k], first);
    if (rc != AFE_OK) { afe_group_destroy(g); return rc; }
    // One host thread launches for every device of the group: a second launch per shard and step (the automatic
    // split stepping of large engines) would make that thread, not the devices, the limit.  The host can still ask
    // for it per shard (afe_group_shard + afe_set_split_stepping).
    if (n_devices > 1 && std::getenv("AFE_FORCE_SPLIT") == nullptr) (void)afe_set_split_stepping(e, 1);
    g->engines.push_back(e);
    g->devices.push_back(devices[k]);
    g->first.push_back(first);
    g->count.push_back(cnt);
    g->all_xyz.push_back(nullptr);
    hipEvent_t ev = nullptr;
    if (hipSetDevice(devices[k]) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { afe_group_destroy(g); return AFE_ERR_HIP; }
    g->packed.push_back(ev);
    hipEvent_t ev2 = nullptr;
    if (hipEventCreateWithFlags(&ev2, hipEventDisableTiming) != hipSuccess) { afe_group_destroy(g); return AFE_ERR_HIP; }
    g->pulled.push_back(ev2);
    first += cnt;
  }
  // Every device reads every other one's memory directly where it can (xGMI): afe_group_gather_positions then issues one
  // strided device-to-device copy per pair of shards.  A group with a pair that lacks peer access (IOMMU, a VM, a
  // restricted container) is still served, by the STAGED path: one hipMemcpyPeerAsync per row of a block, which the
  // runtime carries through host memory where the devices cannot reach each other -- slower; said once, with the pair,
  // and remembered (afe_group_peer_access).  afe_group_set_staged_copies selects that path by hand.  Not verified on a
  // machine whose devices really lack peer access (none available): the staged path is exercised on peers and on logical
  // shards of one device (tests/test_gpu_sharedworld.py).
  for (int a = 0; a < n_devices; a++)
    for (int b = 0; b < n_devices; b++) {
      if (devices[a] == devices[b]) continue;
      int can = 0;
      hipError_t perr = hipDeviceCanAccessPeer(&can, devices[a], devices[b]);
      if (perr == hipSuccess && can) {
        (void)hipSetDevice(devices[a]);
        perr = hipDeviceEnablePeerAccess(devices[b], 0);
        if (perr == hipErrorPeerAccessAlreadyEnabled) perr = hipSuccess;
      } else if (perr == hipSuccess) {
        perr = hipErrorPeerAccessUnsupported;
      }
      (void)hipGetLastError();
      if (perr != hipSuccess) {
        if (g->peer_ok)
          std::fprintf(stderr, "agrifly_engine: afe_group_create: device %d cannot access device %d's memory directly (%s); the group's position "
                               "gather uses staged copies (hipMemcpyPeerAsync)\n", devices[a], devices[b], hipGetErrorString(perr));
        g->peer_ok = false;
      }
    }
  g->staged = !g->peer_ok;
  *out = g;
  return AFE_OK;
}

extern "C" int afe_group_set_staged_copies(afe_group *g, int staged) {
  if (!g) return AFE_ERR_INVALID_ARG;
  if (!staged && !g->peer_ok) { g->err = "a pair of the group's devices lacks peer access: the direct copies are not available"; return AFE_ERR_INVALID_ARG; }
  g->staged = staged != 0;
  return AFE_OK;
}

extern "C" int afe_group_destroy(afe_group *g) {
  if (!g) return AFE_ERR_INVALID_ARG;
  for (size_t k = 0; k < g->engines.size(); k++) {
    (void)hipSetDevice(g->devices[k]);
    if (g->all_xyz[k]) (void)hipFree(g->all_xyz[k]);
    if (k < g->packed.size() && g->packed[k]) (void)hipEventDestroy(g->packed[k]);
    if (k < g->pulled.size() && g->pulled[k]) (void)hipEventDestroy(g->pulled[k]);
    if (g->engines[k]) afe_destroy(g->engines[k]);
  }
  delete g;
  return AFE_OK;
}

extern "C" int afe_group_size(const afe_group *g, int *n_shards, int64_t *n_vehicles) {
  if (!g) return AFE_ERR_INVALID_ARG;
  if (n_shards) *n_shards = (int)g->engines.size();
  if (n_vehicles) *n_vehicles = g->n_total;
  return AFE_OK;
}

extern "C" int afe_group_shard(afe_group *g, int shard, afe_engine **engine, int64_t *first, int64_t *count) {
  if (!g || shard < 0 || shard >= (int)g->engines.size()) return AFE_ERR_INVALID_ARG;
  if (engine) *engine = g->engines[(size_t)shard];
  if (first) *first = g->first[(size_t)shard];
  if (count) *count = g->count[(size_t)shard];
  return AFE_OK;
}

extern "C" const char *afe_group_last_error(const afe_group *g) { return g ? g->err.c_str() : "null group"; }

extern "C" int afe_group_peer_access(const afe_group *g, int *all_pairs) {
  if (!g || !all_pairs) return AFE_ERR_INVALID_ARG;
  *all_pairs = g->peer_ok ? 1 : 0;
  return AFE_OK;
}

// for (v : vehicles) v->Run(); timer.Advance(dt) over every shard: the launches go to each device's
// own stream and run concurrently; nothing is exchanged (the step reads no other vehicle)
extern "C" int afe_group_step(afe_group *g, uint64_t dt_us, int n_steps) {
  if (!g) return AFE_ERR_INVALID_ARG;
  for (size_t k = 0; k < g->engines.size(); k++) {
    const int rc = afe_step(g->engines[k], dt_us, n_steps);
    if (rc) { g->err = std::string("shard ") + std::to_string(k) + ": " + afe_last_error(g->engines[k]); return rc; }
  }
  return AFE_OK;
}

extern "C" int afe_group_sync(afe_group *g) {
  if (!g) return AFE_ERR_INVALID_ARG;
  for (size_t k = 0; k < g->engines.size(); k++) {
    const int rc = afe_sync(g->engines[k]);
    if (rc) { g->err = afe_last_error(g->engines[k]); return rc; }
  }
  return AFE_OK;
}

// every shard's positions onto every shard's device: planar fp32 [3][n_total] each, global order.
// dev_xyz_all_out[k] (optional) receives shard k's buffer (owned by the group).
extern "C" int afe_group_gather_positions(afe_group *g, float **dev_xyz_all_out) {
  if (!g) return AFE_ERR_INVALID_ARG;
  const size_t G = g->engines.size();
  std::vector<float *> scratch(G, nullptr);
  std::vector<hipStream_t> streams(G, nullptr);
  for (size_t k = 0; k < G; k++) {
    void *sv = nullptr;
    int dev = 0;
    engine_stream_device(g->engines[k], &sv, &dev);
    streams[k] = (hipStream_t)sv;
    if (hipSetDevice(dev) != hipSuccess) { g->err = "hipSetDevice failed"; return AFE_ERR_HIP; }
    if (!g->all_xyz[k] && hipMalloc((void **)&g->all_xyz[k], (size_t)g->n_total * 3 * sizeof(float)) != hipSuccess) {
      g->err = "hipMalloc of the gathered-position buffer failed";
      return AFE_ERR_HIP;
    }
    const int rc = engine_pack_to_scratch(g->engines[k], &scratch[k]);
    if (rc) { g->err = afe_last_error(g->engines[k]); return rc; }
    if (hipEventRecord(g->packed[k], streams[k]) != hipSuccess) { g->err = "hipEventRecord (packed) failed"; return AFE_ERR_HIP; }
  }
  // destination d pulls source s's block: one strided copy per pair (3 rows of count_s floats),
  // ordered on d's stream behind s's pack
  for (size_t d = 0; d < G; d++) {
    if (hipSetDevice(g->devices[d]) != hipSuccess) { g->err = "hipSetDevice failed"; return AFE_ERR_HIP; }
    for (size_t off = 0; off < G; off++) {
      const size_t s = (d + off) % G;   // start with the own block, then walk the ring so the links load evenly
      if (s != d && hipStreamWaitEvent(streams[d], g->packed[s], 0) != hipSuccess) { g->err = "hipStreamWaitEvent (packed) failed"; return AFE_ERR_HIP; }
      hipError_t err = hipSuccess;
      if (g->staged && s != d) {
        // no direct access between the two devices (or the host asked): the runtime's peer copy, which stages through
        // host memory where it has to; contiguous rows, so one call per component
        for (int c = 0; c < 3 && err == hipSuccess; c++)
          err = hipMemcpyPeerAsync(g->all_xyz[d] + (size_t)c * g->n_total + g->first[s], g->devices[d], scratch[s] + (size_t)c * g->count[s], g->devices[s],
                                   (size_t)g->count[s] * 4, streams[d]);
      } else {
        err = hipMemcpy2DAsync(g->all_xyz[d] + g->first[s], (size_t)g->n_total * 4, scratch[s], (size_t)g->count[s] * 4,
                               (size_t)g->count[s] * 4, 3, hipMemcpyDeviceToDevice, streams[d]);
      }
      if (err != hipSuccess) { g->err = std::string(g->staged && s != d ? "staged peer copy: " : "peer copy: ") + hipGetErrorString(err); return AFE_ERR_HIP; }
    }
    if (dev_xyz_all_out) dev_xyz_all_out[d] = g->all_xyz[d];
  }
  // a shard's scratch must not be repacked before every reader has pulled it: readers' streams are
  // joined back into the owner's stream
  for (size_t d = 0; d < G; d++) {
    if (hipSetDevice(g->devices[d]) != hipSuccess) { g->err = "hipSetDevice failed"; return AFE_ERR_HIP; }
    if (hipEventRecord(g->pulled[d], streams[d]) != hipSuccess) { g->err = "hipEventRecord (pulled) failed"; return AFE_ERR_HIP; }
  }
  for (size_t s = 0; s < G; s++)
    for (size_t d = 0; d < G; d++)
      if (d != s && hipStreamWaitEvent(streams[s], g->pulled[d], 0) != hipSuccess) { g->err = "hipStreamWaitEvent (pulled) failed"; return AFE_ERR_HIP; }
  return AFE_OK;
}
