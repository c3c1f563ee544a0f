// afe_aql.cpp -- the engine's own user-mode AQL queue for the resident step grid (see afe_aql.h).
// Everything here goes through the ROCr instance HIP itself runs on: the library is found among the objects already
// loaded into the process (never a second copy), its entry points are taken with dlsym, the kernel's descriptor is
// found in the executable HIP loaded.
#include "afe_aql.h"
#include "afe_host.h"   // afe_dev_env

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <hsa/hsa_ven_amd_loader.h>
#include <link.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

namespace afe {
namespace {

#define AFE_HSA_FUNCS(X)                                                                                               \
  X(hsa_init) X(hsa_shut_down) X(hsa_status_string) X(hsa_iterate_agents) X(hsa_agent_get_info) X(hsa_queue_create)    \
  X(hsa_queue_destroy) X(hsa_signal_create) X(hsa_signal_destroy) X(hsa_signal_store_relaxed)                          \
  X(hsa_signal_store_screlease) X(hsa_signal_load_scacquire) X(hsa_signal_wait_scacquire)                              \
  X(hsa_queue_load_read_index_scacquire) X(hsa_queue_load_write_index_relaxed) X(hsa_queue_store_write_index_screlease) \
  X(hsa_system_get_major_extension_table) X(hsa_system_get_info) X(hsa_executable_get_symbol_by_name)                  \
  X(hsa_executable_symbol_get_info) X(hsa_amd_agent_iterate_memory_pools) X(hsa_amd_memory_pool_get_info)              \
  X(hsa_amd_memory_pool_allocate) X(hsa_amd_memory_pool_free) X(hsa_amd_agents_allow_access) X(hsa_amd_agent_memory_pool_get_info)                           \
  X(hsa_amd_profiling_set_profiler_enabled) X(hsa_amd_profiling_get_dispatch_time)

struct Api {
#define X(f) decltype(&::f) f = nullptr;
  AFE_HSA_FUNCS(X)
#undef X
  void *lib = nullptr;
  bool ok = false;
  std::string why;
};

int find_hsa_cb(struct dl_phdr_info *info, size_t, void *data) {
  if (info->dlpi_name && std::strstr(info->dlpi_name, "libhsa-runtime64")) {
    *static_cast<std::string *>(data) = info->dlpi_name;
    return 1;
  }
  return 0;
}

Api &api() {
  static Api a;
  static std::once_flag once;
  std::call_once(once, [] {
    std::string path;
    dl_iterate_phdr(find_hsa_cb, &path);
    if (path.empty()) { a.why = "libhsa-runtime64 is not loaded in this process"; return; }
    a.lib = dlopen(path.c_str(), RTLD_NOW | RTLD_NOLOAD);
    if (!a.lib) { a.why = "dlopen(RTLD_NOLOAD) of " + path + " failed"; return; }
#define X(f)                                                             \
  a.f = reinterpret_cast<decltype(&::f)>(dlsym(a.lib, #f));              \
  if (!a.f) { a.why = std::string("symbol not found: ") + #f; return; }
    AFE_HSA_FUNCS(X)
#undef X
    a.ok = true;
  });
  return a;
}

std::string hsa_err(const Api &a, hsa_status_t st, const char *what) {
  const char *s = nullptr;
  if (a.hsa_status_string) (void)a.hsa_status_string(st, &s);
  return std::string(what) + ": " + (s ? s : "unknown HSA status");
}

// Which HSA agent is HIP device `hip_device`?  HIP numbers the devices it was allowed to see (HIP_VISIBLE_DEVICES,
// ROCR_VISIBLE_DEVICES re-order and filter), HSA enumerates agents: the index proves nothing.  Two identities HIP reports
// for its device are the agent's own: the PCI address (domain : bus : device) and the 16-character unique id (HIP copies
// it from HSA_AMD_AGENT_INFO_UUID's "GPU-xxxxxxxxxxxxxxxx").  An agent must match the PCI address; where several do
// (never seen) or none does (a hypervisor that renumbers the bus for one of the two views), the unique id decides.  Only
// when both fail and the process sees exactly one GPU agent is that one taken -- and AFE_PERSIST_DEBUG says which rule
// chose (tests/test_gpu_resident_sync.py holds the first rule on the boxes the suite runs on).
struct FindAgent {
  const Api *a;
  uint32_t domain, bus, dev;
  char uuid[17];
  hsa_agent_t by_pci{}, by_uuid{}, cpu{}, only_gpu{};
  int n_pci = 0, n_uuid = 0, n_gpus = 0;
  bool have_cpu = false;
  std::string seen;      // what the GPU agents looked like (for the failure message)
};

hsa_status_t agent_cb(hsa_agent_t agent, void *data) {
  FindAgent *f = static_cast<FindAgent *>(data);
  hsa_device_type_t type;
  if (f->a->hsa_agent_get_info(agent, HSA_AGENT_INFO_DEVICE, &type) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
  if (type == HSA_DEVICE_TYPE_CPU && !f->have_cpu) { f->cpu = agent; f->have_cpu = true; }
  if (type == HSA_DEVICE_TYPE_GPU) {
    f->n_gpus++;
    f->only_gpu = agent;
    uint32_t bdf = 0, domain = 0;
    char uuid[64] = {0};
    (void)f->a->hsa_agent_get_info(agent, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf);
    (void)f->a->hsa_agent_get_info(agent, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &domain);
    (void)f->a->hsa_agent_get_info(agent, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_UUID, uuid);
    if (((bdf >> 8) & 0xffu) == f->bus && ((bdf >> 3) & 0x1fu) == f->dev && domain == f->domain) { if (f->n_pci++ == 0) f->by_pci = agent; }
    if (f->uuid[0] && std::strncmp(uuid, "GPU-", 4) == 0 && std::strncmp(uuid + 4, f->uuid, 16) == 0) { if (f->n_uuid++ == 0) f->by_uuid = agent; }
    char line[128];
    std::snprintf(line, sizeof(line), " [%04x:%02x:%02x %s]", domain, (bdf >> 8) & 0xffu, (bdf >> 3) & 0x1fu, uuid);
    f->seen += line;
  }
  return HSA_STATUS_SUCCESS;
}

struct FindPool {
  const Api *a;
  hsa_amd_memory_pool_t pool{};
  bool found = false;
};

hsa_status_t pool_cb(hsa_amd_memory_pool_t pool, void *data) {
  FindPool *f = static_cast<FindPool *>(data);
  hsa_amd_segment_t seg;
  uint32_t flags = 0;
  bool alloc = false;
  if (f->a->hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg) != HSA_STATUS_SUCCESS || seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
  (void)f->a->hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
  (void)f->a->hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
  if (alloc && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT)) { f->pool = pool; f->found = true; return HSA_STATUS_INFO_BREAK; }
  return HSA_STATUS_SUCCESS;
}

struct FindDevPool {
  const Api *a;
  hsa_agent_t cpu;
  hsa_amd_memory_pool_t pool{};
  bool found = false;
};

// a pool of the GPU's own memory that is fine-grained and that the host may be given access to (large-BAR systems)
hsa_status_t dev_pool_cb(hsa_amd_memory_pool_t pool, void *data) {
  FindDevPool *f = static_cast<FindDevPool *>(data);
  hsa_amd_segment_t seg;
  uint32_t flags = 0;
  bool alloc = false;
  if (f->a->hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg) != HSA_STATUS_SUCCESS || seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
  (void)f->a->hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
  (void)f->a->hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
  if (!alloc || !(flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_FINE_GRAINED)) return HSA_STATUS_SUCCESS;
  hsa_amd_memory_pool_access_t acc = HSA_AMD_MEMORY_POOL_ACCESS_NEVER_ALLOWED;
  if (f->a->hsa_amd_agent_memory_pool_get_info(f->cpu, pool, HSA_AMD_AGENT_MEMORY_POOL_INFO_ACCESS, &acc) != HSA_STATUS_SUCCESS ||
      acc == HSA_AMD_MEMORY_POOL_ACCESS_NEVER_ALLOWED) return HSA_STATUS_SUCCESS;
  f->pool = pool; f->found = true;
  return HSA_STATUS_INFO_BREAK;
}

struct FindSymbol {
  const Api *a;
  hsa_agent_t agent;
  const char *name_kd;
  AqlKernel k;
  bool found = false;
};

hsa_status_t exec_cb(hsa_executable_t exe, void *data) {
  FindSymbol *f = static_cast<FindSymbol *>(data);
  hsa_executable_symbol_t sym;
  if (f->a->hsa_executable_get_symbol_by_name(exe, f->name_kd, &f->agent, &sym) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
  hsa_symbol_kind_t kind;
  if (f->a->hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_TYPE, &kind) != HSA_STATUS_SUCCESS || kind != HSA_SYMBOL_KIND_KERNEL) return HSA_STATUS_SUCCESS;
  if (f->a->hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &f->k.object) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
  (void)f->a->hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &f->k.kernarg_bytes);
  (void)f->a->hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &f->k.group_bytes);
  (void)f->a->hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &f->k.private_bytes);
  f->found = f->k.object != 0;
  return f->found ? HSA_STATUS_INFO_BREAK : HSA_STATUS_SUCCESS;
}

}  // namespace

struct AqlQueue {
  int hip_device = 0;
  hsa_agent_t gpu{}, cpu{};
  hsa_queue_t *queue = nullptr;
  hsa_signal_t done{};
  bool have_signal = false;
  void *kernarg = nullptr;          // host kernarg pool (kept for devices without the device-side copy): two slots of KERNARG_SLOT bytes
  void *kernarg_dev = nullptr;      // the same two slots in DEVICE memory: what the packets point at (see aql_dispatch)
  bool kernarg_dev_mapped = false;  // ... which the host writes directly (fine-grained device memory through the BAR); else by hipMemcpy
  unsigned slot = 0;
  bool in_flight = false;
  bool initialised = false;         // hsa_init taken (to be given back)
  std::atomic<int> error{0};        // the queue's error callback
  uint64_t last_ns = 0;
  uint64_t ticks_per_s = 0;
};
static const size_t KERNARG_SLOT = 4096;

static void queue_error_cb(hsa_status_t status, hsa_queue_t *, void *data) {
  AqlQueue *q = static_cast<AqlQueue *>(data);
  if (q) q->error.store((int)status ? (int)status : -1);
}

AqlQueue *aql_open(int hip_device, std::string *why) {
  Api &a = api();
  if (!a.ok) { if (why) *why = a.why; return nullptr; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, hip_device) != hipSuccess) { if (why) *why = "hipGetDeviceProperties failed"; return nullptr; }
  hsa_status_t st = a.hsa_init();      // reference-counted: HIP holds the runtime open already
  if (st != HSA_STATUS_SUCCESS) { if (why) *why = hsa_err(a, st, "hsa_init"); return nullptr; }
  AqlQueue *q = new AqlQueue();
  q->hip_device = hip_device;
  q->initialised = true;
  auto bail = [&](const std::string &msg) -> AqlQueue * { if (why) *why = msg; aql_close(q); return nullptr; };
  FindAgent fa{};
  fa.a = &a; fa.domain = (uint32_t)prop.pciDomainID; fa.bus = (uint32_t)prop.pciBusID; fa.dev = (uint32_t)prop.pciDeviceID;
  std::memcpy(fa.uuid, prop.uuid.bytes, 16);
  fa.uuid[16] = 0;
  for (int i = 0; i < 16; i++) if (fa.uuid[i] < 0x20 || fa.uuid[i] > 0x7e) { fa.uuid[0] = 0; break; }     // (not the printable id: no such rule)
  st = a.hsa_iterate_agents(agent_cb, &fa);
  if (st != HSA_STATUS_SUCCESS && st != HSA_STATUS_INFO_BREAK) return bail(hsa_err(a, st, "hsa_iterate_agents"));
  const char *rule = nullptr;
  if (fa.n_pci == 1) { q->gpu = fa.by_pci; rule = "PCI address"; }
  else if (fa.n_uuid == 1) { q->gpu = fa.by_uuid; rule = "unique id"; }
  else if (fa.n_gpus == 1) { q->gpu = fa.only_gpu; rule = "being the only GPU agent"; }
  if (!rule || !fa.have_cpu) {
    char want[96];
    std::snprintf(want, sizeof(want), "%04x:%02x:%02x %s", fa.domain, fa.bus, fa.dev, fa.uuid);
    return bail(std::string("no HSA agent matches HIP device ") + std::to_string(hip_device) + " (" + want + "); GPU agents:" + fa.seen);
  }
  q->cpu = fa.cpu;
  if (std::getenv("AFE_PERSIST_DEBUG"))
    std::fprintf(stderr, "agrifly_engine: AQL queue: HIP device %d (%04x:%02x:%02x %s) is the HSA agent chosen by %s; %d GPU agent(s):%s\n", hip_device, fa.domain, fa.bus,
                 fa.dev, fa.uuid, rule, fa.n_gpus, fa.seen.c_str());
  (void)a.hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &q->ticks_per_s);
  st = afe_fault("queue_create") ? HSA_STATUS_ERROR_OUT_OF_RESOURCES
                                 : a.hsa_queue_create(q->gpu, 64, HSA_QUEUE_TYPE_SINGLE, queue_error_cb, q, UINT32_MAX, UINT32_MAX, &q->queue);
  if (st != HSA_STATUS_SUCCESS) { q->queue = nullptr; return bail(hsa_err(a, st, "hsa_queue_create")); }
  (void)a.hsa_amd_profiling_set_profiler_enabled(q->queue, 1);     // begin / end device timestamps on the completion signal
  st = a.hsa_signal_create(0, 0, nullptr, &q->done);
  if (st != HSA_STATUS_SUCCESS) return bail(hsa_err(a, st, "hsa_signal_create"));
  q->have_signal = true;
  FindPool fp{};
  fp.a = &a;
  st = a.hsa_amd_agent_iterate_memory_pools(q->cpu, pool_cb, &fp);
  if (!fp.found) return bail("no kernarg memory pool on the host agent");
  st = a.hsa_amd_memory_pool_allocate(fp.pool, 2 * KERNARG_SLOT, 0, &q->kernarg);
  if (st != HSA_STATUS_SUCCESS) { q->kernarg = nullptr; return bail(hsa_err(a, st, "hsa_amd_memory_pool_allocate (kernarg)")); }
  st = a.hsa_amd_agents_allow_access(1, &q->gpu, nullptr, q->kernarg);
  if (st != HSA_STATUS_SUCCESS) return bail(hsa_err(a, st, "hsa_amd_agents_allow_access (kernarg)"));
  // The arguments of a resident grid are not read once: 186 dwords do not all stay in scalar registers, and what the
  // scalar cache drops is fetched again from the kernel-argument segment inside the step loop.  In host memory that is a
  // PCIe round trip per miss -- measured: grids dispatched with host-side arguments step 2.5 % slower than the same kernel
  // launched by HIP (which keeps arguments in device memory), 11 % in an engine's first 100 ms.  So: device memory.
  static const bool host_kernarg = afe_dev_env("AFE_AQL_HOST_KERNARG") != nullptr;      // measurement aid
  static const bool no_bar = afe_dev_env("AFE_AQL_NO_BAR_KERNARG") != nullptr;          // measurement aid
  if (!host_kernarg && !no_bar) {
    // ... written by the host itself where the device's memory is host-visible (fine-grained pool, large BAR): a copy call
    // costs the host 12 us before every dispatch, stores through the BAR one
    FindDevPool fd{};
    fd.a = &a; fd.cpu = q->cpu;
    (void)a.hsa_amd_agent_iterate_memory_pools(q->gpu, dev_pool_cb, &fd);
    void *p = nullptr;
    if (fd.found && a.hsa_amd_memory_pool_allocate(fd.pool, 2 * KERNARG_SLOT, 0, &p) == HSA_STATUS_SUCCESS) {
      hsa_agent_t both[2] = {q->cpu, q->gpu};
      if (a.hsa_amd_agents_allow_access(2, both, nullptr, p) == HSA_STATUS_SUCCESS) { q->kernarg_dev = p; q->kernarg_dev_mapped = true; }
      else (void)a.hsa_amd_memory_pool_free(p);
    }
  }
  if (!host_kernarg && !q->kernarg_dev && hipMalloc(&q->kernarg_dev, 2 * KERNARG_SLOT) != hipSuccess) { (void)hipGetLastError(); q->kernarg_dev = nullptr; }
  if (std::getenv("AFE_PERSIST_DEBUG")) std::fprintf(stderr, "agrifly_engine: AQL queue: kernel arguments in %s\n", !q->kernarg_dev ? "host memory" : q->kernarg_dev_mapped ? "device memory written through the BAR" : "device memory written by hipMemcpy");
  return q;
}

bool aql_close(AqlQueue *q) {
  if (!q) return true;
  Api &a = api();
  if (a.ok) {
    if (q->in_flight) {
      std::string w;
      if (aql_wait(q, 30000000ull, &w) != 0) {
        // The grid did not leave (or the queue broke under it): it may still be writing.  Nothing it can reach is freed --
        // the queue, its signal and the argument slots are LEAKED, and the caller (afe_destroy) leaks the rings and the
        // arena for the same reason (aql_close returns false).
        std::fprintf(stderr, "agrifly_engine: the resident grid did not leave its queue within 30 s (%s); its queue and buffers are leaked rather than freed under it\n", w.c_str());
        return false;
      }
    }
    if (q->queue) (void)a.hsa_queue_destroy(q->queue);
    if (q->have_signal) (void)a.hsa_signal_destroy(q->done);
    if (q->kernarg) (void)a.hsa_amd_memory_pool_free(q->kernarg);
    if (q->kernarg_dev && q->kernarg_dev_mapped) (void)a.hsa_amd_memory_pool_free(q->kernarg_dev);
    else if (q->kernarg_dev) (void)hipFree(q->kernarg_dev);
    if (q->initialised) (void)a.hsa_shut_down();
  }
  delete q;
  return true;
}

bool aql_find_kernel(AqlQueue *q, const void *fn, AqlKernel *out, std::string *why) {
  Api &a = api();
  if (!q || !a.ok || !fn || !out) { if (why) *why = "aql_find_kernel: bad arguments"; return false; }
  // HIP loads a code object when one of its kernels is first needed: asking for the attributes is such a need
  hipFuncAttributes attr;
  if (hipFuncGetAttributes(&attr, fn) != hipSuccess) { (void)hipGetLastError(); if (why) *why = "hipFuncGetAttributes failed for the kernel"; return false; }
  const char *name = hipKernelNameRefByPtr(fn, nullptr);
  if (!name || !*name) { (void)hipGetLastError(); if (why) *why = "hipKernelNameRefByPtr gave no name"; return false; }
  const std::string kd = std::string(name) + ".kd";
  hsa_ven_amd_loader_1_03_pfn_t loader{};
  hsa_status_t st = a.hsa_system_get_major_extension_table(HSA_EXTENSION_AMD_LOADER, 1, sizeof(loader), &loader);
  if (st != HSA_STATUS_SUCCESS || !loader.hsa_ven_amd_loader_iterate_executables) { if (why) *why = hsa_err(a, st, "loader extension table"); return false; }
  FindSymbol fs{};
  fs.a = &a; fs.agent = q->gpu; fs.name_kd = kd.c_str();
  st = loader.hsa_ven_amd_loader_iterate_executables(exec_cb, &fs);
  if (afe_fault("kernel_symbol")) fs.found = false;
  if (!fs.found) { if (why) *why = "kernel descriptor " + kd + " not found in any loaded executable"; return false; }
  *out = fs.k;
  return true;
}

bool aql_in_flight(const AqlQueue *q) { return q && q->in_flight; }
uint64_t aql_last_duration_ns(const AqlQueue *q) { return q ? q->last_ns : 0; }

bool aql_dispatch(AqlQueue *q, const AqlKernel &k, const void *kernarg, size_t bytes, uint32_t workgroups, uint32_t wg_size, std::string *why) {
  Api &a = api();
  if (!q || !a.ok || q->in_flight || !k.object || workgroups == 0) { if (why) *why = "aql_dispatch: queue busy or bad arguments"; return false; }
  const size_t declared = (size_t)k.kernarg_bytes + (afe_fault("kernarg_size") ? 16u : 0u);
  if (bytes != declared || bytes > KERNARG_SLOT) {
    if (why) *why = "kernel-argument segment is " + std::to_string(declared) + " bytes in the code object, " + std::to_string(bytes) + " packed by the host";
    return false;
  }
  if (q->error.load()) { if (why) *why = "the AQL queue reported an error earlier (" + std::to_string(q->error.load()) + ")"; return false; }
  q->slot ^= 1u;
  char *ka = static_cast<char *>(q->kernarg) + q->slot * KERNARG_SLOT;
  std::memcpy(ka, kernarg, bytes);
  if (q->kernarg_dev) {
    char *kd = static_cast<char *>(q->kernarg_dev) + q->slot * KERNARG_SLOT;
    if (q->kernarg_dev_mapped) {
      std::memcpy(kd, kernarg, bytes);
      __atomic_thread_fence(__ATOMIC_SEQ_CST);                                           // (write-combined stores out before the doorbell's)
      volatile char probe = *static_cast<volatile char *>(static_cast<void *>(kd + bytes - 1)); (void)probe;   // a read behind the writes: they have landed
      ka = kd;
    } else if (hipMemcpy(kd, kernarg, bytes, hipMemcpyHostToDevice) == hipSuccess) ka = kd;      // (synchronous: in memory before the doorbell rings)
    else (void)hipGetLastError();
  }
  const uint64_t idx = a.hsa_queue_load_write_index_relaxed(q->queue);
  if (idx - a.hsa_queue_load_read_index_scacquire(q->queue) >= q->queue->size) { if (why) *why = "AQL queue full"; return false; }
  hsa_kernel_dispatch_packet_t *p = static_cast<hsa_kernel_dispatch_packet_t *>(q->queue->base_address) + (idx & (q->queue->size - 1));
  a.hsa_signal_store_relaxed(q->done, 1);
  p->workgroup_size_x = (uint16_t)wg_size; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
  p->reserved0 = 0;
  p->grid_size_x = workgroups * wg_size; p->grid_size_y = 1; p->grid_size_z = 1;
  p->private_segment_size = k.private_bytes;
  p->group_segment_size = k.group_bytes;
  p->kernel_object = k.object;
  p->kernarg_address = ka;
  p->reserved2 = 0;
  p->completion_signal = q->done;
  const uint16_t header = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1u << HSA_PACKET_HEADER_BARRIER) |
                                     (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                                     (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
  const uint16_t setup = (uint16_t)(1u << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS);
  __atomic_store_n(reinterpret_cast<uint32_t *>(p), (uint32_t)header | ((uint32_t)setup << 16), __ATOMIC_RELEASE);   // the packet becomes valid with its header
  a.hsa_queue_store_write_index_screlease(q->queue, idx + 1);
  a.hsa_signal_store_screlease(q->queue->doorbell_signal, (hsa_signal_value_t)idx);
  q->in_flight = true;
  return true;
}

int aql_wait(AqlQueue *q, uint64_t timeout_us, std::string *why) {
  Api &a = api();
  if (!q || !a.ok) return -1;
  if (!q->in_flight) return 0;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; spins++) {
    if (a.hsa_signal_load_scacquire(q->done) < 1) break;
    if (q->error.load()) { if (why) *why = "the AQL queue reported error " + std::to_string(q->error.load()); return -1; }
    if ((spins & 0x3ffu) == 0x3ffu) {
      const auto waited = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
      if ((uint64_t)waited > timeout_us) return 1;
      if (waited > 2000) (void)a.hsa_signal_wait_scacquire(q->done, HSA_SIGNAL_CONDITION_LT, 1, q->ticks_per_s ? q->ticks_per_s / 1000 : 1000000, HSA_WAIT_STATE_BLOCKED);   // long waits sleep (1 ms slices)
    }
  }
  q->in_flight = false;
  hsa_amd_profiling_dispatch_time_t t{};
  if (a.hsa_amd_profiling_get_dispatch_time(q->gpu, q->done, &t) == HSA_STATUS_SUCCESS && t.end > t.start && q->ticks_per_s)
    q->last_ns = (uint64_t)((double)(t.end - t.start) * 1e9 / (double)q->ticks_per_s);
  else q->last_ns = 0;
  return 0;
}

}  // namespace afe
