// afe_engine.cpp -- host side of the C ABI (include/agrifly_engine.h): owns the
// SoA slabs in HBM, the engine clock and the logic-gate plan, and launches the
// kernels of afe_kernels.hip.  Compiled with hipcc; there is no CPU fallback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "afe_aql.h"
#include "afe_host.h"
#include "afe_render.h"
#include "afe_world.h"

using namespace afe;

struct afe_engine {
  int64_t n = 0;
  int64_t stride = 0;       // n rounded up to 256 elements x odd (see afe_create)
  int precision = AFE_F32;
  int device = 0;
  int64_t first_global = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  // afe_set_split_stepping(2): the second half of the ensemble steps on `side_stream`; `split_dirty`: it holds steps
  // the main stream has not been ordered after yet (join_streams does that; everything but afe_step goes through it)
  int split_parts = 0;      // 0: automatic (two parts for 2^19 vehicles and more while the engine owns its stream), 1: off, 2: on
  hipStream_t side_stream = nullptr;
  hipEvent_t ev_main = nullptr, ev_side = nullptr;
  bool split_dirty = false;

  // device slabs
  void *arena = nullptr;
  size_t arena_bytes = 0;
  size_t kernel_bytes = 0;   // the part of the arena the step kernels address (pos .. type): what a buffer resource has to span
  void *pos = nullptr, *vel = nullptr, *att = nullptr, *ang_vel = nullptr, *motor = nullptr;
  void *ext_force = nullptr, *ext_torque = nullptr;
  float *cmd = nullptr, *gyro = nullptr, *acc = nullptr;
  uint32_t *rng = nullptr;
  uint8_t *type = nullptr;
  // afe_create_host_visible: the arena lives in pinned, coherent host memory that the device addresses over the bus;
  // getters and setters are plain host copies once the authorised steps are done (no transfer call, no parked grid)
  bool host_arena = false;
  char *arena_host = nullptr;   // the arena as the host addresses it
  char *logic_arena_host = nullptr;
  size_t logic_arena_alloc = 0;
  bool stream_pending = false;  // something may be queued on the stream since the last synchronisation
  double *anchor = nullptr;   // [2][stride]: where each vehicle's x, y were last set; the pos slab holds x, y RELATIVE to it (fp32 engine)
  void *dev_table = nullptr;  // n_types DevParams<R>
  std::vector<DevParams<float>> table_f32;   // host copies of the device table
  std::vector<DevParams<double>> table_f64;
  bool types_uniform = true;  // every vehicle uses record 0 (kernel-argument fast path)
  bool force_global_addressing = false;   // afe_set_addressing(1): the kernels of arenas beyond 4 GiB, on any arena
  std::vector<uint8_t> type_host;   // host mirror of the type slab (zeros until afe_set_vehicle_types / a checkpoint)
  std::vector<uint8_t> run_nonzero, run_mixed;   // per aligned run of 64 vehicles: count off record 0 (<= 64), mixed types?
  int64_t n_nonzero = 0, n_mixed_runs = 0;
  bool types_wave_uniform = true;   // the type index is constant over every aligned run of 64 vehicles
  float *pack_scratch = nullptr;  // 3*n floats, lazily allocated
  afe_world *world = nullptr;     // shared-world query scratch (uniform grid), lazily created
  // afe_nearest_neighbour_async: the query runs on its own stream behind a snapshot of the gathered positions
  hipStream_t query_stream = nullptr;
  hipEvent_t ev_q_start = nullptr, ev_q_done = nullptr;
  bool query_pending = false;
  // on-device rates logic (allocated by afe_set_rates_logic)
  bool logic_on = false;
  void *logic_arena = nullptr;
  float *lpf = nullptr, *rates_cmd = nullptr;
  uint8_t *have_cmd = nullptr, *imu_init = nullptr;
  DevLogic *dev_logic_table = nullptr;
  std::vector<afe_rates_logic_params> logic_params;
  std::vector<DevLogic> logic_table;
  float logic_table_period = -1.0f;

  // configuration
  std::vector<HostParams> table;
  double table_dt = -1.0;   // dt the device table was built for
  bool table_dirty = true;
  double logic_period = 1.0 / 500.0;  // main.cpp:177
  bool noise = true;
  double sigma_gyro = 0.1, sigma_acc = 0.2;  // Quadcopter_T.cpp:5-6
  int seed_policy = AFE_SEED_REFERENCE;
  uint64_t noise_seed = 0;  // AFE_SEED_COUNTER: the key of the counter-based generator
  // gust process (afe_set_gust_process): the ext_force slab holds the force of epoch `gust_applied`
  bool gust_on = false;
  uint64_t gust_seed = 0, gust_period_us = 0, gust_n_global = 0;
  double gust_sigma_max = 0;
  uint64_t gust_applied = ~0ull;   // epoch whose force is in the slab (~0: none yet)
  bool has_ext_force = false, has_ext_torque = false;
  // stateless motors (tau_m == 0, J_m == 0) driven by held commands: the rotor-speed slab is not
  // written by the step kernel; motor_stale says it has to be rebuilt from the commands first
  bool motor_stale = false;

  // clock (ManualTimer + Timer semantics)
  uint64_t now_us = 0;
  uint64_t logic_elapsed_us = 0;
  uint64_t n_ticks = 0;
  int max_fused = 64;
  int cache_policy = -1;       // afe_set_cache_policy: -1 automatic, 0..3 as LaunchFlags::cache_policy
  uint64_t steps_issued = 0;   // afe_steps_completed

  // persistent stepping (afe_set_step_mode; afe_device.h PersistArgs)
  int step_mode = AFE_STEP_LAUNCH;
  bool p_running = false;   // a resident grid is on the device
  bool p_failed = false;    // a resident grid gave up: the ensemble may be torn between two steps, stepping is refused
  std::string p_fail_msg;   // what it said when it did (kept apart from err, which every later refusal rewrites)
  unsigned long long *p_host = nullptr;      // pinned host memory: ring[AFE_PERSIST_HOST_RING] + status[8]
  unsigned long long *p_host_dev = nullptr;  // the same memory as the device addresses it
  unsigned long long *p_dev = nullptr;       // device memory: ring[AFE_PERSIST_DEV_RING] + done[p_workers]
  int p_workers = 0;
  int p_cus = 0;
  int p_shrink_num = 16;    // sixteenths of the computed capacity still trusted (two stalled grids in a row take one off)
  int p_stall_streak = 0;
  int p_capacity = 0;       // resident one-wave workgroups per CU of the current configuration's kernel (p_capacity_key)
  unsigned p_capacity_key = 0;
  bool p_balanced = false;
  uint64_t p_next = 0;      // ring entries written so far == index of the next step to authorise
  uint64_t p_resume = 0;    // where the next grid starts (every worker's done[] stands there while none runs)
  uint64_t p_dt_us = 0;     // what the resident grid was launched with
  unsigned p_epoch = 0;     // launches so far
  uint64_t p_seg_start = 0, p_seg_t0_us = 0, p_seg_gust_applied = ~0ull;   // the current run of equally long steps: its first index, the engine time and the slab's gust epoch there
  LaunchFlags p_flags = {};
  // The resident grid on a queue of the engine's own (afe_aql.h): hipDeviceSynchronize does not wait for it, afe_sync
  // waits for the steps and lets it live.  Falls back to a launch on the HIP stream when the runtime cannot be reached.
  afe::AqlQueue *aql = nullptr;
  bool aql_tried = false;
  int aql_mode = -1;                // afe_set_resident_queue: -1 automatic (by size), 0 the HIP stream, 1 the engine's own queue
  bool p_on_aql = false;            // the grid now resident was dispatched there
  std::map<unsigned, afe::AqlKernel> aql_kernels;    // by configuration key (persist_size_grid's) | precision << 8
  bool view_exported = false;       // afe_get_device_view has handed the slabs to somebody: afe_sync must leave them readable
  uint64_t p_grid_ns = 0, p_grid_steps = 0;          // device time and steps of the grids collected so far (afe_grid_time)
  uint64_t p_launch_start = 0;                       // the step the grid now resident started from
  bool p_prio = false;                               // the grid now resident balances its workers by issue priority (persist_launch)
  uint64_t p_quiesced = 0;                           // every step below this index is known to be done (the last successful wait)
  uint64_t p_sync_posted = 0;                        // 1 + the step count of the sync request posted to the grid now resident (0: none)

  std::string err;
};

namespace {
void join_streams(afe_engine *e);
hipStream_t main_stream(afe_engine *e);
int persist_park(afe_engine *e);
int quiesce(afe_engine *e);
int gust_resample(afe_engine *e, uint64_t epoch);
}  // namespace

namespace {

inline size_t elem(const afe_engine *e) { return e->precision == AFE_F64 ? 8 : 4; }

int fail(afe_engine *e, int status, const std::string &msg) {
  if (e) e->err = msg;
  return status;
}

#define AFE_HIP(e, call)                                                              \
  do {                                                                                \
    hipError_t err__ = (call);                                                        \
    if (err__ != hipSuccess)                                                          \
      return fail((e), AFE_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(err__)); \
  } while (0)

int check_range(afe_engine *e, int64_t first, int64_t count) {
  if (!e) return AFE_ERR_INVALID_ARG;
  // (count > n - first, not first + count > n: the sum of two large arguments wraps and would pass)
  if (first < 0 || count < 0 || first > e->n || count > e->n - first)
    return fail(e, AFE_ERR_OUT_OF_RANGE, "vehicle range: first " + std::to_string(first) + ", count " +
                                             std::to_string(count) + " outside [0, " + std::to_string(e->n) + ")");
  return AFE_OK;
}

// host planar [comps][count] <-> device planar [comps][stride] at offset first
inline bool in_span(const void *p, const void *base, size_t bytes) { return base && (const char *)p >= (const char *)base && (const char *)p < (const char *)base + bytes; }
inline bool in_host_arena(const afe_engine *e, const void *dev) {
  return e->host_arena && (in_span(dev, e->arena, e->arena_bytes) || in_span(dev, e->logic_arena, e->logic_arena_alloc));
}
inline char *host_of(const afe_engine *e, const void *dev) {
  if (in_span(dev, e->arena, e->arena_bytes)) return e->arena_host + ((const char *)dev - (const char *)e->arena);
  return e->logic_arena_host + ((const char *)dev - (const char *)e->logic_arena);
}

int copy_in(afe_engine *e, void *dev, size_t esz, int comps, int64_t first, int64_t count, const void *host) {
  if (!host || count == 0) return AFE_OK;
  if (in_host_arena(e, dev)) {
    // the slab is host memory: wait until nothing on the device is stepping (a resident grid stays where it is -- its
    // workers touch the slabs only between a ring entry and their completion mark), then write
    const int rc = quiesce(e);
    if (rc) return rc;
    char *d = host_of(e, dev);
    for (int c = 0; c < comps; c++)
      std::memcpy(d + ((size_t)c * e->stride + first) * esz, (const char *)host + (size_t)c * count * esz, (size_t)count * esz);
    __atomic_thread_fence(__ATOMIC_RELEASE);   // ahead of the ring entry / the launch that lets the device read it
    return AFE_OK;
  }
  hipStream_t st = main_stream(e);                   // (ends a resident grid)
  if (e->p_failed) return fail(e, AFE_ERR_HIP, e->err);   // a grid that gave up or did not come back: the ensemble may be torn, nothing is written over it
  AFE_HIP(e, hipMemcpy2DAsync((char *)dev + first * esz, e->stride * esz, host, count * esz,
                              count * esz, comps, hipMemcpyHostToDevice, st));
  AFE_HIP(e, hipStreamSynchronize(st));              // host buffer may be reused by the caller
  return AFE_OK;
}
int copy_out(afe_engine *e, const void *dev, size_t esz, int comps, int64_t first, int64_t count, void *host) {
  if (!host || count == 0) return AFE_OK;
  if (in_host_arena(e, dev)) {
    const int rc = quiesce(e);
    if (rc) return rc;
    const char *d = host_of(e, dev);
    for (int c = 0; c < comps; c++)
      std::memcpy((char *)host + (size_t)c * count * esz, d + ((size_t)c * e->stride + first) * esz, (size_t)count * esz);
    return AFE_OK;
  }
  hipStream_t st = main_stream(e);                   // (ends a resident grid)
  if (e->p_failed) return fail(e, AFE_ERR_HIP, e->err);   // ... and a torn ensemble is not handed out as a state
  AFE_HIP(e, hipMemcpy2DAsync(host, count * esz, (const char *)dev + first * esz, e->stride * esz,
                              count * esz, comps, hipMemcpyDeviceToHost, st));
  AFE_HIP(e, hipStreamSynchronize(st));
  return AFE_OK;
}

// `byte` into rows [first, first + count) of `comps` planar components
int fill_rows(afe_engine *e, void *dev, size_t esz, int comps, int64_t first, int64_t count, int byte) {
  if (count == 0) return AFE_OK;
  if (in_host_arena(e, dev)) {
    const int rc = quiesce(e);
    if (rc) return rc;
    char *d = host_of(e, dev);
    for (int c = 0; c < comps; c++) std::memset(d + ((size_t)c * e->stride + first) * esz, byte, (size_t)count * esz);
    __atomic_thread_fence(__ATOMIC_RELEASE);
    return AFE_OK;
  }
  AFE_HIP(e, hipMemset2DAsync((char *)dev + first * esz, e->stride * esz, byte, count * esz, comps, main_stream(e)));
  return AFE_OK;
}

template <typename Dst, typename Src>
int set_field(afe_engine *e, void *dev, int comps, int64_t first, int64_t count, const Src *host) {
  if (!host) return AFE_OK;
  if (sizeof(Dst) == sizeof(Src)) return copy_in(e, dev, sizeof(Dst), comps, first, count, host);
  std::vector<Dst> tmp((size_t)comps * count);
  for (size_t k = 0; k < tmp.size(); k++) tmp[k] = (Dst)host[k];
  return copy_in(e, dev, sizeof(Dst), comps, first, count, tmp.data());
}
template <typename Src, typename Dst>
int get_field(afe_engine *e, const void *dev, int comps, int64_t first, int64_t count, Dst *host) {
  if (!host) return AFE_OK;
  if (sizeof(Dst) == sizeof(Src)) return copy_out(e, dev, sizeof(Src), comps, first, count, host);
  std::vector<Src> tmp((size_t)comps * count);
  int rc = copy_out(e, dev, sizeof(Src), comps, first, count, tmp.data());
  if (rc) return rc;
  for (size_t k = 0; k < tmp.size(); k++) host[k] = (Dst)tmp[k];
  return AFE_OK;
}

int materialize_motor(afe_engine *e);   // defined below

// Positions.  The dynamics are translation invariant, and an fp32 position far from the origin cannot resolve a slow
// vehicle's motion (at 4 km an ulp is 0.24 mm; a hovering vehicle in a light gust moves microns per step and an absolute
// fp32 x simply stops changing: measured 1.1 cm of lost displacement in 150 steps on bench.py's 4 km lattice).  So the
// fp32 engine integrates x and y RELATIVE to where they were last set: the slab holds the offset, `anchor` (double, never
// touched by the step kernel: no extra bytes per step) holds the set point, and everything that means an absolute
// position -- getters, afe_pack_positions, the depth camera's poses -- adds the two in double.  z stays absolute (the
// ground-contact test of Quadcopter_T.cpp:146 is on it, and altitudes are small).  The fp64 engine keeps absolute
// positions like the reference; its anchors are zero.
template <typename H>
int set_positions(afe_engine *e, int64_t first, int64_t count, const H *pos3) {
  if (!pos3 || count == 0) return AFE_OK;
  int rc;
  std::vector<double> anchors((size_t)2 * count, 0.0);
  if (e->precision == AFE_F64) {
    if ((rc = set_field<double>(e, e->pos, 3, first, count, pos3))) return rc;
  } else {
    std::vector<float> off((size_t)3 * count, 0.0f);
    for (int64_t k = 0; k < count; k++) {
      anchors[(size_t)k] = (double)pos3[k];
      anchors[(size_t)(count + k)] = (double)pos3[count + k];
      off[(size_t)(2 * count + k)] = (float)pos3[2 * count + k];
    }
    if ((rc = copy_in(e, e->pos, 4, 3, first, count, off.data()))) return rc;
  }
  return copy_in(e, e->anchor, 8, 2, first, count, anchors.data());
}
template <typename H>
int get_positions(afe_engine *e, int64_t first, int64_t count, H *pos3) {
  if (!pos3 || count == 0) return AFE_OK;
  int rc;
  std::vector<double> anchors((size_t)2 * count);
  if ((rc = copy_out(e, e->anchor, 8, 2, first, count, anchors.data()))) return rc;
  if (e->precision == AFE_F64) {
    std::vector<double> p((size_t)3 * count);
    if ((rc = copy_out(e, e->pos, 8, 3, first, count, p.data()))) return rc;
    for (int64_t k = 0; k < count; k++) {
      pos3[k] = (H)(anchors[(size_t)k] + p[(size_t)k]);
      pos3[count + k] = (H)(anchors[(size_t)(count + k)] + p[(size_t)(count + k)]);
      pos3[2 * count + k] = (H)p[(size_t)(2 * count + k)];
    }
  } else {
    std::vector<float> p((size_t)3 * count);
    if ((rc = copy_out(e, e->pos, 4, 3, first, count, p.data()))) return rc;
    for (int64_t k = 0; k < count; k++) {
      pos3[k] = (H)(anchors[(size_t)k] + (double)p[(size_t)k]);
      pos3[count + k] = (H)(anchors[(size_t)(count + k)] + (double)p[(size_t)(count + k)]);
      pos3[2 * count + k] = (H)p[(size_t)(2 * count + k)];
    }
  }
  return AFE_OK;
}

template <typename H>
int set_state_any(afe_engine *e, int64_t first, int64_t count, const H *pos3, const H *vel3,
                  const H *att4, const H *ang_vel3, const H *motor4) {
  int rc = check_range(e, first, count);
  if (rc) return rc;
  AFE_HIP(e, hipSetDevice(e->device));
  if (motor4 && (rc = materialize_motor(e))) return rc;   // the rest of the slab must be current
  if (e->precision == AFE_F64) {
    if ((rc = set_positions(e, first, count, pos3))) return rc;
    if ((rc = set_field<double>(e, e->vel, 3, first, count, vel3))) return rc;
    if ((rc = set_field<double>(e, e->att, 4, first, count, att4))) return rc;
    if ((rc = set_field<double>(e, e->ang_vel, 3, first, count, ang_vel3))) return rc;
    if ((rc = set_field<double>(e, e->motor, 4, first, count, motor4))) return rc;
  } else {
    if ((rc = set_positions(e, first, count, pos3))) return rc;
    if ((rc = set_field<float>(e, e->vel, 3, first, count, vel3))) return rc;
    if ((rc = set_field<float>(e, e->att, 4, first, count, att4))) return rc;
    if ((rc = set_field<float>(e, e->ang_vel, 3, first, count, ang_vel3))) return rc;
    if ((rc = set_field<float>(e, e->motor, 4, first, count, motor4))) return rc;
  }
  return AFE_OK;
}

template <typename H>
int get_state_any(afe_engine *e, int64_t first, int64_t count, H *pos3, H *vel3, H *att4,
                  H *ang_vel3, H *motor4) {
  int rc = check_range(e, first, count);
  if (rc) return rc;
  AFE_HIP(e, hipSetDevice(e->device));
  if (motor4 && (rc = materialize_motor(e))) return rc;
  if (e->precision == AFE_F64) {
    if ((rc = get_positions(e, first, count, pos3))) return rc;
    if ((rc = get_field<double>(e, e->vel, 3, first, count, vel3))) return rc;
    if ((rc = get_field<double>(e, e->att, 4, first, count, att4))) return rc;
    if ((rc = get_field<double>(e, e->ang_vel, 3, first, count, ang_vel3))) return rc;
    if ((rc = get_field<double>(e, e->motor, 4, first, count, motor4))) return rc;
  } else {
    if ((rc = get_positions(e, first, count, pos3))) return rc;
    if ((rc = get_field<float>(e, e->vel, 3, first, count, vel3))) return rc;
    if ((rc = get_field<float>(e, e->att, 4, first, count, att4))) return rc;
    if ((rc = get_field<float>(e, e->ang_vel, 3, first, count, ang_vel3))) return rc;
    if ((rc = get_field<float>(e, e->motor, 4, first, count, motor4))) return rc;
  }
  return AFE_OK;
}

int set_wrench(afe_engine *e, void *dev, bool &flag, int64_t first, int64_t count, const double *w3) {
  int rc = check_range(e, first, count);
  if (rc) return rc;
  AFE_HIP(e, hipSetDevice(e->device));
  if (!w3) {
    return fill_rows(e, dev, elem(e), 3, first, count, 0);
  }
  flag = true;
  if (e->precision == AFE_F64) return set_field<double>(e, dev, 3, first, count, w3);
  return set_field<float>(e, dev, 3, first, count, w3);
}

// (re)build the device type table when the table or dt changed
int refresh_table(afe_engine *e, double dt) {
  if (!e->table_dirty && dt == e->table_dt) return AFE_OK;
  const size_t n = e->table.size();
  if (e->precision == AFE_F64) {
    std::vector<DevParams<double>> &t = e->table_f64;
    t.resize(n);
    for (size_t k = 0; k < n; k++) to_device_params<double>(e->table[k], dt, t[k]);
    AFE_HIP(e, hipMemcpyAsync(e->dev_table, t.data(), n * sizeof(t[0]), hipMemcpyHostToDevice, main_stream(e)));
    AFE_HIP(e, hipStreamSynchronize(main_stream(e)));
  } else {
    std::vector<DevParams<float>> &t = e->table_f32;
    t.resize(n);
    for (size_t k = 0; k < n; k++) to_device_params<float>(e->table[k], dt, t[k]);
    AFE_HIP(e, hipMemcpyAsync(e->dev_table, t.data(), n * sizeof(t[0]), hipMemcpyHostToDevice, main_stream(e)));
    AFE_HIP(e, hipStreamSynchronize(main_stream(e)));
  }
  e->table_dt = dt;
  e->table_dirty = false;
  return AFE_OK;
}

// (re)build the logic constants when the table or the logic period changed
int refresh_logic(afe_engine *e) {
  const float period = (float)e->logic_period;  // float(onboardLogicPeriod), Quadcopter_T.cpp:18
  if (period == e->logic_table_period) return AFE_OK;
  const size_t n = e->logic_params.size();
  e->logic_table.resize(n);
  for (size_t k = 0; k < n; k++) {
    const char *why = "";
    int rc = expand_logic(e->logic_params[k], period, e->logic_table[k], &why);
    if (rc) return fail(e, rc, "logic type " + std::to_string(k) + ": " + why);
  }
  AFE_HIP(e, hipMemcpyAsync(e->dev_logic_table, e->logic_table.data(), n * sizeof(DevLogic), hipMemcpyHostToDevice, main_stream(e)));
  AFE_HIP(e, hipStreamSynchronize(main_stream(e)));
  e->logic_table_period = period;
  return AFE_OK;
}

bool motors_stateless(const afe_engine *e) {
  if (e->table.empty()) return false;
  for (const HostParams &h : e->table) if (h.tau_m != 0 || h.Jm != 0) return false;
  return true;
}
// the step kernel may skip the rotor-speed store: the speed is clamp(cmd) and the commands
// stay what they were during the step (the on-device logic rewrites them at every tick)
// (a host-visible arena is for small ensembles with the host in the loop: there the sixteen bytes are written and
// nothing ever has to be rebuilt by a launch)
bool motor_lazy(const afe_engine *e) { return motors_stateless(e) && !e->logic_on && !e->host_arena; }

// rebuild the rotor-speed slab from the commands of the last step, if it was skipped
int materialize_motor(afe_engine *e) {
  if (!e->motor_stale) return AFE_OK;
  AFE_HIP(e, hipSetDevice(e->device));
  // the device table holds the parameters the last step ran with (refresh_table precedes every launch)
  const int lrc = e->precision == AFE_F64
      ? launch_motor_from_cmd_f64((double *)e->motor, e->cmd, e->types_uniform ? nullptr : e->type,
                                  (const DevParams<double> *)e->dev_table, e->stride, e->n, main_stream(e))
      : launch_motor_from_cmd_f32((float *)e->motor, e->cmd, e->types_uniform ? nullptr : e->type,
                                  (const DevParams<float> *)e->dev_table, e->stride, e->n, main_stream(e));
  if (lrc != 0) return fail(e, AFE_ERR_HIP, std::string("rotor-speed rebuild: ") + hipGetErrorString((hipError_t)lrc));
  e->motor_stale = false;
  return AFE_OK;
}

size_t logic_arena_bytes(const afe_engine *e);

template <typename R>
void fill_view(const afe_engine *e, StepView<R> &v) {
  v.lpf = e->lpf; v.rates_cmd = e->rates_cmd; v.have_cmd = e->have_cmd; v.imu_init = e->imu_init;
  v.cmd_out = e->cmd; v.logic_table = e->dev_logic_table;
  v.pos = (R *)e->pos; v.vel = (R *)e->vel; v.att = (R *)e->att;
  v.ang_vel = (R *)e->ang_vel; v.motor = (R *)e->motor;
  v.ext_force = (const R *)e->ext_force; v.ext_torque = (const R *)e->ext_torque;
  v.cmd = e->cmd; v.gyro = e->gyro; v.acc = e->acc; v.rng = e->rng; v.type = e->type;
  v.table = (const DevParams<R> *)e->dev_table;
  v.n_types = (int)e->table.size();
  v.n = e->n; v.stride = e->stride;
  v.first = 0; v.end = e->n;
  v.motor_stateless = motors_stateless(e);
  v.motor_write = !motor_lazy(e);
  v.sigma_gyro = (float)e->sigma_gyro;  // float(_stdDevRateGyroNoise), Quadcopter_T.cpp:170
  v.sigma_acc = (float)e->sigma_acc;
  v.noise_seed = e->noise_seed;
  v.tick_base = e->n_ticks;
  v.first_global = e->first_global;
  // one buffer resource spans the arena (pos is its first slab), another the logic arena (lpf first)
  const size_t lbytes = logic_arena_bytes(e);
  const bool fits = !e->force_global_addressing && e->kernel_bytes < 0xffff0000ull && lbytes < 0xffff0000ull;
  v.buf_bytes = fits ? (uint32_t)e->kernel_bytes : 0u;
  v.logic_buf_bytes = fits ? (uint32_t)lbytes : 0u;
}

// Orders the main stream after whatever the side stream still holds (split stepping).  Cheap when nothing is
// pending.  Every entry point that touches device state or the stream reaches it through main_stream().
void join_streams(afe_engine *e) {
  if (!e->split_dirty) return;
  e->split_dirty = false;
  if (hipEventRecord(e->ev_side, e->side_stream) == hipSuccess) (void)hipStreamWaitEvent(e->stream, e->ev_side, 0);
  else (void)hipStreamSynchronize(e->side_stream);
}
hipStream_t main_stream(afe_engine *e) {
  join_streams(e);
  if (e->p_running) (void)persist_park(e);   // a failure is sticky (p_failed) and reported by the next afe_step / afe_sync
  e->stream_pending = true;                  // whoever asks is about to queue something
  return e->stream;
}

// ---- persistent stepping, host side -------------------------------------------------------------------------
inline volatile unsigned long long *p_status(afe_engine *e) { return e->p_host + AFE_PERSIST_HOST_RING; }

int persist_alloc(afe_engine *e) {
  if (e->p_host && e->p_host_dev && e->p_dev) return AFE_OK;
  hipDeviceProp_t prop;
  AFE_HIP(e, hipGetDeviceProperties(&prop, e->device));
  e->p_cus = prop.multiProcessorCount;
  const int64_t chunks = (e->n + 63) / 64;
  const int64_t most = (int64_t)e->p_cus * 32;      // a CU has 32 wave slots: done[] never needs more
  const size_t hbytes = (AFE_PERSIST_HOST_RING + AFE_PERSIST_STATUS_WORDS) * sizeof(unsigned long long);
  const size_t dwords = 8 + (size_t)AFE_PERSIST_DEV_RING + (((size_t)(chunks < most ? chunks : most) + 8 + 15) & ~(size_t)15);
  const size_t dbytes = (dwords + AFE_PERSIST_SYNC_AREA_WORDS) * sizeof(unsigned long long);   // [8 words: the workers' call-for-help word is the last of them][ring][done[]][sync area, a line per counter]
  // all three or none: a half-made set (the second or third call failing) must not look complete to the next afe_step
  hipError_t herr = e->p_host ? hipSuccess : hipHostMalloc((void **)&e->p_host, hbytes, hipHostMallocCoherent | hipHostMallocMapped);
  if (herr == hipSuccess) {
    std::memset(e->p_host, 0, hbytes);
    herr = hipHostGetDevicePointer((void **)&e->p_host_dev, e->p_host, 0);
  }
  if (herr == hipSuccess && !e->p_dev) herr = hipMalloc((void **)&e->p_dev, dbytes);
  if (herr == hipSuccess) herr = hipMemsetAsync(e->p_dev, 0, dbytes, e->stream);
  if (herr != hipSuccess) {
    if (e->p_dev) (void)hipFree(e->p_dev);
    if (e->p_host) (void)hipHostFree(e->p_host);
    e->p_dev = nullptr; e->p_host = nullptr; e->p_host_dev = nullptr;
    (void)hipGetLastError();
    return fail(e, AFE_ERR_HIP, std::string("persistent stepping: allocating the rings: ") + hipGetErrorString(herr));
  }
  return AFE_OK;
}

// worker waves of the grid about to be launched: one per chunk if the device keeps that many resident, else
// every resident slot but the pump's (the instantiation's occupancy x CUs), less whatever earlier stalls took off
void persist_size_grid(afe_engine *e) {
  // (the occupancy query and the environment are asked once per configuration, not at every launch: a grid is started
  // after every synchronisation of a host that steps in short blocks)
  const unsigned key = 1u | (e->p_flags.ext_force ? 2u : 0u) | (e->p_flags.noise ? 4u : 0u) | (e->p_flags.logic ? 8u : 0u) |
                       (e->p_flags.counter_noise ? 16u : 0u) | (e->p_flags.resident ? 32u : 0u);
  if (e->p_capacity_key != key) {
    int cap = e->precision == AFE_F64 ? persistent_capacity_f64(e->p_flags) : persistent_capacity_f32(e->p_flags);
    if (cap < 1) cap = 1;
    if (const char *s = afe_dev_env("AFE_PERSIST_WAVES_PER_CU")) { const int k = std::atoi(s); if (k >= 1 && k <= 32) cap = k; }
    e->p_capacity = cap;
    e->p_capacity_key = key;
    e->p_balanced = afe_dev_env("AFE_PERSIST_BALANCED") != nullptr;
  }
  const int per_cu = e->p_capacity;
  const int64_t chunks = (e->n + 63) / 64;
  const int cus = e->p_cus;
  int64_t cap = (int64_t)cus * per_cu - 1;          // the pump takes one slot
  cap = cap * e->p_shrink_num / 16;
  if (cap < 1) cap = 1;
  // Every resident slot gets a wave, even when the chunks do not divide evenly (2^20 vehicles: 16 384 chunks over 6 143
  // waves, three for most, two for a third of them).  Measured (tools/persist_waves_probe.py):
  // the waves with less to do run ahead into the ring's window and wait there with backed-off polls, costing nothing,
  // while equal shares on fewer waves (5 462 x 3) leave memory parallelism unused: 19.4 against 21.0 us per step.
  e->p_workers = (int)(chunks < cap ? chunks : cap);
  if (e->p_balanced) {      // AFE_PERSIST_BALANCED, measurement aid: the fewest waves with equal shares
    const int64_t per_wave = (chunks + cap - 1) / cap;
    e->p_workers = (int)((chunks + per_wave - 1) / per_wave);
  }
}

// the configuration a resident grid carries in its kernel arguments
LaunchFlags persist_flags(const afe_engine *e) {
  LaunchFlags f;
  f.ext_force = e->has_ext_force; f.ext_torque = false; f.noise = e->noise; f.logic = e->logic_on;
  f.counter_noise = e->seed_policy == AFE_SEED_COUNTER;
  // AFE_STEP_AUTO takes the steps already authorised together (state in registers between them, every step stored): the
  // same bits, and faster at every size measured (tools/small_n_probe.py: 4 096 vehicles 1.46 -> 0.94 us per step,
  // 131 072 2.40 -> 1.56; bench.py companions at 2^20: 19.5 -> 10.9)
  f.resident = e->step_mode == AFE_STEP_RESIDENT || e->step_mode == AFE_STEP_AUTO;
  return f;
}

// launch mode: the gust force of `epoch` into the ext_force slab, stream-ordered before the step that needs it
int gust_resample(afe_engine *e, uint64_t epoch) {
  const int lrc = e->precision == AFE_F64
      ? launch_gust_f64((double *)e->ext_force, e->stride, e->n, e->first_global, e->gust_n_global, e->gust_seed, epoch, e->gust_sigma_max, main_stream(e))
      : launch_gust_f32((float *)e->ext_force, e->stride, e->n, e->first_global, e->gust_n_global, e->gust_seed, epoch, e->gust_sigma_max, main_stream(e));
  if (lrc != 0) return fail(e, AFE_ERR_HIP, std::string("gust kernel launch: ") + hipGetErrorString((hipError_t)lrc));
  e->gust_applied = epoch;
  return AFE_OK;
}
bool persist_eligible(const afe_engine *e) {
  if (e->step_mode == AFE_STEP_LAUNCH || e->p_failed) return false;
  if (e->step_mode == AFE_STEP_AUTO && e->n > (int64_t(1) << 20)) return false;   // measured: DESIGN.md section 6 (beyond the Infinity Cache the split launches are ahead)
  if (!e->types_uniform || e->has_ext_torque || e->stream != e->own_stream || e->force_global_addressing) return false;
  // a host-visible arena is streamed over the bus by every step: beyond a few thousand vehicles a step takes longer than
  // the grid's patience with itself (measured: 2^18 vehicles stall it), and the launches are no slower there
  if (e->host_arena && e->n > 16384) return false;
  return e->kernel_bytes < 0xffff0000ull && logic_arena_bytes(e) < 0xffff0000ull;
}

// The first user-mode queue on which a process dispatches a KERNEL is slower than every later one, for as long as it
// lives: measured, an engine whose queue was that one stepped 2^20 vehicles in 19.9-20.1 us where every later engine of the
// process -- and the same engine after giving that queue up for a new one -- took 19.2-19.4 (queues created but not used
// before it, or used for a barrier packet only, change nothing; the HIP runtime's own queues do not show it).  Cause
// unknown (firmware scheduling state is the guess).  So the process's first dispatch goes to a throwaway queue: the
// resident kernel with no workers and a park entry already waiting -- one wave that zeroes a few counters and leaves.
void aql_prime_process(afe_engine *e) {
  // once per device (the observation is per device: its first dispatching queue).  Engines may be created from several
  // host threads (one engine per thread): the lock is held across the priming dispatch, so no second engine's grid can
  // become the device's first dispatch while the throwaway one is still on its way.
  static std::mutex lock;
  static std::set<int> primed;
  std::lock_guard<std::mutex> guard(lock);
  if (primed.count(e->device) || afe_dev_env("AFE_AQL_NO_PRIME")) return;
  primed.insert(e->device);
  std::string why;
  afe::AqlQueue *q = afe::aql_open(e->device, &why);
  if (!q) return;
  afe::AqlKernel k;
  LaunchFlags f;
  f.ext_force = false; f.ext_torque = false; f.noise = false; f.logic = false;
  unsigned long long *host = nullptr, *host_dev = nullptr, *dev = nullptr;
  const size_t hwords = 64 + AFE_PERSIST_STATUS_WORDS, dwords = 8 + 64 + 32 + AFE_PERSIST_SYNC_AREA_WORDS;
  if (afe::aql_find_kernel(q, persistent_kernel_fn_f32(f), &k, &why) &&
      hipHostMalloc((void **)&host, hwords * 8, hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess &&
      hipHostGetDevicePointer((void **)&host_dev, host, 0) == hipSuccess && hipMalloc((void **)&dev, dwords * 8) == hipSuccess &&
      hipMemset(dev, 0, dwords * 8) == hipSuccess) {
    std::memset(host, 0, hwords * 8);
    host[0] = (1ull << 2) | AFE_PERSIST_PARK;            // "park before step 0": the pump's first look ends the launch
    PersistArgs a = {};
    a.host_ring = host_dev; a.host_status = host_dev + 64; a.dev_ring = dev + 8; a.done = dev + 8 + 64;
    a.start = 0; a.host_mask = 63; a.dev_mask = 63; a.n_workers = 0; a.n_chunks = 0;
    a.idle_ticks = 1000; a.give_up_ticks = 100000; a.epoch = 1;
    StepView<float> v = {};
    DevParams<float> P = {};
    DevLogic G = {};
    alignas(16) char buf[sizeof(PersistKernarg<float>)];
    std::memset(buf, 0, sizeof(buf));
    std::memcpy(buf + offsetof(PersistKernarg<float>, v), &v, sizeof(v));
    std::memcpy(buf + offsetof(PersistKernarg<float>, P), &P, sizeof(P));
    std::memcpy(buf + offsetof(PersistKernarg<float>, G), &G, sizeof(G));
    std::memcpy(buf + offsetof(PersistKernarg<float>, a), &a, sizeof(a));
    if (afe::aql_dispatch(q, k, buf, persist_kernarg_bytes<float>(), 1, 64, &why)) (void)afe::aql_wait(q, 2000000ull, &why);
  }
  (void)hipGetLastError();
  if (!afe::aql_close(q)) return;     // (the throwaway dispatch never left: its two buffers stay, see aql_close)
  if (dev) (void)hipFree(dev);
  if (host) (void)hipHostFree(host);
}

// The engine's own AQL queue and the descriptor of the kernel of the current configuration, or nullptr: this grid goes to
// the HIP stream (AFE_PERSIST_AQL=0; a caller's stream; the runtime out of reach -- said once on stderr).
const afe::AqlKernel *aql_kernel_for(afe_engine *e) {
  // Where the resident grid lives.  On the engine's own queue it survives afe_sync (and no HIP synchronisation waits for
  // it), which is worth a launch and a park per synchronised block; but a dispatch there costs the host 14 us against 3
  // of a HIP launch, workers that are done poll where on the HIP stream they would have left, and the pauses between
  // blocks are inside it.  Measured, 20-step blocks (afe_sync alone on both sides) / 2 000-step blocks, us per step, own
  // queue against HIP stream: 131 072 vehicles 2.70 / 2.20 against 3.22 / 2.19; 262 144: 4.15 / 3.54 against 4.70 / 3.61;
  // 524 288: 10.56 / 10.11 against 11.14 / 10.08; 2^20: 20.43 / 19.24 against 20.13 / 19.32.
  // Automatic: the own queue up to 262 144 vehicles, the HIP stream beyond -- not 524 288, where the own queue is 5 % ahead in
  // 20-step blocks: from ~400 000 vehicles on a grid holds every wave slot of the device, and a grid that afe_sync leaves
  // resident makes whatever else the process launches next (an RCCL collective between two blocks, a torch kernel) wait
  // for its 200 us of idle patience; at 262 144 it holds 16 of a compute unit's 24 and others run beside it.
  // afe_set_resident_queue / AFE_PERSIST_AQL = 0 | 1 force one.
  static const int env_mode = [] { const char *s = std::getenv("AFE_PERSIST_AQL"); return !s || !*s ? -1 : (s[0] == '0' ? 0 : 1); }();
  const int mode = e->aql_mode >= 0 ? e->aql_mode : env_mode;
  if (mode == 0 || (mode < 0 && e->n > 262144) || e->stream != e->own_stream) return nullptr;
  if (!e->aql_tried) {
    e->aql_tried = true;
    std::string why;
    aql_prime_process(e);
    e->aql = afe::aql_open(e->device, &why);
    if (!e->aql) std::fprintf(stderr, "agrifly_engine: no AQL queue for the resident grid (%s); it is launched on the HIP stream and parked at every synchronisation\n", why.c_str());
  }
  if (!e->aql) return nullptr;
  const unsigned key = (e->p_flags.ext_force ? 2u : 0u) | (e->p_flags.noise ? 4u : 0u) | (e->p_flags.logic ? 8u : 0u) | (e->p_flags.counter_noise ? 16u : 0u) |
                       (e->p_flags.resident ? 32u : 0u) | ((unsigned)e->precision << 8);
  auto it = e->aql_kernels.find(key);
  if (it == e->aql_kernels.end()) {
    afe::AqlKernel k;
    std::string why;
    const void *fn = e->precision == AFE_F64 ? persistent_kernel_fn_f64(e->p_flags) : persistent_kernel_fn_f32(e->p_flags);
    if (!afe::aql_find_kernel(e->aql, fn, &k, &why)) {
      std::fprintf(stderr, "agrifly_engine: resident grid stays on the HIP stream (%s)\n", why.c_str());
      k = afe::AqlKernel();       // object 0: remembered as not available
    }
    if (std::getenv("AFE_PERSIST_DEBUG")) std::fprintf(stderr, "agrifly_engine: kernel descriptor at %#llx (kernarg %u B, scratch %u B) for configuration %#x\n", (unsigned long long)k.object, k.kernarg_bytes, k.private_bytes, key);
    it = e->aql_kernels.emplace(key, k).first;
  }
  return it->second.object ? &it->second : nullptr;
}

// The kernel-argument segment of a resident grid: PersistKernarg<R> (afe_device.h) IS the layout -- the four by-value
// arguments in order, each at its natural alignment, which is the rule the code object's own argument offsets follow
// (tests/test_kernel_resources.py compares every argument's offset and size in the code object's metadata with
// afe_persistent_kernarg_layout(); aql_dispatch compares the total).
template <typename R>
bool aql_launch(afe_engine *e, const afe::AqlKernel &k, const StepView<R> &v, const DevParams<R> &P, const DevLogic &G, const PersistArgs &a, int *stream_error) {
  alignas(16) char buf[sizeof(PersistKernarg<R>)];
  std::memset(buf, 0, sizeof(buf));                    // (padding between the arguments is zero, not stack)
  std::memcpy(buf + offsetof(PersistKernarg<R>, v), &v, sizeof(v));
  std::memcpy(buf + offsetof(PersistKernarg<R>, P), &P, sizeof(P));
  std::memcpy(buf + offsetof(PersistKernarg<R>, G), &G, sizeof(G));
  std::memcpy(buf + offsetof(PersistKernarg<R>, a), &a, sizeof(a));
  // the queue is not ordered behind the HIP stream: whatever the stream still holds (setters, a memset) finishes first
  const hipError_t serr = hipStreamSynchronize(e->stream);
  if (serr != hipSuccess) { *stream_error = (int)serr; return false; }     // (a broken stream is an error, not a reason to launch on it)
  e->stream_pending = false;
  std::string why;
  if (!afe::aql_dispatch(e->aql, k, buf, persist_kernarg_bytes<R>(), (uint32_t)(1 + a.n_workers), 64, &why)) {
    static bool said = false;
    if (!said) std::fprintf(stderr, "agrifly_engine: AQL dispatch of the resident grid refused (%s); using the HIP stream\n", why.c_str());
    said = true;
    return false;
  }
  return true;
}

int persist_launch(afe_engine *e) {
  volatile unsigned long long *st = p_status(e);
  st[0] = 0; st[1] = e->p_resume; st[2] = 0; st[7] = 0;
  st[AFE_PERSIST_SYNC_WORD] = 0; st[AFE_PERSIST_SYNCREQ_WORD] = 0;
  e->p_sync_posted = 0;
  const afe::AqlKernel *ak = aql_kernel_for(e);      // nullptr: launch on the HIP stream
  persist_size_grid(e);
  for (int w = 0; w < AFE_PERSIST_HOST_MARKS; w++) st[8 + w] = w < e->p_workers ? e->p_resume : ~0ull;
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
  PersistArgs a;
  a.host_ring = e->p_host_dev;
  a.host_status = e->p_host_dev + AFE_PERSIST_HOST_RING;
  a.dev_ring = e->p_dev + 8;
  a.done = e->p_dev + 8 + AFE_PERSIST_DEV_RING;
  a.start = e->p_resume;
  a.host_mask = AFE_PERSIST_HOST_RING - 1; a.dev_mask = AFE_PERSIST_DEV_RING - 1;
  a.n_workers = e->p_workers;
  a.n_chunks = (int)((e->n + 63) / 64);
  a.idle_ticks = 20000;         // 200 us
  a.give_up_ticks = 5000000;    // 50 ms without any progress while steps are waiting
  // Issue priority by steps in hand (afe_kernels.hip, persist_set_priority): where workers share a SIMD and the step is
  // bound by instruction issue and latency, not by HBM.  Measured (tools/sync_cost_probe.py / refresh_probe.py, 20-step /
  // 2 000-step blocks, us per step without -> with, fixed bands of 8 / 4 steps): 131 072 vehicles 2.88 -> 2.70 / 2.36 -> 2.20, 262 144 4.96 -> 4.19 /
  // 4.21 -> 3.58, 393 216 6.9 -> 6.7 / 6.4 -> 5.95, 524 288 11.4 -> 11.3 / 10.7 -> 10.1; one worker per SIMD (65 536) 2.13 ->
  // 2.18: nothing to arbitrate; 2^20 (2.7 chunks per worker, HBM-bound) 20.3 -> 20.5 / no change: waves in step with each
  // other load together and compute together.  AFE_PERSIST_PRIO=0|1 forces it (measurement aid).
  static const int prio_env = [] { const char *s = afe_dev_env("AFE_PERSIST_PRIO"); return s && *s ? std::atoi(s) : -1; }();
  const int64_t chunks_now = (e->n + 63) / 64;
  const bool prio = prio_env >= 0 ? prio_env != 0 : (e->p_workers > 4 * e->p_cus && chunks_now <= 2 * (int64_t)e->p_workers);
  e->p_prio = prio;
  a.epoch = (++e->p_epoch & 0xffffu) | (e->host_arena ? AFE_PERSIST_HOST_IO : 0u) | (prio ? AFE_PERSIST_PRIO : 0u);
  // The books at step p_resume, where this grid starts (the host's own clock, tick count and gust epoch are already
  // those of step p_next, the end of everything authorised): its start time is linear in the step index since
  // p_seg_start; the ticks before it are the engine's count less the tick flags of the entries still ahead of it;
  // the slab holds the gust of the step before it (or what it held when the run began).
  const uint64_t t0 = e->p_seg_t0_us + (e->p_resume - e->p_seg_start) * e->p_dt_us;
  uint64_t ticks_ahead = 0;
  for (uint64_t k = e->p_resume; k < e->p_next; k++) ticks_ahead += e->p_host[k & (AFE_PERSIST_HOST_RING - 1)] & AFE_PERSIST_TICK;
  const uint64_t ticks0 = e->n_ticks - ticks_ahead;
  a.t0_us = t0; a.dt_us = e->p_dt_us;
  a.gust_period_us = e->gust_on ? e->gust_period_us : 0;
  a.gust_seed = e->gust_seed; a.gust_n_global = e->gust_n_global; a.gust_sigma_max = e->gust_sigma_max;
  a.gust_epoch0 = e->gust_on ? t0 / e->gust_period_us : 0;
  a.gust_epoch_applied = !e->gust_on ? 0 : (e->p_resume > e->p_seg_start ? (t0 - e->p_dt_us) / e->gust_period_us : e->p_seg_gust_applied);
  const double dt = us_to_seconds(e->p_dt_us);
  const LaunchFlags &f = e->p_flags;
  const DevLogic *ulogic = e->logic_on ? &e->logic_table[0] : nullptr;
  static const DevLogic no_logic = {};
  int lrc = 0, stream_error = 0;
  bool on_aql = false;
  if (e->precision == AFE_F64) {
    StepView<double> v;
    fill_view(e, v);
    v.dt = dt; v.inv_dt = 1.0 / dt; v.n_steps = 1; v.tick_mask = 0; v.tick_base = ticks0;
    if (ak) on_aql = aql_launch(e, *ak, v, e->table_f64[0], ulogic ? *ulogic : no_logic, a, &stream_error);
    if (!on_aql && !stream_error) lrc = launch_persistent_f64(v, f, e->table_f64[0], ulogic, a, e->stream);
  } else {
    StepView<float> v;
    fill_view(e, v);
    v.dt = (float)dt; v.inv_dt = (float)(1.0 / dt); v.n_steps = 1; v.tick_mask = 0; v.tick_base = ticks0;
    if (ak) on_aql = aql_launch(e, *ak, v, e->table_f32[0], ulogic ? *ulogic : no_logic, a, &stream_error);
    if (!on_aql && !stream_error) lrc = launch_persistent_f32(v, f, e->table_f32[0], ulogic, a, e->stream);
  }
  if (stream_error) return fail(e, AFE_ERR_HIP, std::string("engine stream ahead of the resident grid's dispatch: ") + hipGetErrorString((hipError_t)stream_error));
  if (lrc != 0) return fail(e, AFE_ERR_HIP, std::string("persistent step kernel launch: ") + hipGetErrorString((hipError_t)lrc));
  e->p_on_aql = on_aql;
  e->p_launch_start = e->p_resume;
  if (!on_aql) e->stream_pending = true;
  static const bool debug = std::getenv("AFE_PERSIST_DEBUG") != nullptr;   // development aid: one line per grid
  if (debug) std::fprintf(stderr, "agrifly_engine: grid %u: %d workers (%d per CU allowed) for %lld chunks, from step %llu, resident-state %d noise %d logic %d force %d\n",
                          e->p_epoch, e->p_workers, e->p_capacity, (long long)a.n_chunks, (unsigned long long)a.start, (int)f.resident, (int)f.noise + (int)f.counter_noise, (int)f.logic, (int)f.ext_force);
  e->p_running = true;
  return AFE_OK;
}

// The resident grid has left the device (or is leaving: the stream says when): where did it stop?
int persist_collect(afe_engine *e) {
  hipError_t herr = hipSuccess;
  const bool was_aql = e->p_on_aql;
  if (e->p_on_aql) {
    std::string why;
    const int w = afe_fault("park_timeout") ? 1 : afe::aql_wait(e->aql, 120000000ull, &why);     // (the grid's own patience ends long before: 50 ms without progress)
    if (w != 0) {
      e->p_running = false; e->p_on_aql = false; e->p_failed = true;
      return fail(e, AFE_ERR_HIP, "persistent step kernel on the engine's AQL queue: " + (w > 0 ? std::string("still running after 120 s") : why));
    }
    e->p_grid_ns += afe::aql_last_duration_ns(e->aql);
    e->p_on_aql = false;
    // AFE_GRID_LOG=<file> (one of the release library's six variables; tools/profile_r05.sh): one line per grid that has
    // left the engine's own queue -- the steps it served and its device time -- in dispatch order, to be laid beside
    // rocprofv3's kernel trace of the same run (a grid there serves however many blocks were authorised while it lived)
    static FILE *const grid_log = [] { const char *p = std::getenv("AFE_GRID_LOG"); return p && *p ? std::fopen(p, "a") : (FILE *)nullptr; }();
    if (grid_log) {
      volatile unsigned long long *stl = p_status(e);
      std::fprintf(grid_log, "%lld,%d,%llu,%llu\n", (long long)e->n, e->p_workers, (unsigned long long)(stl[0] ? stl[0] - 1 - e->p_launch_start : 0),
                   (unsigned long long)afe::aql_last_duration_ns(e->aql));
      std::fflush(grid_log);
    }
  } else {
    herr = hipStreamSynchronize(e->stream);
  }
  e->p_running = false;
  volatile unsigned long long *st = p_status(e);
  if (herr == hipSuccess && st[0] != 0 && st[2] == 1 && e->p_shrink_num > 1) {
    // The pump saw steps waiting and no worker moving for 50 ms: the grid was not co-resident (workgroups that never
    // started held the ring's window shut).  It parked at st[0] - 1 and every workgroup, late ones included, stopped
    // there: nothing is torn.  The next grid is cut smaller.
    // (once: the same size again -- beside another engine's launches a grid can find the register file fragmented, which
    // says nothing about the next start; twice in a row: the capacity is wrong, take a sixteenth off)
    if (++e->p_stall_streak >= 2) { e->p_shrink_num--; e->p_stall_streak = 0; }
    std::fprintf(stderr, "agrifly_engine: a resident grid of %d worker waves stalled; continuing with %d/16 of the computed capacity "
                         "(the pump had republished up to step %llu; the slowest worker, #%u, and %u with it stood at %llu; help word %llu)\n",
                 e->p_workers, e->p_shrink_num, (unsigned long long)st[4], (unsigned)(st[5] >> 32), (unsigned)(st[5] & 0xffffffffu),
                 (unsigned long long)st[3], (unsigned long long)(st[6] >> 32));
  } else if (herr != hipSuccess || st[0] == 0 || st[2] != 0) {
    e->p_failed = true;
    return fail(e, AFE_ERR_HIP, herr != hipSuccess ? std::string("persistent step kernel: ") + hipGetErrorString(herr)
                                   : st[2] ? "persistent step kernel gave up waiting (code " + std::to_string(st[2]) + "); the ensemble may be torn between two steps"
                                           : std::string("persistent step kernel ended without parking"));
  }
  else e->p_stall_streak = 0;     // (a grid that parked in the ordinary way)
  e->p_resume = st[0] - 1;
  if (was_aql && e->p_resume >= e->p_launch_start) e->p_grid_steps += e->p_resume - e->p_launch_start;
  return AFE_OK;
}

// End the resident grid after the last authorised step; returns with the stream idle and every step done.
int persist_park(afe_engine *e) {
  if (!e->p_running) return AFE_OK;
  (void)hipSetDevice(e->device);
  __atomic_store_n(&e->p_host[e->p_next & (AFE_PERSIST_HOST_RING - 1)], ((e->p_next + 1) << 2) | AFE_PERSIST_PARK, __ATOMIC_RELEASE);
  for (;;) {
    int rc = persist_collect(e);
    if (rc) return rc;
    if (e->p_resume == e->p_next) return AFE_OK;
    // the grid had parked itself earlier (the host was quiet for a while): a new one finishes the rest
    if (e->p_resume > e->p_next) { e->p_failed = true; return fail(e, AFE_ERR_HIP, "persistent step kernel ran past the authorised steps"); }
    if ((rc = persist_launch(e))) return rc;
  }
}


// Host-visible arenas: every authorised step has run and nothing is queued -- but a resident grid STAYS (its workers
// read and write the slabs only between seeing a ring entry and publishing their completion mark, and the pump's
// completion word is written after the marks it summarises: once it stands at p_next the slabs are the host's).
int quiesce(afe_engine *e) {
  join_streams(e);
  if (e->p_failed) return fail(e, AFE_ERR_HIP, e->err);     // (host-visible arenas' getters come through here)
  if (e->p_running) {
    volatile unsigned long long *st = p_status(e);
    const auto t0 = std::chrono::steady_clock::now();
    const bool deaf = afe_fault("sync_answer");      // (fault injection, dev-hooks builds: the host sees neither a park nor an answer)
    for (unsigned spins = 0;; spins++) {
      if (st[0] != 0 && !deaf) {                 // it has parked (idle host, or a stall): collect, finish what is left
        int rc = persist_collect(e);
        if (rc) return rc;
        if (e->p_resume > e->p_next) { e->p_failed = true; return fail(e, AFE_ERR_HIP, "persistent step kernel ran past the authorised steps"); }
        if (e->p_resume == e->p_next) break;
        if ((rc = persist_launch(e))) return rc;
        continue;
      }
      if (e->p_sync_posted == e->p_next + 1) {
        // A request for this count has been posted: from here on the ONLY way out is the workers' own answer (or the
        // grid's park, above).  The arrival counters of a launch are never reset -- a shard is complete at every multiple
        // of its size -- so a request that is left while only part of the workers have answered its marker (through the
        // pump's older word, say, with the slot then overwritten by the next real entry) would leave them misaligned for
        // the life of the grid, and a later request would be "answered" while workers are still stepping.
        if (st[AFE_PERSIST_SYNC_WORD] >= e->p_next && !deaf) break;
      } else {
        if (st[1] >= e->p_next && !deaf) break;    // the pump's sweep already says so: nothing posted, nothing to answer
        if (e->host_arena && e->p_workers <= AFE_PERSIST_HOST_MARKS && !deaf) {   // small grids write their marks here themselves: no request, no wait for the pump's sweep
          unsigned long long low = ~0ull;
          for (int w = 0; w < e->p_workers; w++) { const unsigned long long d = st[8 + w]; low = d < low ? d : low; }
          if (low >= e->p_next) break;
        } else {
          // ask the workers themselves (afe_device.h, sync marker): the pump's sweep over thousands of marks is tens of
          // microseconds old, a worker answers the moment its own count stands at the request
          __atomic_store_n(&e->p_host[AFE_PERSIST_HOST_RING + AFE_PERSIST_SYNCREQ_WORD], (unsigned long long)e->p_next, __ATOMIC_RELEASE);
          e->p_sync_posted = e->p_next + 1;
        }
      }
      if ((spins & 0xfffu) == 0xfffu && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(afe_sync_patience_s())) {
        // Nothing answered and nothing parked: the grid is not going to.  The engine is marked failed (every later call says
        // so at once instead of spinning another 20 s on the same posted request) and the request is withdrawn.
        char msg[200];
        std::snprintf(msg, sizeof(msg), "persistent step kernel makes no progress (waiting for step %llu; request posted for %llu, answered %llu, pump at %llu)",
                      (unsigned long long)e->p_next, (unsigned long long)(e->p_sync_posted ? e->p_sync_posted - 1 : 0),
                      (unsigned long long)st[AFE_PERSIST_SYNC_WORD], (unsigned long long)st[1]);
        e->p_failed = true;
        e->p_sync_posted = 0;
        return fail(e, AFE_ERR_HIP, msg);
      }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    e->p_quiesced = e->p_next;
  }
  if (!e->p_running && e->stream_pending) {
    AFE_HIP(e, hipStreamSynchronize(e->stream));
    e->stream_pending = false;
  }
  return AFE_OK;
}

long persist_refresh_steps() {
  static const long refresh = [] { const char *s = afe_dev_env("AFE_PERSIST_REFRESH_STEPS"); return s && *s ? std::atol(s) : 512L; }();
  return refresh;
}

// The pump parks itself after 200 us without news.  If it decides to while entries are being written, they would wait
// for a grid nobody starts before the next engine call -- a host that synchronises outside the engine
// (hipDeviceSynchronize) would see state short of the steps it asked for.  The pump announces where it means to park
// (status word 7) before it looks at that slot one last time; the host writes its entries before it looks at the
// announcement: one of the two sees the other.  Seen here: wait (microseconds) for the pump to take the entry after
// all or to leave, and in the second case start the grid that finishes the rest.
int persist_settle(afe_engine *e) {
  if (!e->p_running) return AFE_OK;
  volatile unsigned long long *st = p_status(e);
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; spins++) {
    const unsigned long long intent = st[7];
    if (st[0] != 0) break;                                        // it has left
    if (intent == 0 || intent - 1 >= e->p_next) return AFE_OK;    // not leaving, or leaving behind everything authorised
    if ((spins & 0xfffu) == 0xfffu && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) break;   // (collect below reports what is wrong)
  }
  int rc = persist_collect(e);
  if (rc) return rc;
  if (e->p_resume > e->p_next) { e->p_failed = true; return fail(e, AFE_ERR_HIP, "persistent step kernel ran past the authorised steps"); }
  if (e->p_resume < e->p_next) return persist_launch(e);
  return AFE_OK;
}

// Cache policy of the one-step launches (LaunchFlags::cache_policy).  Automatic: by what the Infinity Cache (256 MiB) can
// hold from one step to the next.  The distinct bytes a step touches (state + inputs + outputs: 104 B per fp32 vehicle of
// the bench workload) fit: default policy, the whole working set is served on-die.  The state alone fits, or nearly (the
// part that does still hits): inputs and outputs stream past it (nt).  Beyond: everything nt, one contiguous range per
// XCD.  Measured on the bench workload, two streams, us per step by policy 0 / 1 / 2 / 3 (tools/cache_policy_probe.py,
// profiles/r04_cache_policy.txt): 2^21 vehicles 42.3 / 46.1 / 53.6 / 54.7; 3 x 2^20 74.8 / 66.6 / 82.2 / 89.7; 2^22
// 115.8 / 91.3 / 108.4 / 106.4; 6 x 2^20 173.7 / 158.0 / 165.0 / 177.1; 2^23 231.7 / 225.9 / 219.6 / 214.8.
int resolve_cache_policy(const afe_engine *e) {
  static const int forced = [] { const char *s = afe_dev_env("AFE_CACHE_POLICY"); return s && *s ? std::atoi(s) : -1; }();
  const int asked = forced >= 0 && forced <= 3 ? forced : e->cache_policy;
  if (asked >= 0) return asked;
  if (e->host_arena) return 0;        // host memory: the policy bits mean nothing the bus honours
  const double es = (double)elem(e), n = (double)e->n;
  double state = 13 * es + (motor_lazy(e) ? 0 : 4 * es) + (e->noise && e->seed_policy != AFE_SEED_COUNTER ? 4 : 0) + (e->logic_on ? 12 * 4 + 4 * 4 : 0);
  double streams = (e->logic_on ? 4 * 4 : 4 * 4) + (e->has_ext_force ? 3 * es : 0) + (e->has_ext_torque ? 3 * es : 0) + 6 * 4;
  const double MiB = 1048576.0;
  if ((state + streams) * n <= 240 * MiB) return 0;
  if (state * n <= 384 * MiB) return 1;
  return 3;
}

// afe_step in persistent mode: n_steps more ring entries; a grid is started if none is resident
int persist_step(afe_engine *e, uint64_t dt_us, int n_steps) {
  int rc = persist_alloc(e);
  if (rc) return rc;
  const LaunchFlags f = persist_flags(e);
  if (e->p_running && (e->p_dt_us != dt_us || f.ext_force != e->p_flags.ext_force || f.noise != e->p_flags.noise || f.logic != e->p_flags.logic || f.counter_noise != e->p_flags.counter_noise))
    if ((rc = persist_park(e))) return rc;
  volatile unsigned long long *st = p_status(e);
  // A grid whose workers do NOT finish a block together steps faster when it is retired now and then: 2^20 vehicles on
  // the own queue, 2 000-step blocks, 19.6 -> 19.2 us per step with grids retired after 512 steps (tools/refresh_probe.py).
  // Not age (round 4's first reading; tools/ageing_probe.py shows a grid's windows getting no slower): the workers that
  // finish their share early -- those with two chunks where others have three -- LEAVE at a park entry and stop competing,
  // while in a grid that stays they poll the ring until the slowest is done.  Where issue priority keeps the workers
  // together (persist_launch) there is nobody to send home and a grid lives on (131 072 / 262 144 vehicles, retired against
  // not: 2.20 / 3.54 against 2.19 / 3.51).  Otherwise a grid that has served persist_refresh_steps() steps is retired here
  // -- the host waits for what it has authorised, ~20 us of dispatch follow -- and at the next afe_sync (afe_sync below).
  // AFE_PERSIST_REFRESH_STEPS=0: never.
  if (e->p_running && !e->p_prio && persist_refresh_steps() > 0 && (long)(e->p_next - e->p_launch_start) >= persist_refresh_steps() && (rc = persist_park(e))) return rc;
  if (e->p_running && st[0] != 0) {       // it parked itself (idle): collect it, a new grid starts below
    if ((rc = persist_collect(e))) return rc;
  }
  if (!e->p_running && e->p_resume == e->p_next) {
    // nothing pending: a new run of equally long steps begins here.  Step index -> start time is linear from now on
    // (a resident grid derives the gust epoch of each of its steps from it, and so does persist_launch for a grid that
    // has to pick up in the middle of the run).
    e->p_seg_start = e->p_next;
    e->p_seg_t0_us = e->now_us;
    e->p_seg_gust_applied = e->gust_applied;
    e->p_dt_us = dt_us; e->p_flags = f;
  }
  for (int s = 0; s < n_steps; s++) {
    // room in the host ring: never more than a ring (less a margin) ahead of the slowest worker
    for (unsigned spins = 0;; spins++) {
      const uint64_t floor_ = e->p_running ? std::max<uint64_t>(st[1], e->p_resume) : e->p_resume;
      if (e->p_next - floor_ < AFE_PERSIST_HOST_RING - 128) break;
      if (!e->p_running) {                // entries are waiting and nobody reads them
        if ((rc = persist_launch(e))) return rc;
      } else if (st[0] != 0) {
        if ((rc = persist_collect(e))) return rc;
      } else if (spins > 2000000000u) {
        return fail(e, AFE_ERR_HIP, "persistent step kernel makes no progress");
      }
    }
    if (e->gust_on) e->gust_applied = e->now_us / e->gust_period_us;   // what the slab holds once this step has run
    e->now_us += dt_us;  // ManualTimer::AdvanceMicroSeconds, main.cpp:392
    unsigned long long entry = (e->p_next + 1) << 2;
    if (gate_step(e->logic_period, e->logic_elapsed_us, dt_us)) { entry |= AFE_PERSIST_TICK; e->n_ticks++; }
    __atomic_store_n(&e->p_host[e->p_next & (AFE_PERSIST_HOST_RING - 1)], entry, __ATOMIC_RELEASE);
    e->p_next++;
    e->steps_issued++;
  }
  if (!e->p_running && (rc = persist_launch(e))) return rc;
  if ((rc = persist_settle(e))) return rc;
  if (motor_lazy(e)) e->motor_stale = true;
  return AFE_OK;
}

}  // namespace

// ---------------------------------------------------------------------------

static int create_engine(afe_engine **out, int64_t n_vehicles, int precision, int device, int64_t first_global_index, bool host_arena) {
  if (!out || n_vehicles <= 0 || (precision != AFE_F32 && precision != AFE_F64) || first_global_index < 0)
    return AFE_ERR_INVALID_ARG;
  // the kernels address a slab component as base + 32-bit byte offset
  if (n_vehicles > (precision == AFE_F64 ? (int64_t(1) << 29) : (int64_t(1) << 30)) - 256) return AFE_ERR_INVALID_ARG;
  *out = nullptr;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return AFE_ERR_NO_DEVICE;
  if (device < 0) {
    if (hipGetDevice(&device) != hipSuccess) return AFE_ERR_NO_DEVICE;
  }
  if (device >= n_dev) return AFE_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return AFE_ERR_NO_DEVICE;
  // the code object carries gfx950 ISA only
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return AFE_ERR_NO_DEVICE;

  afe_engine *e = new afe_engine();
  e->n = n_vehicles;
  // Component stride = 256 elements x an ODD count.  A power-of-two stride puts
  // the ~40 slab components a wave touches on the same HBM channels (measured on
  // MI355X with tools/stream_probe.hip: 6.0 TB/s at stride 2^20 vs 7.9 TB/s at
  // 2^20 + 256 for the same bytes).
  e->stride = (n_vehicles + 255) / 256 * 256;
  if ((e->stride / 256) % 2 == 0) e->stride += 256;
  e->precision = precision;
  e->device = device;
  e->first_global = first_global_index;

  auto bail = [&](const char *what, hipError_t err) {
    std::fprintf(stderr, "agrifly_engine: %s failed: %s\n", what, hipGetErrorString(err));
    afe_destroy(e);
    return AFE_ERR_HIP;
  };
  hipError_t err;
  if ((err = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", err);
  if ((err = hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking)) != hipSuccess)
    return bail("hipStreamCreate", err);
  e->stream = e->own_stream;

  // one arena, 256-B aligned slabs:
  //   state 17 comps + wrench 6 comps (elem size), cmd 4 + imu 6 floats, rng u32, type u8
  const size_t S = (size_t)e->stride, es = elem(e);
  const size_t bytes = S * (17 + 6) * es + S * (4 + 6) * 4 + S * 4 + S + S * 2 * sizeof(double) + 256 * sizeof(DevParams<double>);
  if (host_arena) {
    // pinned, coherent, device-mapped host memory: the kernels address it over the bus (a step of a small ensemble is a
    // handful of bus reads in flight at once), the host reads and writes it in place
    void *hp = nullptr;
    if ((err = hipHostMalloc(&hp, bytes, hipHostMallocCoherent | hipHostMallocMapped)) != hipSuccess) return bail("hipHostMalloc", err);
    e->arena_host = (char *)hp;
    e->host_arena = true;
    if ((err = hipHostGetDevicePointer(&e->arena, hp, 0)) != hipSuccess) return bail("hipHostGetDevicePointer", err);
  } else if ((err = hipMalloc(&e->arena, bytes)) != hipSuccess) return bail("hipMalloc", err);
  e->arena_bytes = bytes;
  if ((err = hipMemsetAsync(e->arena, 0, bytes, main_stream(e))) != hipSuccess) return bail("hipMemset", err);
  char *p = (char *)e->arena;
  auto carve = [&](size_t nbytes) { void *r = p; p += nbytes; return r; };
  e->pos = carve(3 * S * es);
  e->vel = carve(3 * S * es);
  e->att = carve(4 * S * es);
  e->ang_vel = carve(3 * S * es);
  e->motor = carve(4 * S * es);
  e->ext_force = carve(3 * S * es);
  e->ext_torque = carve(3 * S * es);
  e->cmd = (float *)carve(4 * S * 4);
  e->gyro = (float *)carve(3 * S * 4);
  e->acc = (float *)carve(3 * S * 4);
  e->rng = (uint32_t *)carve(S * 4);
  e->type = (uint8_t *)carve(S);
  e->kernel_bytes = (size_t)(p - (char *)e->arena);
  e->anchor = (double *)carve(S * 2 * sizeof(double));
  e->dev_table = carve(256 * sizeof(DevParams<double>));

  // identity attitude (SimulationObject6DOF.hpp:17): w component = 1
  {
    std::vector<char> ones(S * es);
    for (size_t k = 0; k < S; k++) {
      if (precision == AFE_F64) ((double *)ones.data())[k] = 1.0;
      else ((float *)ones.data())[k] = 1.0f;
    }
    if ((err = hipMemcpyAsync(e->att, ones.data(), S * es, hipMemcpyHostToDevice, main_stream(e))) != hipSuccess)
      return bail("hipMemcpy", err);
    if ((err = hipStreamSynchronize(main_stream(e))) != hipSuccess) return bail("hipStreamSynchronize", err);
  }
  if (launch_seed_rng(e->rng, e->n, e->first_global, e->seed_policy, main_stream(e)) != 0)
    return bail("seed kernel launch (is the gfx950 code object present?)", hipGetLastError());
  if ((err = hipStreamSynchronize(main_stream(e))) != hipSuccess) return bail("seed kernel", err);
  if (const char *fm = std::getenv("AFE_FORCE_STEP_MODE")) {   // test hook: every engine of this process steps by the resident grid where it can (1 persistent, 3 resident state)
    const int m = std::atoi(fm);
    if (m >= AFE_STEP_LAUNCH && m <= AFE_STEP_RESIDENT) e->step_mode = m;
  }
  if (e->host_arena) e->step_mode = AFE_STEP_AUTO;   // what such an engine is for: the host in the loop of a small ensemble
  if (std::getenv("AFE_FORCE_SPLIT")) {   // test hook: every engine of this process steps split (tests/: the whole GPU suite runs this way once)
    const int src = afe_set_split_stepping(e, 2);
    if (src) { afe_destroy(e); return src; }
  }
  *out = e;
  return AFE_OK;
}

extern "C" int afe_create(afe_engine **out, int64_t n_vehicles, int precision, int device, int64_t first_global_index) {
  // test hook: every engine of this process up to 16 384 vehicles keeps its arena in host memory (tests/: the GPU suite
  // runs this way once; the large ensembles of the suite would spend the day on the bus)
  return create_engine(out, n_vehicles, precision, device, first_global_index, n_vehicles <= 16384 && std::getenv("AFE_FORCE_HOST_ARENA") != nullptr);
}
extern "C" int afe_create_host_visible(afe_engine **out, int64_t n_vehicles, int precision, int device, int64_t first_global_index) {
  return create_engine(out, n_vehicles, precision, device, first_global_index, true);
}

extern "C" int afe_destroy(afe_engine *e) {
  if (!e) return AFE_ERR_INVALID_ARG;
  (void)hipSetDevice(e->device);
  if (e->p_running) (void)persist_park(e);
  if (e->aql) {
    const bool left = afe::aql_close(e->aql);     // waits (30 s) for a grid that did not take the park
    e->aql = nullptr;
    if (!left) {
      // the grid is still on the device and may write the rings and the slabs: they are leaked, not freed under it
      // (the engine object too: its status block is what the message of the next call would read)
      std::fprintf(stderr, "agrifly_engine: afe_destroy: a resident grid is still running; the engine's device memory is leaked\n");
      return AFE_ERR_HIP;
    }
  }
  if (e->p_dev) (void)hipFree(e->p_dev);
  if (e->p_host) (void)hipHostFree(e->p_host);
  if (e->side_stream) (void)hipStreamSynchronize(e->side_stream);
  if (e->own_stream) (void)hipStreamSynchronize(e->own_stream);
  if (e->ev_main) (void)hipEventDestroy(e->ev_main);
  if (e->ev_side) (void)hipEventDestroy(e->ev_side);
  if (e->side_stream) (void)hipStreamDestroy(e->side_stream);
  if (e->query_stream) (void)hipStreamSynchronize(e->query_stream);
  if (e->ev_q_start) (void)hipEventDestroy(e->ev_q_start);
  if (e->ev_q_done) (void)hipEventDestroy(e->ev_q_done);
  if (e->query_stream) (void)hipStreamDestroy(e->query_stream);
  if (e->world) world_destroy(e->world);
  if (e->pack_scratch) (void)hipFree(e->pack_scratch);
  if (e->host_arena) { if (e->logic_arena_host) (void)hipHostFree(e->logic_arena_host); }
  else if (e->logic_arena) (void)hipFree(e->logic_arena);
  if (e->host_arena) { if (e->arena_host) (void)hipHostFree(e->arena_host); }
  else if (e->arena) (void)hipFree(e->arena);
  if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
  delete e;
  return AFE_OK;
}

extern "C" const char *afe_last_error(const afe_engine *e) { return e ? e->err.c_str() : "null engine"; }

extern "C" int afe_set_stream(afe_engine *e, void *hip_stream) {
  if (!e) return AFE_ERR_INVALID_ARG;
  AFE_HIP(e, hipSetDevice(e->device));
  AFE_HIP(e, hipStreamSynchronize(main_stream(e)));   // (joins the side stream first)
  e->stream = hip_stream ? (hipStream_t)hip_stream : e->own_stream;
  return AFE_OK;
}

extern "C" int afe_set_type_table(afe_engine *e, const afe_vehicle_params *table, int n_types) {
  if (e) { const int mrc = materialize_motor(e); if (mrc) return mrc; }   // with the parameters the last step used
  if (!e || !table || n_types < 1 || n_types > 256) return fail(e, AFE_ERR_INVALID_ARG, "type table must hold 1..256 records");
  std::vector<HostParams> t((size_t)n_types);
  for (int k = 0; k < n_types; k++) {
    const char *why = "";
    int rc = expand_params(table[k], t[(size_t)k], &why);
    if (rc) return fail(e, rc, "type " + std::to_string(k) + ": " + why);
  }
  e->table.swap(t);
  e->table_dirty = true;
  return AFE_OK;
}

// what the step launcher may assume about the type slab: all on record 0 (parameters ride in the kernel
// arguments), or at least one type per wave (scalar loads of the wave's record), or neither (LDS table)
// Kept incrementally -- per aligned run of 64 vehicles: how many are off record 0 and whether the run is
// mixed -- so that a caller who sets the types one vehicle at a time pays for the runs it touches, not for
// the ensemble (first = 0, count = n: everything).
static void refresh_type_flags(afe_engine *e, int64_t first = 0, int64_t count = -1) {
  const int64_t n = e->n, n_runs = (n + 63) / 64;
  if (e->type_host.size() != (size_t)n) e->type_host.assign((size_t)n, 0);
  if (e->run_nonzero.size() != (size_t)n_runs) {
    e->run_nonzero.assign((size_t)n_runs, 0);
    e->run_mixed.assign((size_t)n_runs, 0);
    e->n_nonzero = e->n_mixed_runs = 0;
    first = 0; count = n;
  }
  if (count < 0) count = n - first;
  if (count == 0) return;
  for (int64_t r = first / 64; r <= (first + count - 1) / 64; r++) {
    const int64_t a = r * 64, b = std::min(n, a + 64);
    int nonzero = 0, mixed = 0;
    for (int64_t k = a; k < b; k++) {
      nonzero += e->type_host[(size_t)k] != 0;
      mixed |= e->type_host[(size_t)k] != e->type_host[(size_t)a];
    }
    e->n_nonzero += nonzero - (int)e->run_nonzero[(size_t)r];
    e->n_mixed_runs += mixed - (int)e->run_mixed[(size_t)r];
    e->run_nonzero[(size_t)r] = (uint8_t)nonzero;
    e->run_mixed[(size_t)r] = (uint8_t)mixed;
  }
  e->types_uniform = e->n_nonzero == 0;
  e->types_wave_uniform = e->n_mixed_runs == 0;
}

extern "C" int afe_set_vehicle_types(afe_engine *e, int64_t first, int64_t count, const uint8_t *type_index) {
  int rc = check_range(e, first, count);
  if (rc) return rc;
  if ((rc = materialize_motor(e))) return rc;
  if (!type_index) return fail(e, AFE_ERR_INVALID_ARG, "type_index is NULL");
  for (int64_t k = 0; k < count; k++)
    if (type_index[k] >= e->table.size())
      return fail(e, AFE_ERR_INVALID_ARG, "type index " + std::to_string(type_index[k]) + " of vehicle " +
                                              std::to_string(first + k) + " is outside the type table");
  AFE_HIP(e, hipSetDevice(e->device));
  if ((rc = copy_in(e, e->type, 1, 1, first, count, type_index))) return rc;
  if (e->type_host.size() != (size_t)e->n) e->type_host.assign((size_t)e->n, 0);
  std::memcpy(e->type_host.data() + first, type_index, (size_t)count);
  refresh_type_flags(e, first, count);
  return AFE_OK;
}

extern "C" int afe_set_logic_period(afe_engine *e, double seconds) {
  if (!e || !(seconds > 0) || !std::isfinite(seconds)) return fail(e, AFE_ERR_INVALID_ARG, "logic period must be > 0");
  e->logic_period = seconds;
  return AFE_OK;
}

extern "C" int afe_set_imu_noise(afe_engine *e, int enabled, double sigma_gyro, double sigma_acc, int seed_policy) {
  if (!e || !(sigma_gyro >= 0) || !(sigma_acc >= 0) ||
      (seed_policy != AFE_SEED_REFERENCE && seed_policy != AFE_SEED_DECORRELATED && seed_policy != AFE_SEED_COUNTER))
    return fail(e, AFE_ERR_INVALID_ARG, "bad noise configuration");
  { const int prc = persist_park(e); if (prc) return prc; }   // a resident grid carries the old configuration
  e->noise = enabled != 0;
  e->sigma_gyro = sigma_gyro;
  e->sigma_acc = sigma_acc;
  if (seed_policy != e->seed_policy) {
    e->seed_policy = seed_policy;
    AFE_HIP(e, hipSetDevice(e->device));
    // (the counter policy keeps no per-vehicle word; the slab stays what it was -- valid minstd_rand0 words)
    if (seed_policy != AFE_SEED_COUNTER && launch_seed_rng(e->rng, e->n, e->first_global, seed_policy, main_stream(e)) != 0)
      return fail(e, AFE_ERR_HIP, "seed kernel launch failed");
  }
  return AFE_OK;
}

extern "C" int afe_set_state(afe_engine *e, int64_t first, int64_t count, const double *pos3,
                             const double *vel3, const double *att4, const double *ang_vel3,
                             const double *motor_speed4) {
  return set_state_any<double>(e, first, count, pos3, vel3, att4, ang_vel3, motor_speed4);
}
extern "C" int afe_get_state(afe_engine *e, int64_t first, int64_t count, double *pos3, double *vel3,
                             double *att4, double *ang_vel3, double *motor_speed4) {
  return get_state_any<double>(e, first, count, pos3, vel3, att4, ang_vel3, motor_speed4);
}
extern "C" int afe_set_state_f32(afe_engine *e, int64_t first, int64_t count, const float *pos3,
                                 const float *vel3, const float *att4, const float *ang_vel3,
                                 const float *motor_speed4) {
  return set_state_any<float>(e, first, count, pos3, vel3, att4, ang_vel3, motor_speed4);
}
extern "C" int afe_get_state_f32(afe_engine *e, int64_t first, int64_t count, float *pos3, float *vel3,
                                 float *att4, float *ang_vel3, float *motor_speed4) {
  return get_state_any<float>(e, first, count, pos3, vel3, att4, ang_vel3, motor_speed4);
}

extern "C" int afe_set_rng_state(afe_engine *e, int64_t first, int64_t count, const uint32_t *state) {
  int rc = check_range(e, first, count);
  if (rc) return rc;
  if (!state) return fail(e, AFE_ERR_INVALID_ARG, "state is NULL");
  for (int64_t k = 0; k < count; k++)
    if (state[k] == 0 || state[k] >= 2147483647u) return fail(e, AFE_ERR_INVALID_ARG, "minstd_rand0 state must be in [1, 2^31-2]");
  AFE_HIP(e, hipSetDevice(e->device));
  return copy_in(e, e->rng, 4, 1, first, count, state);
}
extern "C" int afe_get_rng_state(afe_engine *e, int64_t first, int64_t count, uint32_t *state) {
  int rc = check_range(e, first, count);
  if (rc) return rc;
  if (!state) return fail(e, AFE_ERR_INVALID_ARG, "state is NULL");
  AFE_HIP(e, hipSetDevice(e->device));
  return copy_out(e, e->rng, 4, 1, first, count, state);
}

extern "C" int afe_set_motor_cmds(afe_engine *e, int64_t first, int64_t count, const float *cmd4) {
  int rc = check_range(e, first, count);
  if (rc) return rc;
  if (!cmd4) return fail(e, AFE_ERR_INVALID_ARG, "cmd4 is NULL");
  AFE_HIP(e, hipSetDevice(e->device));
  if ((rc = materialize_motor(e))) return rc;   // the speeds of the last step come from the OLD commands
  return copy_in(e, e->cmd, 4, 4, first, count, cmd4);
}
extern "C" int afe_get_motor_cmds(afe_engine *e, int64_t first, int64_t count, float *cmd4) {
  int rc = check_range(e, first, count);
  if (rc) return rc;
  if (!cmd4) return fail(e, AFE_ERR_INVALID_ARG, "cmd4 is NULL");
  AFE_HIP(e, hipSetDevice(e->device));
  return copy_out(e, e->cmd, 4, 4, first, count, cmd4);
}

extern "C" int afe_set_rates_logic(afe_engine *e, const afe_rates_logic_params *table, int n_types) {
  if (!e) return AFE_ERR_INVALID_ARG;
  AFE_HIP(e, hipSetDevice(e->device));
  { const int mrc = materialize_motor(e); if (mrc) return mrc; }   // from here on the logic rewrites the commands
  if (!table) {
    AFE_HIP(e, hipStreamSynchronize(main_stream(e)));
    e->logic_on = false;
    return AFE_OK;
  }
  if (e->table.empty() || n_types != (int)e->table.size())
    return fail(e, AFE_ERR_INVALID_ARG, "the logic table must have one record per vehicle type record");
  std::vector<DevLogic> probe((size_t)n_types);
  for (int k = 0; k < n_types; k++) {
    const char *why = "";
    int rc = expand_logic(table[k], (float)e->logic_period, probe[(size_t)k], &why);
    if (rc) return fail(e, rc, "logic type " + std::to_string(k) + ": " + why);
  }
  const size_t S = (size_t)e->stride;
  if (!e->logic_arena) {
    const size_t bytes = S * 12 * 4 + S * 4 * 4 + S * 2 + 256 * sizeof(DevLogic);
    if (e->host_arena) {
      void *hp = nullptr;
      AFE_HIP(e, hipHostMalloc(&hp, bytes, hipHostMallocCoherent | hipHostMallocMapped));
      e->logic_arena_host = (char *)hp;
      AFE_HIP(e, hipHostGetDevicePointer(&e->logic_arena, hp, 0));
    } else AFE_HIP(e, hipMalloc(&e->logic_arena, bytes));
    e->logic_arena_alloc = bytes;
    char *p = (char *)e->logic_arena;
    e->lpf = (float *)p; p += S * 12 * 4;
    e->rates_cmd = (float *)p; p += S * 4 * 4;
    e->dev_logic_table = (DevLogic *)p; p += 256 * sizeof(DevLogic);
    e->have_cmd = (uint8_t *)p; p += S;
    e->imu_init = (uint8_t *)p;
  }
  // QuadcopterLogic::Initialise: filters at 0 (QuadcopterLogic.cpp:38,133), _kf.Reset(), FS_IDLE
  {
    int frc;
    if ((frc = fill_rows(e, e->lpf, 4, 12, 0, (int64_t)S, 0)) || (frc = fill_rows(e, e->rates_cmd, 4, 4, 0, (int64_t)S, 0)) ||
        (frc = fill_rows(e, e->have_cmd, 1, 1, 0, (int64_t)S, 0)) || (frc = fill_rows(e, e->imu_init, 1, 1, 0, (int64_t)S, 0)) ||
        (frc = fill_rows(e, e->cmd, 4, 4, 0, (int64_t)S, 0)))
      return frc;
  }
  e->logic_params.assign(table, table + n_types);
  e->logic_table_period = -1.0f;
  e->logic_on = true;
  return refresh_logic(e);
}

extern "C" int afe_set_rates_commands(afe_engine *e, int64_t first, int64_t count, const float *thrust_norm,
                                      const float *ang_vel3) {
  int rc = check_range(e, first, count);
  if (rc) return rc;
  if (!e->logic_on) return fail(e, AFE_ERR_NOT_CONFIGURED, "afe_set_rates_logic has not been called");
  if (!thrust_norm || !ang_vel3) return fail(e, AFE_ERR_INVALID_ARG, "command arrays are NULL");
  AFE_HIP(e, hipSetDevice(e->device));
  if ((rc = copy_in(e, e->rates_cmd, 4, 1, first, count, thrust_norm))) return rc;
  if ((rc = copy_in(e, e->rates_cmd + e->stride, 4, 3, first, count, ang_vel3))) return rc;
  return fill_rows(e, e->have_cmd, 1, 1, first, count, 1);
}

extern "C" int afe_set_commands_from_radio(afe_engine *e, int64_t first, int64_t count, const uint8_t *raw_packets) {
  int rc = check_range(e, first, count);
  if (rc) return rc;
  if (!e->logic_on) return fail(e, AFE_ERR_NOT_CONFIGURED, "afe_set_rates_logic has not been called");
  if (!raw_packets) return fail(e, AFE_ERR_INVALID_ARG, "raw_packets is NULL");
  std::vector<float> cmd((size_t)(4 * count));
  std::vector<uint8_t> have((size_t)count);
  for (int64_t k = 0; k < count; k++) {
    afe_radio_message m;
    afe_radio_decode(raw_packets + k * AFE_RADIO_PACKET_SIZE, &m);
    if (m.type == 5) {  // externalRatesCmd -> FS_EXTERNAL_RATES_CONTROL (QuadcopterLogic.cpp:293-295)
      have[(size_t)k] = 1;
      for (int c = 0; c < 4; c++) cmd[(size_t)(c * count + k)] = m.floats[c];
    } else if (m.type == 6 || m.type == 2) {  // idle / kill: motors off
      have[(size_t)k] = 0;
      for (int c = 0; c < 4; c++) cmd[(size_t)(c * count + k)] = 0.0f;
    } else {
      return fail(e, AFE_ERR_INVALID_ARG, "radio message type " + std::to_string(m.type) + " for vehicle " +
                                              std::to_string(first + k) + " needs the host-side logic");
    }
  }
  AFE_HIP(e, hipSetDevice(e->device));
  if ((rc = copy_in(e, e->rates_cmd, 4, 4, first, count, cmd.data()))) return rc;
  return copy_in(e, e->have_cmd, 1, 1, first, count, have.data());
}

extern "C" int afe_set_external_force(afe_engine *e, int64_t first, int64_t count, const double *force3) {
  if (!e) return AFE_ERR_INVALID_ARG;
  return set_wrench(e, e->ext_force, e->has_ext_force, first, count, force3);
}
extern "C" int afe_set_external_torque(afe_engine *e, int64_t first, int64_t count, const double *torque3) {
  if (!e) return AFE_ERR_INVALID_ARG;
  return set_wrench(e, e->ext_torque, e->has_ext_torque, first, count, torque3);
}

extern "C" int afe_step(afe_engine *e, uint64_t dt_us, int n_steps) {
  if (!e || n_steps < 0) return fail(e, AFE_ERR_INVALID_ARG, "n_steps must be >= 0");
  if (e->table.empty()) return fail(e, AFE_ERR_NOT_CONFIGURED, "afe_set_type_table has not been called");
  if (n_steps == 0) return AFE_OK;
  const double dt = us_to_seconds(dt_us);  // Timer::GetSeconds<double>, Timer.hpp:36-38
  if (dt < 1e-6) return AFE_OK;            // Quadcopter_T.cpp:88-90
  AFE_HIP(e, hipSetDevice(e->device));
  int rc = refresh_table(e, dt);
  if (rc) return rc;
  if (e->logic_on && (rc = refresh_logic(e))) return rc;
  if (e->p_failed) {
    if (e->p_fail_msg.empty()) e->p_fail_msg = e->err;      // the first refusal after the failure: err still holds its text
    return fail(e, AFE_ERR_HIP, "a persistent step kernel failed earlier (" + e->p_fail_msg + "); create a new engine");
  }
  // AFE_STEP_AUTO and a call that asks for several steps at once: nobody can look at the steps in between, so the fused
  // launch (state in registers from step to step, one load and one store per call) is the faster way to the same bits --
  // from 2 steps per call at 2^19 vehicles and more, from 8 at any size (measured: bench.py sweep, fused2 / fused50
  // against the resident grid).  One step per call stays with the resident grid.
  const bool fuse = e->step_mode == AFE_STEP_AUTO && e->max_fused > 1 && (n_steps >= 8 || (n_steps >= 2 && e->n >= (int64_t(1) << 19)));
  if (!fuse && persist_eligible(e)) return persist_step(e, dt_us, n_steps);
  if (e->p_running && (rc = persist_park(e))) return rc;
  LaunchFlags f;
  f.ext_force = e->has_ext_force;
  f.ext_torque = e->has_ext_torque;
  f.noise = e->noise;
  f.logic = e->logic_on;
  int done = 0;
  while (done < n_steps) {
    int chunk = (n_steps - done) < e->max_fused ? (n_steps - done) : e->max_fused;
    if (e->gust_on) {
      // the force is constant within an epoch of the gust process: resample (one small launch) when this launch's first
      // step starts in another epoch than the slab holds, and end the launch before the next epoch begins
      const uint64_t ep = e->now_us / e->gust_period_us;
      if (ep != e->gust_applied && (rc = gust_resample(e, ep))) return rc;
      const uint64_t left = ((ep + 1) * e->gust_period_us - e->now_us + dt_us - 1) / dt_us;   // steps that start inside this epoch
      if ((uint64_t)chunk > left) chunk = (int)left;
    }
    const uint64_t tick_base = e->n_ticks;
    unsigned long long mask = 0;
    for (int s = 0; s < chunk; s++) {
      e->now_us += dt_us;  // ManualTimer::AdvanceMicroSeconds, main.cpp:392
      if (gate_step(e->logic_period, e->logic_elapsed_us, dt_us)) {
        mask |= (1ull << s);
        e->n_ticks++;
      }
    }
    // launches without a logic tick draw no noise: use the lean instantiation
    f.noise = e->noise && mask != 0;
    f.counter_noise = e->seed_policy == AFE_SEED_COUNTER;
    f.logic = e->logic_on && mask != 0;
    f.wave_uniform_types = !e->types_uniform && e->types_wave_uniform;
    f.cache_policy = resolve_cache_policy(e);
    const DevLogic *ulogic = (e->logic_on && e->types_uniform) ? &e->logic_table[0] : nullptr;
    // split stepping: vehicles [0, half) on the main stream, [half, n) on the side stream -- the two chains of launches
    // never wait for each other, so each one's drain-and-dispatch gap is covered by the other's streaming
    const bool split = e->n >= 1024 && (e->split_parts == 2 || (e->split_parts == 0 && e->stream == e->own_stream && e->n >= (int64_t(1) << 19)));
    if (split && !e->side_stream) {
      AFE_HIP(e, hipStreamCreateWithFlags(&e->side_stream, hipStreamNonBlocking));
      AFE_HIP(e, hipEventCreateWithFlags(&e->ev_main, hipEventDisableTiming));
      AFE_HIP(e, hipEventCreateWithFlags(&e->ev_side, hipEventDisableTiming));
    }
    const int64_t half = split ? ((e->n / 2) & ~int64_t(255)) : e->n;
    if (split && !e->split_dirty) {   // the side stream first sees everything the main stream has been given so far
      AFE_HIP(e, hipEventRecord(e->ev_main, e->stream));
      AFE_HIP(e, hipStreamWaitEvent(e->side_stream, e->ev_main, 0));
      e->split_dirty = true;
    }
    int lrc;
    if (e->precision == AFE_F64) {
      StepView<double> v;
      fill_view(e, v);
      v.dt = dt; v.inv_dt = 1.0 / dt; v.n_steps = chunk; v.tick_mask = mask; v.tick_base = tick_base;
      v.end = half;
      lrc = launch_step_f64(v, f, e->types_uniform ? &e->table_f64[0] : nullptr, ulogic, e->stream);
      if (split && lrc == 0) {
        v.first = half; v.end = e->n;
        lrc = launch_step_f64(v, f, e->types_uniform ? &e->table_f64[0] : nullptr, ulogic, e->side_stream);
      }
    } else {
      StepView<float> v;
      fill_view(e, v);
      v.dt = (float)dt; v.inv_dt = (float)(1.0 / dt); v.n_steps = chunk; v.tick_mask = mask; v.tick_base = tick_base;
      v.end = half;
      lrc = launch_step_f32(v, f, e->types_uniform ? &e->table_f32[0] : nullptr, ulogic, e->stream);
      if (split && lrc == 0) {
        v.first = half; v.end = e->n;
        lrc = launch_step_f32(v, f, e->types_uniform ? &e->table_f32[0] : nullptr, ulogic, e->side_stream);
      }
    }
    if (lrc != 0) return fail(e, AFE_ERR_HIP, std::string("step kernel launch: ") + hipGetErrorString((hipError_t)lrc));
    e->stream_pending = true;
    if (motor_lazy(e)) e->motor_stale = true;
    done += chunk;
    e->steps_issued += (uint64_t)chunk;
  }
  return AFE_OK;
}

extern "C" int afe_set_noise_seed(afe_engine *e, uint64_t seed) {
  if (!e) return AFE_ERR_INVALID_ARG;
  const int rc = persist_park(e);
  if (rc) return rc;
  e->noise_seed = seed;
  return AFE_OK;
}

extern "C" int afe_set_gust_process(afe_engine *e, int enabled, uint64_t seed, double sigma_max, uint64_t period_us, int64_t n_global) {
  if (!e) return AFE_ERR_INVALID_ARG;
  if (enabled && (!(sigma_max >= 0) || !std::isfinite(sigma_max) || period_us == 0 || (n_global > 0 && n_global < e->first_global + e->n)))
    return fail(e, AFE_ERR_INVALID_ARG, "gust process: sigma_max >= 0, period > 0, n_global >= first_global_index + n_vehicles (or 0: this ensemble alone)");
  AFE_HIP(e, hipSetDevice(e->device));
  const int rc = persist_park(e);
  if (rc) return rc;
  e->gust_on = enabled != 0;
  e->gust_seed = seed;
  e->gust_sigma_max = sigma_max;
  e->gust_period_us = period_us;
  e->gust_n_global = (uint64_t)(n_global > 0 ? n_global : e->first_global + e->n);
  e->gust_applied = ~0ull;              // the next step resamples whatever the slab holds
  if (e->gust_on) e->has_ext_force = true;
  return AFE_OK;
}

extern "C" int afe_get_external_force(afe_engine *e, int64_t first, int64_t count, double *force3) {
  int rc = check_range(e, first, count);
  if (rc) return rc;
  if (!force3) return fail(e, AFE_ERR_INVALID_ARG, "force3 is NULL");
  AFE_HIP(e, hipSetDevice(e->device));
  if (e->precision == AFE_F64) return get_field<double>(e, e->ext_force, 3, first, count, force3);
  return get_field<float>(e, e->ext_force, 3, first, count, force3);
}

extern "C" int afe_set_step_mode(afe_engine *e, int mode) {
  if (!e || mode < AFE_STEP_LAUNCH || mode > AFE_STEP_RESIDENT) return fail(e, AFE_ERR_INVALID_ARG, "step mode: 0 (launches), 1 (persistent), 2 (automatic) or 3 (resident state)");
  AFE_HIP(e, hipSetDevice(e->device));
  const int rc = persist_park(e);
  if (rc) return rc;
  e->step_mode = mode;
  return AFE_OK;
}

extern "C" int afe_steps_completed(afe_engine *e, uint64_t *steps) {
  if (!e || !steps) return AFE_ERR_INVALID_ARG;
  uint64_t pending = 0;
  if (e->p_running && p_status(e)[0] != 0) {
    // the grid has parked itself (a quiet host): steps authorised around that moment wait for a new grid -- start it here,
    // or a host that only watches this word would watch for ever
    AFE_HIP(e, hipSetDevice(e->device));
    int rc = persist_collect(e);
    if (rc) return rc;
    if (e->p_resume < e->p_next && (rc = persist_launch(e))) return rc;
  }
  if (e->p_running) {
    // (the pump's sweep over the workers' marks, or the workers' own answer to a sync request -- whichever is further)
    const uint64_t seen = std::max<uint64_t>(std::max<uint64_t>(std::max<uint64_t>(p_status(e)[1], p_status(e)[AFE_PERSIST_SYNC_WORD]), e->p_resume), e->p_quiesced);
    pending = e->p_next - std::min<uint64_t>(seen, e->p_next);
  }
  *steps = e->steps_issued - pending;
  return AFE_OK;
}

extern "C" int afe_persistent_running(const afe_engine *e, int *running) {
  if (!e || !running) return AFE_ERR_INVALID_ARG;
  *running = e->p_running ? 1 : 0;
  return AFE_OK;
}

extern "C" int afe_step_kernel_info(const afe_engine *e, int *record_path, int *addressing) {
  if (!e) return AFE_ERR_INVALID_ARG;
  if (record_path) *record_path = e->types_uniform ? 0 : (e->types_wave_uniform ? 1 : 2);
  if (addressing) {
    const bool fits = !e->force_global_addressing && e->kernel_bytes < 0xffff0000ull && logic_arena_bytes(e) < 0xffff0000ull;
    *addressing = fits ? 0 : 1;
  }
  return AFE_OK;
}

extern "C" int afe_set_addressing(afe_engine *e, int mode) {
  if (!e || (mode != 0 && mode != 1)) return fail(e, AFE_ERR_INVALID_ARG, "addressing mode must be 0 (automatic) or 1 (global)");
  e->force_global_addressing = mode == 1;
  return AFE_OK;
}

extern "C" int afe_set_split_stepping(afe_engine *e, int parts) {
  if (!e || parts < 0 || parts > 2) return fail(e, AFE_ERR_INVALID_ARG, "split stepping: 0 (automatic), 1 (off) or 2 parts");
  AFE_HIP(e, hipSetDevice(e->device));
  join_streams(e);
  e->split_parts = parts;
  return AFE_OK;
}

extern "C" int afe_set_cache_policy(afe_engine *e, int policy) {
  if (!e || policy < -1 || policy > 3) return fail(e, AFE_ERR_INVALID_ARG, "cache policy: -1 (automatic), 0 (default), 1 (inputs and outputs nt), 2 (everything nt) or 3 (everything nt, one range per XCD)");
  e->cache_policy = policy;
  return AFE_OK;
}

extern "C" int afe_set_resident_queue(afe_engine *e, int mode) {
  if (!e || mode < -1 || mode > 1) return fail(e, AFE_ERR_INVALID_ARG, "resident queue: -1 (automatic), 0 (the HIP stream) or 1 (the engine's own queue)");
  AFE_HIP(e, hipSetDevice(e->device));
  const int rc = persist_park(e);
  if (rc) return rc;
  e->aql_mode = mode;
  return AFE_OK;
}

extern "C" int afe_cache_policy_in_use(const afe_engine *e, int *policy) {
  if (!e || !policy) return AFE_ERR_INVALID_ARG;
  *policy = resolve_cache_policy(e);
  return AFE_OK;
}

extern "C" int afe_set_max_fused_steps(afe_engine *e, int k) {
  if (!e || k < 1 || k > 64) return fail(e, AFE_ERR_INVALID_ARG, "max fused steps must be in 1..64");
  e->max_fused = k;
  return AFE_OK;
}

extern "C" int afe_steps_until_tick(const afe_engine *e, uint64_t dt_us, int *n_steps) {
  if (!e || !n_steps) return AFE_ERR_INVALID_ARG;
  if (us_to_seconds(dt_us) < 1e-6) return AFE_ERR_INVALID_ARG;
  uint64_t el = e->logic_elapsed_us;
  int k = 0;
  for (;;) {
    k++;
    if (gate_step(e->logic_period, el, dt_us)) break;
    if (k >= (1 << 30)) return AFE_ERR_INVALID_ARG;
  }
  *n_steps = k;
  return AFE_OK;
}

extern "C" int afe_sync(afe_engine *e) {
  if (!e) return AFE_ERR_INVALID_ARG;
  AFE_HIP(e, hipSetDevice(e->device));
  if (e->p_failed) return fail(e, AFE_ERR_HIP, e->err);     // a grid that gave up, did not come back or never answered: said at once, nothing is waited for again
  if (e->p_running && e->p_on_aql && !e->view_exported &&
      !(!e->p_prio && persist_refresh_steps() > 0 && (long)(e->p_next - e->p_launch_start) >= persist_refresh_steps())) {     // (an aged grid is retired: persist_step)
    // Every authorised step has run and its stores are acknowledged; the grid STAYS (it lives on the engine's own queue,
    // which no HIP synchronisation waits for) and takes the next afe_step without a launch.  Whoever reads the state does
    // so through an entry point of the engine, which ends the grid first (kernel end = the caches written back).  Once
    // afe_get_device_view has handed the slabs out, afe_sync ends the grid as it always did: a reader the engine does
    // not know about must find them in memory.
    const int rc = quiesce(e);
    if (rc) return rc;
    if (e->p_failed) return fail(e, AFE_ERR_HIP, e->err);
    return AFE_OK;
  }
  AFE_HIP(e, hipStreamSynchronize(main_stream(e)));
  if (e->p_failed) return fail(e, AFE_ERR_HIP, e->err);
  return AFE_OK;
}

extern "C" int afe_grid_time(afe_engine *e, uint64_t *device_ns, uint64_t *steps) {
  if (!e || !device_ns || !steps) return AFE_ERR_INVALID_ARG;
  AFE_HIP(e, hipSetDevice(e->device));
  const int rc = persist_park(e);      // the grid now resident is counted too
  if (rc) return rc;
  *device_ns = e->p_grid_ns;
  *steps = e->p_grid_steps;
  e->p_grid_ns = 0; e->p_grid_steps = 0;
  return AFE_OK;
}

extern "C" int afe_time_us(const afe_engine *e, uint64_t *now_us) {
  if (!e || !now_us) return AFE_ERR_INVALID_ARG;
  *now_us = e->now_us;
  return AFE_OK;
}
extern "C" int afe_logic_ticks(const afe_engine *e, uint64_t *n_ticks) {
  if (!e || !n_ticks) return AFE_ERR_INVALID_ARG;
  *n_ticks = e->n_ticks;
  return AFE_OK;
}

extern "C" int afe_get_imu(afe_engine *e, int64_t first, int64_t count, float *gyro3, float *acc3) {
  int rc = check_range(e, first, count);
  if (rc) return rc;
  AFE_HIP(e, hipSetDevice(e->device));
  if ((rc = copy_out(e, e->gyro, 4, 3, first, count, gyro3))) return rc;
  return copy_out(e, e->acc, 4, 3, first, count, acc3);
}

namespace afe {
// for the other translation units of the library (afe_render.hip)
void engine_stream_device(afe_engine *e, void **stream, int *device) {
  *stream = (void *)main_stream(e);
  *device = e->device;
}
void engine_shard(const afe_engine *e, int64_t *first_global, int64_t *n) {
  *first_global = e->first_global;
  *n = e->n;
}
// An asynchronous neighbour query may still be reading the gathered buffer and the world scratch: whatever is about to
// rewrite them on the main stream is ordered behind it (a device-side wait; in steady state the query issued a cycle
// ago is long done and this costs nothing).
void engine_query_join(afe_engine *e) {
  if (!e->query_pending) return;
  (void)hipStreamWaitEvent(main_stream(e), e->ev_q_done, 0);
  e->query_pending = false;
}
// this shard's positions as planar fp32 [3][n] in the engine's own scratch, on its stream
int engine_pack_to_scratch(afe_engine *e, float **scratch) {
  engine_query_join(e);
  AFE_HIP(e, hipSetDevice(e->device));
  if (!e->pack_scratch) AFE_HIP(e, hipMalloc((void **)&e->pack_scratch, (size_t)e->n * 3 * sizeof(float)));
  const int rc = afe_pack_positions(e, e->pack_scratch);
  if (rc) return rc;
  *scratch = e->pack_scratch;
  return AFE_OK;
}
}  // namespace afe

static int device_view(afe_engine *e, afe_device_view *out, bool exported);
extern "C" int afe_get_device_view(afe_engine *e, afe_device_view *out) { return device_view(e, out, true); }
namespace afe {
// the library's own consumers (depth camera): stream-ordered behind the steps, nothing leaves the engine
int engine_device_view(afe_engine *e, afe_device_view *out) { return device_view(e, out, false); }
}
static int device_view(afe_engine *e, afe_device_view *out, bool exported) {
  if (!e || !out) return AFE_ERR_INVALID_ARG;
  const size_t have = out->struct_bytes;
  if (have < offsetof(afe_device_view, pos_anchor_xy))
    return fail(e, AFE_ERR_INVALID_ARG, "afe_device_view::struct_bytes must be set to sizeof(afe_device_view) before the call (ABI version 2)");
  AFE_HIP(e, hipSetDevice(e->device));
  (void)main_stream(e);            // a resident grid ends here (its last stores are the caller's to read); split streams are joined
  if (e->p_failed) return fail(e, AFE_ERR_HIP, e->err);
  if (exported) e->view_exported = true;   // from now on afe_sync leaves the slabs readable (it ends a resident grid)
  { const int mrc = materialize_motor(e); if (mrc) return mrc; }   // motor_speed is current as of this call
  afe_device_view v;
  v.struct_bytes = have < sizeof(v) ? have : sizeof(v);
  v.n_vehicles = e->n;
  v.stride = e->stride;
  v.state_elem_size = (int)elem(e);
  v.pos = e->pos; v.vel = e->vel; v.att = e->att; v.ang_vel = e->ang_vel;
  v.motor_speed = e->motor;
  v.ext_force = e->ext_force; v.ext_torque = e->ext_torque;
  v.motor_cmd = e->cmd; v.gyro = e->gyro; v.acc = e->acc;
  v.rng = e->rng; v.type_index = e->type;
  v.pos_anchor_xy = e->anchor;
  std::memcpy(out, &v, v.struct_bytes);      // never past what the caller has
  return AFE_OK;
}

extern "C" int afe_algorithmic_bytes_per_step(const afe_engine *e, int imu_tick, double *bytes) {
  if (!e || !bytes) return AFE_ERR_INVALID_ARG;
  const double es = (double)elem(e);
  double b = 17 * es * 2;      // state read + write (pos3 vel3 att4 angvel3 motor4)
  bool stateless = !e->table.empty();
  for (const HostParams &h : e->table) if (h.tau_m != 0 || h.Jm != 0) stateless = false;
  if (stateless) b -= 4 * es;  // tau_m == 0, J_m == 0: rotor speeds are never read ...
  if (motor_lazy(e)) b -= 4 * es;  // ... and with held commands not written either (rebuilt from the commands on demand)
  b += 4 * 4;                  // motor commands (float)
  if (!e->types_uniform) b += 1;  // type index (heterogeneous ensembles only)
  if (e->has_ext_force) b += 3 * es;
  if (e->has_ext_torque) b += 3 * es;
  if (imu_tick) {
    b += 6 * 4;                // gyro + accelerometer sample
    if (e->noise && e->seed_policy != AFE_SEED_COUNTER) b += 8;   // RNG word read + write (the counter policy keeps no word)
    if (e->logic_on) b += 12 * 4 * 2 + 4 * 4 + 4 * 4 + 3;  // LPF state r/w, rates cmd, motor cmd write, flags
  }
  *bytes = b;
  return AFE_OK;
}

extern "C" int afe_stream_probe(int device, int64_t n, int n_read, int n_write, int launches, float *us_per_launch) {
  if (n <= 0 || launches < 1 || !us_per_launch || !((n_read == 20 && n_write == 13) || (n_read == 24 && n_write == 17))) return AFE_ERR_INVALID_ARG;
  if (device >= 0 && hipSetDevice(device) != hipSuccess) return AFE_ERR_NO_DEVICE;
  int64_t stride = (n + 255) / 256 * 256;      // the engine's slab stride: 256 x an odd count
  if ((stride / 256) % 2 == 0) stride += 256;
  if ((uint64_t)stride * 4 * (uint64_t)n_read >= 0xffff0000ull) return AFE_ERR_INVALID_ARG;
  float *buf = nullptr;
  hipStream_t st = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = AFE_ERR_HIP;
  float best = 1e30f;
  if (hipMalloc((void **)&buf, (size_t)stride * 4 * n_read) != hipSuccess) return AFE_ERR_HIP;
  if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) goto out;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) goto out;
  if (hipMemsetAsync(buf, 0, (size_t)stride * 4 * n_read, st) != hipSuccess) goto out;
  for (int rep = 0; rep < 4; rep++) {          // the first repetition warms up
    if (hipEventRecord(e0, st) != hipSuccess) goto out;
    for (int k = 0; k < launches; k++)
      if (launch_stream_probe(buf, stride, n, n_read, n_write, st) != 0) goto out;
    if (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) goto out;
    float ms = 0;
    if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) goto out;
    if (rep > 0 && ms < best) best = ms;
  }
  *us_per_launch = best * 1e3f / (float)launches;
  rc = AFE_OK;
out:
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (st) (void)hipStreamDestroy(st);
  (void)hipFree(buf);
  return rc;
}

extern "C" int afe_event_create(void **event) {
  if (!event) return AFE_ERR_INVALID_ARG;
  hipEvent_t ev;
  if (hipEventCreate(&ev) != hipSuccess) return AFE_ERR_HIP;
  *event = (void *)ev;
  return AFE_OK;
}
extern "C" int afe_event_destroy(void *event) {
  if (!event) return AFE_ERR_INVALID_ARG;
  return hipEventDestroy((hipEvent_t)event) == hipSuccess ? AFE_OK : AFE_ERR_HIP;
}
extern "C" int afe_event_record(afe_engine *e, void *event) {
  if (!e || !event) return AFE_ERR_INVALID_ARG;
  AFE_HIP(e, hipSetDevice(e->device));
  if (e->p_running && !e->p_on_aql && !e->split_dirty && p_status(e)[0] == 0) {     // (a grid that has already left by itself takes the ordinary way)
    // A resident grid on the engine's stream: the event goes onto the stream BEHIND it and the grid is told to leave after
    // the last authorised step -- the event's time is then the moment the grid left the device, not the moment the host had
    // noticed (which is what recording after the park gave: ~10 us later, 0.5 us per step of a 20-step block).
    __atomic_store_n(&e->p_host[e->p_next & (AFE_PERSIST_HOST_RING - 1)], ((e->p_next + 1) << 2) | AFE_PERSIST_PARK, __ATOMIC_RELEASE);
    AFE_HIP(e, hipEventRecord((hipEvent_t)event, e->stream));
    e->stream_pending = true;
    return persist_park(e);
  }
  AFE_HIP(e, hipEventRecord((hipEvent_t)event, main_stream(e)));
  return AFE_OK;
}
extern "C" int afe_event_elapsed_ms(void *start, void *stop, float *ms) {
  if (!start || !stop || !ms) return AFE_ERR_INVALID_ARG;
  if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess) return AFE_ERR_HIP;
  return hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop) == hipSuccess ? AFE_OK : AFE_ERR_HIP;
}

namespace {
// Everything a step depends on that is not in the arenas travels in (or is verified through) the
// header: the clock, which optional inputs are live, the noise / seeding / logic-period
// configuration (restored on load), and fingerprints of the expanded type and logic tables
// (verified on load: a checkpoint cannot carry the tables' host-side source records, so the
// receiving engine must have been given the same ones).
struct CheckpointHeader {
  uint64_t magic, n, stride, precision, arena_bytes, logic_bytes;
  uint64_t now_us, logic_elapsed_us, n_ticks;
  uint64_t has_ext_force, has_ext_torque, logic_on;
  uint64_t n_types, table_hash, logic_hash;
  uint64_t noise, seed_policy;
  double sigma_gyro, sigma_acc, logic_period;
  uint64_t noise_seed, gust_on, gust_seed, gust_period_us, gust_n_global, gust_applied;
  double gust_sigma_max;
};
const uint64_t kCheckpointMagic = 0x4146452d434b5033ull;  // "AFE-CKP3"
size_t logic_arena_bytes(const afe_engine *e) {
  const size_t S = (size_t)e->stride;
  return e->logic_arena ? S * 12 * 4 + S * 4 * 4 + S * 2 + 256 * sizeof(DevLogic) : 0;
}
inline void fnv(uint64_t &h, const void *p, size_t n) {
  const unsigned char *b = (const unsigned char *)p;
  for (size_t k = 0; k < n; k++) { h ^= b[k]; h *= 1099511628211ull; }
}
// field by field: struct padding is not part of the fingerprint
uint64_t table_fingerprint(const afe_engine *e) {
  uint64_t h = 1469598103934665603ull;
  for (const HostParams &t : e->table) {
    fnv(h, &t.mass, sizeof(t.mass)); fnv(h, t.I, sizeof(t.I)); fnv(h, t.Iinv, sizeof(t.Iinv));
    fnv(h, t.mp, sizeof(t.mp)); fnv(h, &t.kf, sizeof(t.kf)); fnv(h, &t.ktau, sizeof(t.ktau));
    fnv(h, &t.tau_m, sizeof(t.tau_m)); fnv(h, &t.Jm, sizeof(t.Jm)); fnv(h, &t.wmin, sizeof(t.wmin));
    fnv(h, &t.wmax, sizeof(t.wmax)); fnv(h, t.drag, sizeof(t.drag)); fnv(h, t.Rimu, sizeof(t.Rimu));
  }
  return h;
}
uint64_t logic_fingerprint(const afe_engine *e) {
  uint64_t h = 1469598103934665603ull;
  if (!e->logic_on) return 0;
  for (const afe_rates_logic_params &t : e->logic_params) {
    fnv(h, &t.mass, sizeof(t.mass)); fnv(h, t.inertia, sizeof(t.inertia));
    fnv(h, &t.ang_vel_time_const_xy, 4); fnv(h, &t.ang_vel_time_const_z, 4); fnv(h, &t.arm_length, 4);
    fnv(h, &t.prop_thrust_from_speed_sqr, 4); fnv(h, &t.prop_torque_from_thrust, 4); fnv(h, &t.prop0_spin_dir, 4);
    fnv(h, &t.max_thrust_per_propeller, 4); fnv(h, &t.min_thrust_per_propeller, 4); fnv(h, &t.max_cmd_total_thrust, 4);
    fnv(h, &t.imu_yaw, 4); fnv(h, &t.imu_pitch, 4); fnv(h, &t.imu_roll, 4); fnv(h, &t.gyro_lowpass_cutoff, 4);
  }
  return h;
}
}  // namespace

extern "C" int afe_checkpoint_size(const afe_engine *e, uint64_t *bytes) {
  if (!e || !bytes) return AFE_ERR_INVALID_ARG;
  *bytes = sizeof(CheckpointHeader) + e->arena_bytes + (e->logic_on ? logic_arena_bytes(e) : 0);
  return AFE_OK;
}

extern "C" int afe_save_checkpoint(afe_engine *e, void *host_buffer, uint64_t bytes) {
  uint64_t need = 0;
  if (!e || !host_buffer || afe_checkpoint_size(e, &need) != AFE_OK) return AFE_ERR_INVALID_ARG;
  if (bytes < need) return fail(e, AFE_ERR_INVALID_ARG, "checkpoint buffer too small");
  AFE_HIP(e, hipSetDevice(e->device));
  { const int mrc = materialize_motor(e); if (mrc) return mrc; }
  AFE_HIP(e, hipStreamSynchronize(main_stream(e)));
  CheckpointHeader h = {};
  h.magic = kCheckpointMagic; h.n = (uint64_t)e->n; h.stride = (uint64_t)e->stride; h.precision = (uint64_t)e->precision;
  h.arena_bytes = (uint64_t)e->arena_bytes; h.logic_bytes = e->logic_on ? (uint64_t)logic_arena_bytes(e) : 0;
  h.now_us = e->now_us; h.logic_elapsed_us = e->logic_elapsed_us; h.n_ticks = e->n_ticks;
  h.has_ext_force = e->has_ext_force; h.has_ext_torque = e->has_ext_torque; h.logic_on = e->logic_on;
  h.n_types = (uint64_t)e->table.size(); h.table_hash = table_fingerprint(e); h.logic_hash = logic_fingerprint(e);
  h.noise = e->noise; h.seed_policy = (uint64_t)e->seed_policy;
  h.sigma_gyro = e->sigma_gyro; h.sigma_acc = e->sigma_acc; h.logic_period = e->logic_period;
  h.noise_seed = e->noise_seed; h.gust_on = e->gust_on; h.gust_seed = e->gust_seed; h.gust_period_us = e->gust_period_us;
  h.gust_n_global = e->gust_n_global; h.gust_applied = e->gust_applied; h.gust_sigma_max = e->gust_sigma_max;
  char *p = (char *)host_buffer;
  std::memcpy(p, &h, sizeof(h));
  p += sizeof(h);
  if (e->host_arena) std::memcpy(p, e->arena_host, e->arena_bytes);
  else AFE_HIP(e, hipMemcpy(p, e->arena, e->arena_bytes, hipMemcpyDeviceToHost));
  p += e->arena_bytes;
  if (h.logic_bytes && e->host_arena) std::memcpy(p, e->logic_arena_host, h.logic_bytes);
  else if (h.logic_bytes) AFE_HIP(e, hipMemcpy(p, e->logic_arena, h.logic_bytes, hipMemcpyDeviceToHost));
  return AFE_OK;
}

extern "C" int afe_load_checkpoint(afe_engine *e, const void *host_buffer, uint64_t bytes) {
  if (!e || !host_buffer || bytes < sizeof(CheckpointHeader)) return AFE_ERR_INVALID_ARG;
  CheckpointHeader h;
  std::memcpy(&h, host_buffer, sizeof(h));
  if (h.magic != kCheckpointMagic) return fail(e, AFE_ERR_INVALID_ARG, "not an engine checkpoint (or one of another format version)");
  if (h.n != (uint64_t)e->n || h.stride != (uint64_t)e->stride || h.precision != (uint64_t)e->precision ||
      h.arena_bytes != e->arena_bytes)
    return fail(e, AFE_ERR_INVALID_ARG, "checkpoint does not match this engine's size / precision");
  if ((h.logic_on != 0) != e->logic_on || h.logic_bytes != (e->logic_on ? (uint64_t)logic_arena_bytes(e) : 0))
    return fail(e, AFE_ERR_INVALID_ARG, "checkpoint and engine disagree about the on-device logic (afe_set_rates_logic)");
  if (h.logic_bytes > bytes || h.arena_bytes > bytes || bytes < sizeof(h) + h.arena_bytes + h.logic_bytes)
    return fail(e, AFE_ERR_INVALID_ARG, "checkpoint buffer is truncated");
  if (h.n_types != (uint64_t)e->table.size() || h.table_hash != table_fingerprint(e))
    return fail(e, AFE_ERR_INVALID_ARG, "checkpoint was taken with a different vehicle type table (afe_set_type_table first)");
  if (h.logic_hash != logic_fingerprint(e))
    return fail(e, AFE_ERR_INVALID_ARG, "checkpoint was taken with different on-device logic parameters");
  if ((h.seed_policy != AFE_SEED_REFERENCE && h.seed_policy != AFE_SEED_DECORRELATED && h.seed_policy != AFE_SEED_COUNTER) || !(h.logic_period > 0) ||
      !(h.sigma_gyro >= 0) || !(h.sigma_acc >= 0) || (h.gust_on && (h.gust_period_us == 0 || !(h.gust_sigma_max >= 0) || h.gust_n_global < (uint64_t)(e->first_global + e->n))))
    return fail(e, AFE_ERR_INVALID_ARG, "checkpoint header holds an invalid configuration");
  // Everything the kernels will index or loop on is checked in the HOST buffer, before a byte reaches the device or
  // the engine's own state changes: a refused checkpoint leaves the engine exactly as it was.
  std::vector<uint8_t> types((size_t)e->n);
  {
    const char *arena_in_ckp = (const char *)host_buffer + sizeof(h);
    std::memcpy(types.data(), arena_in_ckp + ((const char *)e->type - (const char *)e->arena), (size_t)e->n);
    for (size_t k = 0; k < types.size(); k++)
      if (types[k] >= e->table.size()) return fail(e, AFE_ERR_INVALID_ARG, "checkpoint holds a type index outside the type table");
    // minstd_rand0 words live in [1, 2^31 - 2]: 0 is a fixed point of the generator and anything else is not a state of
    // it -- the acceptance loop of the Gaussian draws would never end on such a word
    const uint32_t *words = (const uint32_t *)(arena_in_ckp + ((const char *)e->rng - (const char *)e->arena));
    for (int64_t k = 0; k < e->n; k++) {
      uint32_t w;
      std::memcpy(&w, words + k, 4);
      if (w == 0 || w >= 2147483647u)
        return fail(e, AFE_ERR_INVALID_ARG, "checkpoint holds an engine word outside [1, 2^31-2] (vehicle " + std::to_string(k) + ")");
    }
  }
  AFE_HIP(e, hipSetDevice(e->device));
  AFE_HIP(e, hipStreamSynchronize(main_stream(e)));
  // from here on the arena is being overwritten: whatever happens, nothing derived from the old one may survive
  e->table_dirty = true;
  e->logic_table_period = -1.0f;
  const char *p = (const char *)host_buffer + sizeof(h);
  if (e->host_arena) std::memcpy(e->arena_host, p, e->arena_bytes);
  else AFE_HIP(e, hipMemcpy(e->arena, p, e->arena_bytes, hipMemcpyHostToDevice));
  p += e->arena_bytes;
  if (h.logic_bytes && e->host_arena) std::memcpy(e->logic_arena_host, p, h.logic_bytes);
  else if (h.logic_bytes) AFE_HIP(e, hipMemcpy(e->logic_arena, p, h.logic_bytes, hipMemcpyHostToDevice));
  e->now_us = h.now_us;
  e->logic_elapsed_us = h.logic_elapsed_us;
  e->n_ticks = h.n_ticks;
  e->has_ext_force = h.has_ext_force != 0;
  e->has_ext_torque = h.has_ext_torque != 0;
  e->noise = h.noise != 0;
  e->seed_policy = (int)h.seed_policy;   // the RNG words themselves came with the arena
  e->sigma_gyro = h.sigma_gyro;
  e->sigma_acc = h.sigma_acc;
  e->logic_period = h.logic_period;
  e->noise_seed = h.noise_seed;
  e->gust_on = h.gust_on != 0; e->gust_seed = h.gust_seed; e->gust_period_us = h.gust_period_us; e->gust_n_global = h.gust_n_global;
  e->gust_applied = h.gust_applied; e->gust_sigma_max = h.gust_sigma_max;
  // the kernel-argument fast path is only valid when every vehicle uses record 0: decide from the
  // restored per-vehicle type slab, not from what this engine was told before the load
  e->type_host.swap(types);
  e->run_nonzero.clear();          // re-derive everything from the restored slab
  refresh_type_flags(e);
  // the arena also holds the device copy of the type table for the checkpoint's dt: rebuild on next step
  e->table_dirty = true;
  e->logic_table_period = -1.0f;
  e->motor_stale = false;   // the checkpoint's rotor-speed slab was current when it was taken
  return AFE_OK;
}

namespace {
int selftest_normals_any(afe_engine *e, const uint32_t *seeds, int64_t n, void *normals6, size_t elem_bytes, uint32_t *state_after) {
  if (!e || !seeds || n <= 0 || !normals6 || !state_after) return fail(e, AFE_ERR_INVALID_ARG, "bad selftest arguments");
  for (int64_t k = 0; k < n; k++)
    if (seeds[k] == 0 || seeds[k] >= 2147483647u) return fail(e, AFE_ERR_INVALID_ARG, "minstd_rand0 state must be in [1, 2^31-2]");
  AFE_HIP(e, hipSetDevice(e->device));
  uint32_t *d_seed = nullptr, *d_state = nullptr;
  void *d_out = nullptr;
  AFE_HIP(e, hipMalloc((void **)&d_seed, (size_t)n * 4));
  AFE_HIP(e, hipMalloc((void **)&d_state, (size_t)n * 4));
  AFE_HIP(e, hipMalloc((void **)&d_out, (size_t)n * 6 * elem_bytes));
  int rc = AFE_OK;
  if (hipMemcpy(d_seed, seeds, (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess ||
      (elem_bytes == 8 ? launch_normals_selftest(d_seed, n, (double *)d_out, d_state, main_stream(e))
                       : launch_normals_selftest_f32(d_seed, n, (float *)d_out, d_state, main_stream(e))) != 0 ||
      hipStreamSynchronize(main_stream(e)) != hipSuccess ||
      hipMemcpy(normals6, d_out, (size_t)n * 6 * elem_bytes, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(state_after, d_state, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess)
    rc = fail(e, AFE_ERR_HIP, "normals selftest failed");
  (void)hipFree(d_seed); (void)hipFree(d_state); (void)hipFree(d_out);
  return rc;
}
}  // namespace

extern "C" int afe_selftest_normals(afe_engine *e, const uint32_t *seeds, int64_t n, double *normals6, uint32_t *state_after) {
  return selftest_normals_any(e, seeds, n, normals6, 8, state_after);
}
extern "C" int afe_selftest_normals_f32(afe_engine *e, const uint32_t *seeds, int64_t n, float *normals6, uint32_t *state_after) {
  return selftest_normals_any(e, seeds, n, normals6, 4, state_after);
}

extern "C" int afe_pack_positions(afe_engine *e, float *device_xyz) {
  if (!e || !device_xyz) return fail(e, AFE_ERR_INVALID_ARG, "device_xyz is NULL");
  AFE_HIP(e, hipSetDevice(e->device));
  engine_query_join(e);
  int rc = (e->precision == AFE_F64)
               ? launch_pack_positions_f64((const double *)e->pos, e->anchor, e->stride, e->n, device_xyz, main_stream(e))
               : launch_pack_positions_f32((const float *)e->pos, e->anchor, e->stride, e->n, device_xyz, main_stream(e));
  if (rc) return fail(e, AFE_ERR_HIP, "pack kernel launch failed");
  return AFE_OK;
}

extern "C" int afe_nearest_neighbour(afe_engine *e, const float *all_xyz, int64_t n_all, float *dist2_out,
                                     int32_t *index_out) {
  return afe_nearest_neighbour_grid(e, all_xyz, n_all, 0.0f, dist2_out, index_out);
}

extern "C" int afe_nearest_neighbour_grid(afe_engine *e, const float *all_xyz, int64_t n_all, float cell_size,
                                          float *dist2_out, int32_t *index_out) {
  if (!e || !all_xyz || n_all <= 0 || !dist2_out || !index_out)
    return fail(e, AFE_ERR_INVALID_ARG, "bad nearest-neighbour arguments");
  if (e->first_global + e->n > n_all)
    return fail(e, AFE_ERR_OUT_OF_RANGE, "the gathered ensemble is smaller than this shard's global range");
  AFE_HIP(e, hipSetDevice(e->device));
  if (!e->world) { const int rc = world_create(e->device, &e->world); if (rc) return fail(e, rc, "shared-world scratch"); }
  engine_query_join(e);
  const int rc = world_nearest(e->world, (void *)main_stream(e), all_xyz, n_all, e->first_global, e->n, cell_size, dist2_out, index_out);
  if (rc) return fail(e, rc, world_last_error(e->world));
  return AFE_OK;
}

// The same query on a stream of its own: ordered behind everything the engine's stream holds now (the gather that filled
// all_xyz), it runs while the engine goes on stepping -- afe_step is not ordered behind it; what rewrites the buffer or
// the query's scratch is (engine_query_join).  Results are complete after afe_query_sync.
extern "C" int afe_nearest_neighbour_async(afe_engine *e, const float *all_xyz, int64_t n_all, float *dist2_out, int32_t *index_out) {
  if (!e || !all_xyz || n_all <= 0 || !dist2_out || !index_out) return fail(e, AFE_ERR_INVALID_ARG, "bad nearest-neighbour arguments");
  if (e->first_global + e->n > n_all) return fail(e, AFE_ERR_OUT_OF_RANGE, "the gathered ensemble is smaller than this shard's global range");
  AFE_HIP(e, hipSetDevice(e->device));
  if (!e->world) { const int rc = world_create(e->device, &e->world); if (rc) return fail(e, rc, "shared-world scratch"); }
  if (!e->query_stream) {
    AFE_HIP(e, hipStreamCreateWithFlags(&e->query_stream, hipStreamNonBlocking));
    AFE_HIP(e, hipEventCreateWithFlags(&e->ev_q_start, hipEventDisableTiming));
    AFE_HIP(e, hipEventCreateWithFlags(&e->ev_q_done, hipEventDisableTiming));
  }
  e->query_pending = false;      // a query already on the query stream is ordered before this one by the stream itself
  AFE_HIP(e, hipEventRecord(e->ev_q_start, main_stream(e)));
  AFE_HIP(e, hipStreamWaitEvent(e->query_stream, e->ev_q_start, 0));
  const int rc = world_nearest(e->world, (void *)e->query_stream, all_xyz, n_all, e->first_global, e->n, 0.0f, dist2_out, index_out);
  if (rc) return fail(e, rc, world_last_error(e->world));
  AFE_HIP(e, hipEventRecord(e->ev_q_done, e->query_stream));
  e->query_pending = true;
  return AFE_OK;
}

extern "C" int afe_query_sync(afe_engine *e) {
  if (!e) return AFE_ERR_INVALID_ARG;
  if (!e->query_stream) return AFE_OK;
  AFE_HIP(e, hipSetDevice(e->device));
  AFE_HIP(e, hipStreamSynchronize(e->query_stream));
  return AFE_OK;
}

extern "C" int afe_set_neighbour_grid_refresh(afe_engine *e, int every_n_queries) {
  if (!e || every_n_queries < 1) return fail(e, AFE_ERR_INVALID_ARG, "refresh interval must be >= 1");
  AFE_HIP(e, hipSetDevice(e->device));
  if (!e->world) { const int rc = world_create(e->device, &e->world); if (rc) return fail(e, rc, "shared-world scratch"); }
  return world_set_refresh(e->world, every_n_queries);
}

extern "C" int afe_set_neighbour_sort_reuse(afe_engine *e, int every_n_queries) {
  if (!e || every_n_queries < 1) return fail(e, AFE_ERR_INVALID_ARG, "sort interval must be >= 1");
  AFE_HIP(e, hipSetDevice(e->device));
  if (!e->world) { const int rc = world_create(e->device, &e->world); if (rc) return fail(e, rc, "shared-world scratch"); }
  return world_set_sort_reuse(e->world, every_n_queries);
}

extern "C" int afe_neighbour_grid_info(const afe_engine *e, int dims[3], float *cell_size, int64_t *n_cells, int64_t *n_bruteforce) {
  if (!e || !e->world) return AFE_ERR_NOT_CONFIGURED;
  return world_grid_info(e->world, dims, cell_size, n_cells, n_bruteforce);
}

extern "C" int afe_nearest_neighbour_bruteforce(afe_engine *e, const float *all_xyz, int64_t n_all, const int32_t *dev_queries,
                                                int64_t n_queries, float *dist2_out, int32_t *index_out) {
  if (!e || !all_xyz || n_all <= 0 || !dev_queries || n_queries <= 0 || !dist2_out || !index_out)
    return fail(e, AFE_ERR_INVALID_ARG, "bad nearest-neighbour arguments");
  AFE_HIP(e, hipSetDevice(e->device));
  if (!e->world) { const int rc = world_create(e->device, &e->world); if (rc) return fail(e, rc, "shared-world scratch"); }
  engine_query_join(e);
  const int rc = world_nearest_bruteforce(e->world, (void *)main_stream(e), all_xyz, n_all, e->first_global, dev_queries, n_queries,
                                          dist2_out, index_out);
  if (rc) return fail(e, rc, world_last_error(e->world));
  return AFE_OK;
}
