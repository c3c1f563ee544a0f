// afe_device.h -- structures shared by the host engine and the HIP kernels.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace afe {

// Per-type constant record as the kernel wants it: the Quadcopter_T ctor
// arguments (reference Components/Components/Simulation/Quadcopter_T.hpp:24-32)
// expanded the way the ctor body does (Quadcopter_T.cpp:20,45-65,75-80), plus
// the motor-lag factor exp(-dt/tau_m) of Motor.cpp:54-58, which depends only on
// (type, dt) and is therefore evaluated once per dt on the host in double.
// Staged into LDS at kernel start; all lanes of a homogeneous ensemble read
// the same LDS words (broadcast, conflict-free).
template <typename R>
struct alignas(16) DevParams {
  R mass;
  R inv_mass;  // 1/mass (fp32 kernel multiplies; the fp64 kernel divides like the reference)
  R I[9];      // _inertiaMatrix, row major
  R Iinv[9];   // _inertiaMatrixInv
  R mpx[4];    // motor positions (FR, RR, RL, FL)
  R mpy[4];
  R mpz[4];
  R kf;        // _thrustFromSpeedSqr
  R ktau;      // _torqueFromSpeedSqr
  R c_lag;     // exp(-dt/_timeConstant), or 0 when _timeConstant == 0
  R omc_lag;   // 1 - c_lag evaluated in double (-expm1(-dt/tau)): the fp32 kernel's rotor-speed increment factor
  R Jm;        // Motor::_inertia
  R wmin;
  R wmax;
  R drag[3];   // _linDragCoeffB
  float Rimu[9];  // _R_inverse (float in the reference too)
  // kf / mass in the host's double, for the fp32 kernel: the vertical chain  sum k_f w|w| -> / m -> - 9.81  is two 9.81 m/s^2
  // terms that cancel on a hovering vehicle; that one product chain is carried in double registers (run_vehicle), which needs
  // its constant unrounded.  (A scalar-register pair of the kernel-argument segment: an fp64 operand as it is.)
  double kf_over_mass_d;
};
static_assert(sizeof(DevParams<float>) % 16 == 0, "DevParams<float> size");
static_assert(sizeof(DevParams<double>) % 16 == 0, "DevParams<double> size");

// On-device onboard rates logic (optional): constants of the rates-control slice
// of Onboard::QuadcopterLogic, all float like the reference's onboard code.
struct alignas(16) DevLogic {
  float mass;
  float I[9];        // QuadcopterConstants::inertiaMatrix
  float tc_xy, tc_z; // angular-velocity controller time constants
  float d, kt, kf;   // QuadcopterMixer::_d, _kt, _kf
  float max_thrust, min_thrust, max_cmd_total;
  float R[9];        // QuadcopterLogic::_R (IMU mount, NOT inverted)
  float a1, a2, b0, b1, b2;  // LowPassFilterSecondOrder coefficients (gyro)
};

// Motor i spins about s_i * z with s = (+1,-1,+1,-1) and pushes along +z
// (Quadcopter_T.cpp:45-65 with Motor.cpp:32-36).
#define AFE_MOTOR_SPIN(i) (((i) & 1) ? -1 : +1)

template <typename R>
struct StepView {
  R *pos, *vel, *att, *ang_vel, *motor;  // planar SoA, `stride` between comps
  const R *ext_force, *ext_torque;
  const float *cmd;
  float *gyro, *acc;
  uint32_t *rng;
  const uint8_t *type;
  const DevParams<R> *table;
  int n_types;
  int64_t n;
  int64_t first, end;   // this launch steps vehicles [first, end) of the n (afe_set_split_stepping: the two halves on two streams)
  int64_t stride;
  R dt;
  R inv_dt;  // 1/dt evaluated in double on the host
  int n_steps;
  unsigned long long tick_mask;  // bit s set: sub-step s fires the logic gate
  float sigma_gyro, sigma_acc;
  // every type has tau_m == 0 and J_m == 0 (all shipped vehicle types): the rotor
  // speed is then a pure function of the command, Motor.cpp:54-66, and is not
  // read back (it is still written, for GetMotorForce / afe_get_state)
  int motor_stateless;
  int motor_write;   // 0: the rotor-speed slab is not written (it is clamp(cmd), rebuilt on demand)
  // on-device rates logic (null when disabled)
  float *lpf;               // 12 comps: xm0[3] xm1[3] ym0[3] ym1[3]
  const float *rates_cmd;   // 4 comps: thrust_norm, wx, wy, wz
  const uint8_t *have_cmd;  // 1: EXTERNAL_RATES_CONTROL, 0: IDLE
  uint8_t *imu_init;        // KalmanFilter6DOF::_IMUInitialized
  float *cmd_out;           // == cmd (written by the logic)
  const DevLogic *logic_table;
  // Buffer addressing (afe_kernels.hip run_vehicle): bytes of the engine's arena reachable from `pos`, its first
  // slab -- every slab up to `type` lies inside -- and of the logic arena from `lpf`; 0 when an arena does not
  // fit 32-bit offsets (> 4 GiB: ensembles beyond ~31 M fp32 vehicles), which selects the global-address kernels.
  uint32_t buf_bytes;
  uint32_t logic_buf_bytes;
  // AFE_SEED_COUNTER (counter-based IMU noise): sample = f(noise_seed, first_global + i, logic-tick number)
  uint64_t noise_seed;
  uint64_t tick_base;       // logic ticks before this launch (persistent grid: before its first step)
  int64_t first_global;     // this shard's offset in the whole ensemble
};

struct LaunchFlags {
  bool ext_force, ext_torque, noise, logic;
  bool resident = false;        // persistent grid only: AFE_STEP_RESIDENT (fused batches, loads shared)
  bool counter_noise = false;   // noise from the counter-based generator (AFE_SEED_COUNTER) instead of the per-vehicle libstdc++ stream
  // heterogeneous ensemble whose type index is constant over every aligned run of 64 vehicles (fleets
  // laid out type by type): each wave then reads its one record by scalar loads -- no LDS table
  bool wave_uniform_types = false;
  // cache-policy bits of the one-step launches' slab accesses (afe_kernels.hip run_vehicle, CP): 0 default, 1 inputs and
  // outputs nt, 2 everything nt, 3 everything nt + one contiguous range per XCD.  Hints only: never a different bit.
  int cache_policy = 0;
};

// ---------------------------------------------------------------------------
// Persistent stepping (afe_set_step_mode(AFE_STEP_PERSISTENT)): ONE resident grid steps the ensemble for as long
// as the host keeps authorising steps.  A step is one 64-bit ring entry the host writes into pinned memory:
//   entry = (absolute step index + 1) << 2 | park << 1 | tick        (index in bits 2-47)
// (a slot whose index part does not match is not written yet).  Workgroup 0 -- the pump -- copies new entries
// from the host ring into a device ring; every other workgroup (one wave each) polls the device ring and steps
// ITS vehicles through every entry it finds: loads, the step, stores -- the body of the one-step launch, state
// back in HBM after every step.  Vehicles do not interact, so no wave ever waits for another one; what a kernel
// boundary costs between two dependent launches (drain, dispatch, ramp-up) is simply not there.  A park entry
// ends the launch; the pump writes one itself when the host has not authorised anything for `idle_ticks`
// (nothing ever waits on a host that went away), and every spin gives up after `give_up_ticks`.
struct PersistArgs {
  const unsigned long long *host_ring;   // pinned host memory, as the device sees it
  unsigned long long *host_status;       // pinned: [0] park position + 1 (0 while running), [1] steps every worker has consumed, [2] error code
  unsigned long long *dev_ring;          // device copy of the ring (workers poll this one)
  unsigned long long *done;              // per worker wave: steps consumed
  unsigned long long start;              // first step of this launch (absolute index)
  unsigned int host_mask, dev_mask;      // ring sizes - 1 (powers of two)
  int n_workers;                         // worker waves (= workgroups - 1)
  int n_chunks;                          // aligned runs of 64 vehicles; worker w steps chunks w, w + n_workers, ...
  unsigned int idle_ticks;               // 100 MHz ticks (s_memrealtime)
  unsigned int give_up_ticks;
  unsigned int epoch;                    // bits 0-15: launch counter (stamps the device ring's entries); AFE_PERSIST_HOST_IO: the arena is host memory
  // gust process (afe_set_gust_process; gust_period_us == 0: off).  A step's start time is t0_us + (its index - start) * dt_us;
  // its epoch floor(time / period) is tracked incrementally from gust_epoch0 = epoch of t0_us.
  unsigned long long gust_period_us, gust_seed, gust_n_global, gust_epoch0, gust_epoch_applied, t0_us, dt_us;
  double gust_sigma_max;
};
// The kernel-argument segment of afe_step_persistent_kernel(StepView<R>, DevParams<R>, DevLogic, PersistArgs) as the
// host packs it for a dispatch on the engine's own queue (afe_engine.cpp aql_launch): the same four objects in the same
// order, each at its natural alignment -- the rule the compiler lays the segment out by.  The segment ends with its last
// argument (no tail padding), hence persist_kernarg_bytes.
template <typename R>
struct PersistKernarg {
  StepView<R> v;
  DevParams<R> P;
  DevLogic G;
  PersistArgs a;
};
template <typename R>
constexpr size_t persist_kernarg_bytes() { return offsetof(PersistKernarg<R>, a) + sizeof(PersistArgs); }

// "tell me when step S - 1 is done" without ending the grid (afe_sync on a grid that stays resident): the host writes S
// into host_status[AFE_PERSIST_SYNCREQ_WORD]; once everything before S is republished the pump puts a MARKER into slot S of
// the device ring (the index of an entry for step S, both flags set: neither a step nor a park); a worker that finds the
// marker under its own count says so ONCE -- one atomic on its shard's counter (64 shards, a line each, behind done[]),
// the shard's last arrival one more on the top counter, the last of those writes S to host_status[AFE_PERSIST_SYNC_WORD].
// Arrivals of two requests never interleave (the host waits for each): a counter is complete at every multiple of its size.
#define AFE_PERSIST_SYNC_SHARDS 64
#define AFE_PERSIST_SYNC_AREA_WORDS (16 * (AFE_PERSIST_SYNC_SHARDS + 3))   /* a pad line, 64 shard lines, the top line, slack */
#define AFE_PERSIST_HOST_IO 0x10000u   /* PersistArgs::epoch */
#define AFE_PERSIST_PRIO 0x20000u      /* PersistArgs::epoch: workers set their issue priority by the steps they have left (afe_kernels.hip) */
#define AFE_PERSIST_HOST_MARKS 64      /* host-visible arenas: grids of up to this many workers also write their marks to host_status[8 + w] */
#define AFE_PERSIST_SYNC_WORD (8 + AFE_PERSIST_HOST_MARKS)      /* host_status: the step count the last sync request was answered for */
#define AFE_PERSIST_SYNCREQ_WORD (9 + AFE_PERSIST_HOST_MARKS)   /* host_status: the host's sync request (a step count) */
#define AFE_PERSIST_STATUS_WORDS (16 + AFE_PERSIST_HOST_MARKS)
#define AFE_PERSIST_TICK 1ull
#define AFE_PERSIST_PARK 2ull
#define AFE_PERSIST_HOST_RING 4096
#define AFE_PERSIST_DEV_RING 1024

// kernel launchers (afe_kernels.hip); stream is a hipStream_t
// `uniform` != nullptr: every vehicle uses this one record, passed by value in
// the kernel-argument segment (scalar registers); otherwise v.table is staged
// into LDS and indexed per lane by v.type (or, LaunchFlags::wave_uniform_types, read per wave by scalar loads).
int launch_step_f32(const StepView<float> &v, const LaunchFlags &f, const DevParams<float> *uniform,
                    const DevLogic *uniform_logic, void *stream);
int launch_step_f64(const StepView<double> &v, const LaunchFlags &f, const DevParams<double> *uniform,
                    const DevLogic *uniform_logic, void *stream);
// the persistent grid: 1 + a.n_workers one-wave workgroups (homogeneous ensembles, no external torque, buffer addressing)
int launch_persistent_f32(const StepView<float> &v, const LaunchFlags &f, const DevParams<float> &uniform,
                          const DevLogic *uniform_logic, const PersistArgs &a, void *stream);
int launch_persistent_f64(const StepView<double> &v, const LaunchFlags &f, const DevParams<double> &uniform,
                          const DevLogic *uniform_logic, const PersistArgs &a, void *stream);
// the host address of that instantiation (for the engine's own AQL queue, afe_aql.h)
const void *persistent_kernel_fn_f32(const LaunchFlags &f);
const void *persistent_kernel_fn_f64(const LaunchFlags &f);
// one-wave workgroups of that instantiation a CU keeps resident (0: the query failed)
int persistent_capacity_f32(const LaunchFlags &f);
int persistent_capacity_f64(const LaunchFlags &f);
// rotor speeds of stateless motors from the commands: w = clamp(max(0, cmd), w_min, w_max)
int launch_motor_from_cmd_f32(float *motor, const float *cmd, const uint8_t *type, const DevParams<float> *table,
                              int64_t stride, int64_t n, void *stream);
int launch_motor_from_cmd_f64(double *motor, const float *cmd, const uint8_t *type, const DevParams<double> *table,
                              int64_t stride, int64_t n, void *stream);
int launch_pack_positions_f32(const float *pos, const double *anchor_xy, int64_t stride, int64_t n, float *out, void *stream);
int launch_pack_positions_f64(const double *pos, const double *anchor_xy, int64_t stride, int64_t n, float *out, void *stream);
int launch_normals_selftest(const uint32_t *seeds, int64_t n, double *out, uint32_t *state_out, void *stream);
int launch_normals_selftest_f32(const uint32_t *seeds, int64_t n, float *out, uint32_t *state_out, void *stream);
int launch_gust_f32(float *ext_force, int64_t stride, int64_t n, int64_t first_global, uint64_t n_global, uint64_t seed, uint64_t epoch,
                    double sigma_max, void *stream);
int launch_gust_f64(double *ext_force, int64_t stride, int64_t n, int64_t first_global, uint64_t n_global, uint64_t seed, uint64_t epoch,
                    double sigma_max, void *stream);
int launch_stream_probe(float *base, int64_t stride, int64_t n, int n_read, int n_write, void *stream);
int launch_seed_rng(uint32_t *rng, int64_t n, int64_t first_global, int policy, void *stream);

}  // namespace afe
