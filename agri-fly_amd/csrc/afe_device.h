// afe_device.h -- structures shared by the host engine and the HIP kernels.
#pragma once
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
};

struct LaunchFlags {
  bool ext_force, ext_torque, noise, logic;
  // heterogeneous ensemble whose type index is constant over every aligned run of 64 vehicles (fleets
  // laid out type by type): each wave then reads its one record by scalar loads -- no LDS table
  bool wave_uniform_types = false;
};

// kernel launchers (afe_kernels.hip); stream is a hipStream_t
// `uniform` != nullptr: every vehicle uses this one record, passed by value in
// the kernel-argument segment (scalar registers); otherwise v.table is staged
// into LDS and indexed per lane by v.type (or, LaunchFlags::wave_uniform_types, read per wave by scalar loads).
int launch_step_f32(const StepView<float> &v, const LaunchFlags &f, const DevParams<float> *uniform,
                    const DevLogic *uniform_logic, void *stream);
int launch_step_f64(const StepView<double> &v, const LaunchFlags &f, const DevParams<double> *uniform,
                    const DevLogic *uniform_logic, void *stream);
// rotor speeds of stateless motors from the commands: w = clamp(max(0, cmd), w_min, w_max)
int launch_motor_from_cmd_f32(float *motor, const float *cmd, const uint8_t *type, const DevParams<float> *table,
                              int64_t stride, int64_t n, void *stream);
int launch_motor_from_cmd_f64(double *motor, const float *cmd, const uint8_t *type, const DevParams<double> *table,
                              int64_t stride, int64_t n, void *stream);
int launch_pack_positions_f32(const float *pos, int64_t stride, int64_t n, float *out, void *stream);
int launch_pack_positions_f64(const double *pos, int64_t stride, int64_t n, float *out, void *stream);
int launch_normals_selftest(const uint32_t *seeds, int64_t n, double *out, uint32_t *state_out, void *stream);
int launch_normals_selftest_f32(const uint32_t *seeds, int64_t n, float *out, uint32_t *state_out, void *stream);
int launch_seed_rng(uint32_t *rng, int64_t n, int64_t first_global, int policy, void *stream);

}  // namespace afe
