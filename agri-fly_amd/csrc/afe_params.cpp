// afe_params.cpp -- host-only part of the engine: vehicle-type table, expansion
// of the Quadcopter_T constructor arguments into kernel constants, and the
// logic-gate planner.  No HIP calls in this file.
#include "afe_host.h"

#include <cmath>
#include <cstring>

namespace afe {

namespace {

// Rotationf::FromEulerYPR(y,p,r).Inverse().GetRotationMatrix() in float,
// reference Quadcopter_T.cpp:78-80 with Rotation.hpp:68,99-110,196-220.
void imu_mount_matrix(float yaw, float pitch, float roll, float M[9]) {
  const float h = 0.5f;
  const float cy = cosf(h * yaw), sy = sinf(h * yaw);
  const float cp = cosf(h * pitch), sp = sinf(h * pitch);
  const float cr = cosf(h * roll), sr = sinf(h * roll);
  const float q0 = cy * cp * cr + sy * sp * sr;
  const float q1 = -(cy * cp * sr - sy * sp * cr);
  const float q2 = -(cy * sp * cr + sy * cp * sr);
  const float q3 = -(sy * cp * cr - cy * sp * sr);
  const float r0 = q0 * q0, r1 = q1 * q1, r2 = q2 * q2, r3 = q3 * q3;
  M[0] = r0 + r1 - r2 - r3;
  M[1] = 2 * q1 * q2 - 2 * q0 * q3;
  M[2] = 2 * q1 * q3 + 2 * q0 * q2;
  M[3] = 2 * q1 * q2 + 2 * q0 * q3;
  M[4] = r0 - r1 + r2 - r3;
  M[5] = 2 * q2 * q3 - 2 * q0 * q1;
  M[6] = 2 * q1 * q3 - 2 * q0 * q2;
  M[7] = 2 * q2 * q3 + 2 * q0 * q1;
  M[8] = r0 - r1 - r2 + r3;
}

// 3x3 inverse by cofactors (what Eigen's fixed-size inverse() does for
// Quadcopter_T.cpp:20).
bool invert3(const double a[9], double o[9]) {
  const double c0 = a[4] * a[8] - a[5] * a[7];
  const double c1 = a[5] * a[6] - a[3] * a[8];
  const double c2 = a[3] * a[7] - a[4] * a[6];
  const double det = a[0] * c0 + a[1] * c1 + a[2] * c2;
  if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return false;
  const double id = 1.0 / det;
  o[0] = c0 * id;
  o[1] = (a[2] * a[7] - a[1] * a[8]) * id;
  o[2] = (a[1] * a[5] - a[2] * a[4]) * id;
  o[3] = c1 * id;
  o[4] = (a[0] * a[8] - a[2] * a[6]) * id;
  o[5] = (a[2] * a[3] - a[0] * a[5]) * id;
  o[6] = c2 * id;
  o[7] = (a[1] * a[6] - a[0] * a[7]) * id;
  o[8] = (a[0] * a[4] - a[1] * a[3]) * id;
  return true;
}

}  // namespace

int expand_params(const afe_vehicle_params &in, HostParams &out, const char **why) {
  // the reference's constructor-time invariants (Motor.cpp:27-30)
  if (!(in.prop_thrust_from_speed_sqr >= 0)) { *why = "prop_thrust_from_speed_sqr must be >= 0"; return AFE_ERR_INVALID_ARG; }
  if (!(in.prop_torque_from_speed_sqr >= 0)) { *why = "prop_torque_from_speed_sqr must be >= 0"; return AFE_ERR_INVALID_ARG; }
  if (!(in.motor_max_speed > in.motor_min_speed)) { *why = "motor_max_speed must exceed motor_min_speed"; return AFE_ERR_INVALID_ARG; }
  // the lag-free rotor's clamp is one v_med3 between max(motor_min_speed, 0) and motor_max_speed (afe_kernels.hip): that is
  // Motor.cpp:48-66's max(0, cmd) followed by its clamp only while the lower bound does not exceed the upper one
  if (!(in.motor_max_speed > 0)) { *why = "motor_max_speed must be positive"; return AFE_ERR_INVALID_ARG; }
  if (!(in.mass > 0)) { *why = "mass must be > 0"; return AFE_ERR_INVALID_ARG; }
  if (!(in.motor_time_const >= 0)) { *why = "motor_time_const must be >= 0"; return AFE_ERR_INVALID_ARG; }
  out.mass = in.mass;
  std::memcpy(out.I, in.inertia, sizeof(out.I));
  if (!invert3(in.inertia, out.Iinv)) { *why = "inertia matrix is singular"; return AFE_ERR_INVALID_ARG; }
  // Quadcopter_T.cpp:45-65: FR(+,-) RR(-,-) RL(-,+) FL(+,+) at arm/sqrt(2)
  static const double sx[4] = {+1, -1, -1, +1};
  static const double sy[4] = {-1, -1, +1, +1};
  const double a = in.arm_length / std::sqrt(2);
  for (int m = 0; m < 4; m++) {
    out.mp[m][0] = a * sx[m] + in.com_error[0];
    out.mp[m][1] = a * sy[m] + in.com_error[1];
    out.mp[m][2] = a * 0.0 + in.com_error[2];
  }
  out.kf = in.prop_thrust_from_speed_sqr;
  out.ktau = in.prop_torque_from_speed_sqr;
  out.tau_m = in.motor_time_const;
  out.Jm = in.motor_inertia;
  out.wmin = in.motor_min_speed;
  out.wmax = in.motor_max_speed;
  for (int k = 0; k < 3; k++) out.drag[k] = in.lin_drag_coeff_b[k];
  imu_mount_matrix(in.imu_yaw, in.imu_pitch, in.imu_roll, out.Rimu);
  return AFE_OK;
}

int expand_logic(const afe_rates_logic_params &in, float onboard_period, DevLogic &g, const char **why) {
  if (!(in.mass > 0) || !(in.ang_vel_time_const_xy > 0) || !(in.ang_vel_time_const_z > 0) ||
      !(in.arm_length > 0) || !(in.prop_thrust_from_speed_sqr > 0) || !(in.prop_torque_from_thrust != 0) ||
      (in.prop0_spin_dir != 1 && in.prop0_spin_dir != -1) || !(in.gyro_lowpass_cutoff > 0)) {
    *why = "rates logic parameters out of range";
    return AFE_ERR_INVALID_ARG;
  }
  std::memset(&g, 0, sizeof(g));
  g.mass = in.mass;
  for (int k = 0; k < 9; k++) g.I[k] = in.inertia[k];
  g.tc_xy = in.ang_vel_time_const_xy;
  g.tc_z = in.ang_vel_time_const_z;
  // QuadcopterMixer::SetParameters, reference QuadcopterMixer.hpp:36-52
  g.d = in.arm_length / sqrtf(2.0f);
  g.kt = in.prop0_spin_dir * in.prop_torque_from_thrust;
  g.kf = in.prop_thrust_from_speed_sqr;
  g.max_thrust = in.max_thrust_per_propeller;
  g.min_thrust = in.min_thrust_per_propeller;
  g.max_cmd_total = in.max_cmd_total_thrust < 0 ? 4 * in.max_thrust_per_propeller * 0.8f : in.max_cmd_total_thrust;
  // _R = Rotationf::FromEulerYPR(yaw, pitch, roll).GetRotationMatrix(), QuadcopterLogic.cpp:116-117:
  // the transpose of the (bitwise transposable) inverse-mount matrix
  float Rinv[9];
  imu_mount_matrix(in.imu_yaw, in.imu_pitch, in.imu_roll, Rinv);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) g.R[3 * i + j] = Rinv[3 * j + i];
  // LowPassFilterSecondOrder<float,...>::Initialise, LowPassFilterSecondOrder.hpp:22-49
  const float dt = onboard_period, wc = in.gyro_lowpass_cutoff;
  const float sqrt2 = float(std::sqrt(2.0));
  g.a1 = (dt * dt * wc * wc - 2 * sqrt2 * dt * wc + 4) / (dt * dt * wc * wc + 2 * sqrt2 * dt * wc + 4);
  g.a2 = 2 * (dt * dt * wc * wc - 4) / (dt * dt * wc * wc + 2 * sqrt2 * dt * wc + 4);
  g.b0 = dt * dt * wc * wc / (dt * dt * wc * wc + 2 * sqrt2 * dt * wc + 4);
  g.b1 = dt * dt * wc * wc / (dt * dt * wc * wc + 2 * sqrt2 * dt * wc + 4);
  g.b2 = 2 * dt * dt * wc * wc / (dt * dt * wc * wc + 2 * sqrt2 * dt * wc + 4);
  return AFE_OK;
}

template <typename R>
void to_device_params(const HostParams &h, double dt, DevParams<R> &d) {
  std::memset(&d, 0, sizeof(d));
  d.mass = (R)h.mass;
  d.inv_mass = (R)(1.0 / h.mass);
  for (int k = 0; k < 9; k++) { d.I[k] = (R)h.I[k]; d.Iinv[k] = (R)h.Iinv[k]; d.Rimu[k] = h.Rimu[k]; }
  for (int m = 0; m < 4; m++) { d.mpx[m] = (R)h.mp[m][0]; d.mpy[m] = (R)h.mp[m][1]; d.mpz[m] = (R)h.mp[m][2]; }
  d.kf = (R)h.kf;
  d.ktau = (R)h.ktau;
  // Motor.cpp:54-58, evaluated in double like the reference
  d.c_lag = (R)((h.tau_m == 0) ? 0.0 : std::exp(-dt / h.tau_m));
  d.omc_lag = (R)((h.tau_m == 0) ? 1.0 : -std::expm1(-dt / h.tau_m));
  d.Jm = (R)h.Jm;
  d.wmin = (R)h.wmin;
  d.wmax = (R)h.wmax;
  for (int k = 0; k < 3; k++) d.drag[k] = (R)h.drag[k];
  d.kf_over_mass_d = h.kf / h.mass;
}
template void to_device_params<float>(const HostParams &, double, DevParams<float> &);
template void to_device_params<double>(const HostParams &, double, DevParams<double> &);

}  // namespace afe

// ---------------------------------------------------------------------------
// C ABI: pure host entry points

namespace {

// GetMaxCFSpeedFromPWMConsts / GetMaxESCSpeedFromPWMConsts, reference
// Components/Components/Logic/QuadcopterConstants.hpp:370-405 (float maths).
float cf_max_speed(const float k[3][2]) {
  const int max_pwm = 255;
  const float max_batt = 4.1;
  const float k1 = k[0][0] + k[0][1] * max_batt;
  const float k2 = k[1][0] + k[1][1] * max_batt;
  const float k3 = k[2][0] + k[2][1] * max_batt;
  return (-k2 + sqrtf(powf(k2, 2) - 4 * k3 * (k1 - max_pwm))) / (2 * k3);
}
float esc_max_speed(float offset, float slope) {
  const int esc_period_max = 2000;
  return (esc_period_max - offset) / slope;
}

struct TypeRow {  // the float members the simulator loops read
  float mass, ixx, izz, arm, kf, torque_per_thrust, max_speed, drag[3];
};

struct LogicRow {  // controller / mixer members, QuadcopterConstants.hpp:34-47 and per type
  float tc_xy, tc_z, min_thrust, max_cmd_frac;  // max_cmd_frac < 0: mixer default
};
bool logic_row(int type, LogicRow &r) {
  switch (type) {
    case 1: r = {0.04f, 0.5f, 0.0f, 0.9f}; return true;     // :78-82
    case 2: r = {0.03f, 0.5f, 0.0f, 0.8f}; return true;     // :36-39,113
    case 4: r = {0.0457f, 0.2545f, 0.0f, -1.0f}; return true;  // :183-186
    case 5: r = {0.04f, 0.04f * 5, 0.03f, 0.7f}; return true;  // :214-224
    default: return false;
  }
}

bool type_row(int type, TypeRow &r) {
  switch (type) {
    case 1: {  // QC_TYPE_CF_STANDARD, QuadcopterConstants.hpp:54-90
      const float k[3][2] = {{-86.19993685f, 22.87189816f}, {0.30208677f, -0.07345602f},
                             {-1.59346434e-05f, 1.53209239e-05f}};
      r = {38e-3, 16e-6f, 29e-6f, 46e-3f, 3.58e-8f, 0.0006, cf_max_speed(k), {0.0f, 0.0f, 0.0f}};
      return true;
    }
    case 2: {  // QC_TYPE_CF_BIGMOTORSPROPS, :91-124
      const float k[3][2] = {{-379.31113434f, 84.84738207f}, {0.65309704f, -0.13852527f},
                             {-1.34462353e-04f, 3.57662798e-05f}};
      r = {39e-3, 30e-6f, 60e-6f, 48e-3f, 4.14e-8f, 0.001, cf_max_speed(k), {0.0206185f, 0.0216621f, 0.0f}};
      return true;
    }
    case 4:  // QC_TYPE_CF_LARGEQUAD, :157-195
      r = {0.760, 0.004406f, 0.008611f, 0.166f, 7.64e-6f, 0.0140f, esc_max_speed(972.0f, 0.742f),
           {0.1286181f, 0.1286181f, 0.1286181f}};
      return true;
    case 5:  // QC_TYPE_CF_MINIQUAD, :196-235
      r = {0.142, 92.7e-6f, 158.57e-6f, 58e-3f, 4.32e-8f, 0.00808f, esc_max_speed(999.0f, 0.14f),
           {0.0f, 0.0f, 0.0f}};
      return true;
    default:  // INVALID / FEEDTHROUGH have valid == false (:125-156,237-266)
      return false;
  }
}

}  // namespace

extern "C" int afe_params_from_type(int quadcopter_type, afe_vehicle_params *out) {
  if (!out) return AFE_ERR_INVALID_ARG;
  TypeRow r;
  if (!type_row(quadcopter_type, r)) return AFE_ERR_INVALID_ARG;
  std::memset(out, 0, sizeof(*out));
  // widening float -> double as Simulator/Rappids_Simulator/main.cpp:152-164
  out->mass = r.mass;
  out->inertia[0] = r.ixx;
  out->inertia[4] = r.ixx;  // inertia_yy = inertia_xx, main.cpp:154
  out->inertia[8] = r.izz;
  out->arm_length = r.arm;
  out->motor_min_speed = 0.0f;   // defaults, QuadcopterConstants.hpp:42-45
  out->motor_max_speed = r.max_speed;
  out->prop_thrust_from_speed_sqr = r.kf;
  const float ktau = r.torque_per_thrust * r.kf;  // float product, main.cpp:158-159
  out->prop_torque_from_speed_sqr = ktau;
  out->motor_time_const = 0.0f;
  out->motor_inertia = 0.0f;
  for (int k = 0; k < 3; k++) out->lin_drag_coeff_b[k] = r.drag[k];
  out->imu_yaw = out->imu_pitch = out->imu_roll = 0.0f;
  return AFE_OK;
}

extern "C" int afe_rates_logic_params_from_type(int quadcopter_type, afe_rates_logic_params *out) {
  if (!out) return AFE_ERR_INVALID_ARG;
  TypeRow r;
  LogicRow l;
  if (!type_row(quadcopter_type, r) || !logic_row(quadcopter_type, l)) return AFE_ERR_INVALID_ARG;
  std::memset(out, 0, sizeof(*out));
  out->mass = r.mass;
  out->inertia[0] = r.ixx; out->inertia[4] = r.ixx; out->inertia[8] = r.izz;  // QuadcopterConstants.hpp:269-271
  out->ang_vel_time_const_xy = l.tc_xy;
  out->ang_vel_time_const_z = l.tc_z;
  out->arm_length = r.arm;
  out->prop_thrust_from_speed_sqr = r.kf;
  out->prop_torque_from_thrust = r.torque_per_thrust;
  out->prop0_spin_dir = 1;
  out->max_thrust_per_propeller = r.kf * powf(r.max_speed, 2);
  out->min_thrust_per_propeller = l.min_thrust;
  out->max_cmd_total_thrust = l.max_cmd_frac < 0 ? -1.0f
      : (quadcopter_type == 5 ? l.max_cmd_frac * (out->max_thrust_per_propeller * 4)   // :221
                              : l.max_cmd_frac * out->max_thrust_per_propeller * 4);   // :80,113
  out->imu_yaw = out->imu_pitch = out->imu_roll = 0.0f;
  out->gyro_lowpass_cutoff = 200.0f;  // QuadcopterLogic.cpp:103
  return AFE_OK;
}

extern "C" int afe_type_from_id(unsigned id) {
  // GetVehicleTypeFromID, QuadcopterConstants.hpp:297-332
  switch (id) {
    case 3: case 4: case 10: return 1;
    case 2: case 5: case 6: case 7: case 9: case 12: case 15: case 17: return 2;
    case 13: case 14: case 18: case 19: return 4;
    case 1: case 16: case 20: case 21: case 22: case 24: case 26: return 5;
    default: return 0;
  }
}

extern "C" int afe_plan_ticks(double logic_period_s, uint64_t *elapsed_us, uint64_t dt_us,
                              int n_steps, uint8_t *tick_out) {
  if (!elapsed_us || n_steps < 0 || (n_steps > 0 && !tick_out)) return AFE_ERR_INVALID_ARG;
  for (int s = 0; s < n_steps; s++) {
    // dt < 1e-6 s: Quadcopter_T::Run returns before touching anything (:88-90)
    if (afe::us_to_seconds(dt_us) < 1e-6) { tick_out[s] = 0; continue; }
    tick_out[s] = afe::gate_step(logic_period_s, *elapsed_us, dt_us) ? 1 : 0;
  }
  return AFE_OK;
}

extern "C" int afe_abi_version(void) { return AFE_ABI_VERSION; }

extern "C" int afe_has_dev_hooks(void) {
#ifdef AFE_DEV_HOOKS
  return 1;
#else
  return 0;
#endif
}

namespace {
template <typename R>
void kernarg_layout(int32_t offsets[4], int32_t sizes[4], int32_t *segment_bytes) {
  using K = afe::PersistKernarg<R>;
  offsets[0] = (int32_t)offsetof(K, v); sizes[0] = (int32_t)sizeof(afe::StepView<R>);
  offsets[1] = (int32_t)offsetof(K, P); sizes[1] = (int32_t)sizeof(afe::DevParams<R>);
  offsets[2] = (int32_t)offsetof(K, G); sizes[2] = (int32_t)sizeof(afe::DevLogic);
  offsets[3] = (int32_t)offsetof(K, a); sizes[3] = (int32_t)sizeof(afe::PersistArgs);
  *segment_bytes = (int32_t)afe::persist_kernarg_bytes<R>();
}
}  // namespace

extern "C" int afe_persistent_kernarg_layout(int precision, int32_t offsets[4], int32_t sizes[4], int32_t *segment_bytes) {
  if (!offsets || !sizes || !segment_bytes || (precision != AFE_F32 && precision != AFE_F64)) return AFE_ERR_INVALID_ARG;
  if (precision == AFE_F64) kernarg_layout<double>(offsets, sizes, segment_bytes);
  else kernarg_layout<float>(offsets, sizes, segment_bytes);
  return AFE_OK;
}

extern "C" const char *afe_status_string(int status) {
  switch (status) {
    case AFE_OK: return "ok";
    case AFE_ERR_INVALID_ARG: return "invalid argument";
    case AFE_ERR_NO_DEVICE: return "no usable gfx950 HIP device";
    case AFE_ERR_HIP: return "HIP runtime error";
    case AFE_ERR_OUT_OF_RANGE: return "vehicle range out of bounds";
    case AFE_ERR_NOT_CONFIGURED: return "engine not configured (type table missing)";
    case AFE_ERR_COMM: return "communicator error";
    default: return "unknown status";
  }
}
