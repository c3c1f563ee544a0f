/*
 * agrifly_oracle.c -- CPU restatement of agri-fly's vehicle-step hot path.
 * TEST INFRASTRUCTURE ONLY; see agrifly_oracle.h for the parity status
 * ("parity unpinned" for the rigid-body part, pinned clock and RNG).
 *
 * Plain C, double precision, same operation order as the reference source so
 * that a bit-level comparison is meaningful wherever the reference can be
 * built.  Build with -ffp-contract=off (see Makefile): the reference's own
 * build has no FMA contraction on x86-64.
 *
 * Citations are file:line under /root/reference.
 */
#include "agrifly_oracle.h"

#include <math.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* Vec3 helpers (Common/Common/Math/Vec3.hpp)                                */

static void v3_cross(const double a[3], const double b[3], double o[3]) {
  /* Vec3.hpp:106-109 */
  double x = a[1] * b[2] - a[2] * b[1];
  double y = a[2] * b[0] - a[0] * b[2];
  double z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}

static void m33_mul_v3(const double M[9], const double v[3], double o[3]) {
  /* Vec3.hpp:201-210: outVec starts at 0 and accumulates j = 0,1,2 */
  double r[3];
  for (int i = 0; i < 3; i++) {
    double acc = 0.0;
    for (int j = 0; j < 3; j++) acc += M[3 * i + j] * v[j];
    r[i] = acc;
  }
  o[0] = r[0]; o[1] = r[1]; o[2] = r[2];
}

static void m33f_mul_v3f(const float M[9], const float v[3], float o[3]) {
  /* same template, Real = float (Quadcopter_T.cpp:166,175) */
  float r[3];
  for (int i = 0; i < 3; i++) {
    float acc = 0.0f;
    for (int j = 0; j < 3; j++) acc += M[3 * i + j] * v[j];
    r[i] = acc;
  }
  o[0] = r[0]; o[1] = r[1]; o[2] = r[2];
}

/* ------------------------------------------------------------------------ */
/* Rotation<double> (Common/Common/Math/Rotation.hpp)                        */

void ora_rot_matrix(const double v[4], double R[9]) {
  /* Rotation.hpp:196-220, literal */
  const double r0 = v[0] * v[0];
  const double r1 = v[1] * v[1];
  const double r2 = v[2] * v[2];
  const double r3 = v[3] * v[3];
  R[0] = r0 + r1 - r2 - r3;
  R[1] = 2 * v[1] * v[2] - 2 * v[0] * v[3];
  R[2] = 2 * v[1] * v[3] + 2 * v[0] * v[2];
  R[3] = 2 * v[1] * v[2] + 2 * v[0] * v[3];
  R[4] = r0 - r1 + r2 - r3;
  R[5] = 2 * v[2] * v[3] - 2 * v[0] * v[1];
  R[6] = 2 * v[1] * v[3] - 2 * v[0] * v[2];
  R[7] = 2 * v[2] * v[3] + 2 * v[0] * v[1];
  R[8] = r0 - r1 - r2 + r3;
}

void ora_rotate(const double q[4], const double in[3], double out[3]) {
  /* Rotation.hpp:236-245 */
  double R[9];
  ora_rot_matrix(q, R);
  double x = R[0] * in[0] + R[1] * in[1] + R[2] * in[2];
  double y = R[3] * in[0] + R[4] * in[1] + R[5] * in[2];
  double z = R[6] * in[0] + R[7] * in[1] + R[8] * in[2];
  out[0] = x; out[1] = y; out[2] = z;
}

void ora_rotate_inv(const double q[4], const double in[3], double out[3]) {
  /* Inverse() (Rotation.hpp:68) then operator* (:134) */
  const double qi[4] = {q[0], -q[1], -q[2], -q[3]};
  ora_rotate(qi, in, out);
}

void ora_rot_mul(const double a[4], const double r1[4], double out[4]) {
  /* Rotation.hpp:124-131 with this == a */
  double c0 = r1[0] * a[0] - r1[1] * a[1] - r1[2] * a[2] - r1[3] * a[3];
  double c1 = r1[1] * a[0] + r1[0] * a[1] + r1[3] * a[2] - r1[2] * a[3];
  double c2 = r1[2] * a[0] - r1[3] * a[1] + r1[0] * a[2] + r1[1] * a[3];
  double c3 = r1[3] * a[0] + r1[2] * a[1] - r1[1] * a[2] + r1[0] * a[3];
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void ora_rot_from_rotvec(const double r[3], double out[4]) {
  /* Rotation.hpp:84-97; MIN_ANGLE :39 */
  const double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  if (theta < 4.84813681e-6) {
    out[0] = 1; out[1] = 0; out[2] = 0; out[3] = 0;
    return;
  }
  const double ux = r[0] / theta, uy = r[1] / theta, uz = r[2] / theta;
  out[0] = cos(theta * 0.5);
  out[1] = sin(theta * 0.5) * ux;
  out[2] = sin(theta * 0.5) * uy;
  out[3] = sin(theta * 0.5) * uz;
}

void ora_rot_from_euler_ypr(double y, double p, double r, double o[4]) {
  /* Rotation.hpp:99-110 */
  const double h = 0.5;
  o[0] = cos(h * y) * cos(h * p) * cos(h * r) + sin(h * y) * sin(h * p) * sin(h * r);
  o[1] = cos(h * y) * cos(h * p) * sin(h * r) - sin(h * y) * sin(h * p) * cos(h * r);
  o[2] = cos(h * y) * sin(h * p) * cos(h * r) + sin(h * y) * cos(h * p) * sin(h * r);
  o[3] = sin(h * y) * cos(h * p) * cos(h * r) - cos(h * y) * sin(h * p) * sin(h * r);
}

void ora_rot_to_euler_ypr(const double v[4], double ypr[3]) {
  /* Rotation.hpp:163-169 */
  ypr[0] = atan2(2.0 * v[1] * v[2] + 2.0 * v[0] * v[3],
                 v[1] * v[1] + v[0] * v[0] - v[3] * v[3] - v[2] * v[2]);
  ypr[1] = -asin(2.0 * v[1] * v[3] - 2.0 * v[0] * v[2]);
  ypr[2] = atan2(2.0 * v[2] * v[3] + 2.0 * v[0] * v[1],
                 v[3] * v[3] - v[2] * v[2] - v[1] * v[1] + v[0] * v[0]);
}

static void rotf_from_euler_ypr(float y, float p, float r, float o[4]) {
  /* Rotationf instance of Rotation.hpp:99-110 (cosf/sinf, :261-267) */
  const float h = 0.5f;
  o[0] = cosf(h * y) * cosf(h * p) * cosf(h * r) + sinf(h * y) * sinf(h * p) * sinf(h * r);
  o[1] = cosf(h * y) * cosf(h * p) * sinf(h * r) - sinf(h * y) * sinf(h * p) * cosf(h * r);
  o[2] = cosf(h * y) * sinf(h * p) * cosf(h * r) + sinf(h * y) * cosf(h * p) * sinf(h * r);
  o[3] = sinf(h * y) * cosf(h * p) * cosf(h * r) - cosf(h * y) * sinf(h * p) * sinf(h * r);
}

static void rotf_matrix(const float v[4], float R[9]) {
  /* Rotationf instance of Rotation.hpp:196-220 */
  const float r0 = v[0] * v[0], r1 = v[1] * v[1], r2 = v[2] * v[2], r3 = v[3] * v[3];
  R[0] = r0 + r1 - r2 - r3;
  R[1] = 2 * v[1] * v[2] - 2 * v[0] * v[3];
  R[2] = 2 * v[1] * v[3] + 2 * v[0] * v[2];
  R[3] = 2 * v[1] * v[2] + 2 * v[0] * v[3];
  R[4] = r0 - r1 + r2 - r3;
  R[5] = 2 * v[2] * v[3] - 2 * v[0] * v[1];
  R[6] = 2 * v[1] * v[3] - 2 * v[0] * v[2];
  R[7] = 2 * v[2] * v[3] + 2 * v[0] * v[1];
  R[8] = r0 - r1 - r2 + r3;
}

/* ------------------------------------------------------------------------ */
/* Vehicle constants                                                         */

static void m33_inverse(const double a[9], double o[9]) {
  /* _inertiaMatrixInv(inertiaMatrix.inverse()) Quadcopter_T.cpp:20 -- Eigen
   * arithmetic (cofactors times 1/det), not under /root/reference; <= 2 ulp
   * ambiguity versus the real Eigen. */
  double c00 = a[4] * a[8] - a[5] * a[7];
  double c01 = a[5] * a[6] - a[3] * a[8];
  double c02 = a[3] * a[7] - a[4] * a[6];
  double det = a[0] * c00 + a[1] * c01 + a[2] * c02;
  double id = 1.0 / det;
  o[0] = c00 * id;
  o[1] = (a[2] * a[7] - a[1] * a[8]) * id;
  o[2] = (a[1] * a[5] - a[2] * a[4]) * id;
  o[3] = c01 * id;
  o[4] = (a[0] * a[8] - a[2] * a[6]) * id;
  o[5] = (a[2] * a[3] - a[0] * a[5]) * id;
  o[6] = c02 * id;
  o[7] = (a[1] * a[6] - a[0] * a[7]) * id;
  o[8] = (a[0] * a[4] - a[1] * a[3]) * id;
}

void ora_params_init(ora_params *p, double mass, const double inertia[9],
                     double arm_length, const double com_error[3],
                     double motor_min_speed, double motor_max_speed,
                     double k_thrust, double k_torque, double motor_time_const,
                     double motor_inertia, const double lin_drag[3],
                     float imu_yaw, float imu_pitch, float imu_roll) {
  memset(p, 0, sizeof(*p));
  p->mass = mass;
  memcpy(p->inertia, inertia, sizeof(p->inertia));
  m33_inverse(inertia, p->inertia_inv);
  /* Quadcopter_T.cpp:45-65: FR, RR, RL, FL */
  static const double sx[4] = {+1, -1, -1, +1};
  static const double sy[4] = {-1, -1, +1, +1};
  static const double spin[4] = {+1, -1, +1, -1};
  static const int clockwise[4] = {1, 0, 1, 0};
  const double a = arm_length / sqrt(2);
  for (int i = 0; i < 4; i++) {
    p->motor_pos[i][0] = a * sx[i] + com_error[0];
    p->motor_pos[i][1] = a * sy[i] + com_error[1];
    p->motor_pos[i][2] = a * 0.0 + com_error[2];
    p->motor_rot_axis[i][0] = spin[i] * 0.0;
    p->motor_rot_axis[i][1] = spin[i] * 0.0;
    p->motor_rot_axis[i][2] = spin[i] * 1.0;
    for (int k = 0; k < 3; k++) {
      /* Motor.cpp:32-36 (operator-() is *(-1), Vec3.hpp:143-145) */
      p->motor_thrust_axis[i][k] = clockwise[i] ? p->motor_rot_axis[i][k]
                                                : p->motor_rot_axis[i][k] * -1.0;
    }
  }
  p->motor_min_speed = motor_min_speed;
  p->motor_max_speed = motor_max_speed;
  p->k_thrust = k_thrust;
  p->k_torque = k_torque;
  p->motor_time_const = motor_time_const;
  p->motor_inertia = motor_inertia;
  memcpy(p->lin_drag, lin_drag, sizeof(p->lin_drag));
  /* Quadcopter_T.cpp:78-80 */
  float q[4];
  rotf_from_euler_ypr(imu_yaw, imu_pitch, imu_roll, q);
  const float qi[4] = {q[0], -q[1], -q[2], -q[3]};
  rotf_matrix(qi, p->R_imu_inv);
  p->sigma_acc = 0.2;  /* Quadcopter_T.cpp:5 */
  p->sigma_gyro = 0.1; /* Quadcopter_T.cpp:6 */
}

static float max_cf_speed(float c[3][2]) {
  /* QuadcopterConstants.hpp:370-392, float arithmetic */
  int MAX_PWM = 255;
  float MAX_BATT = 4.1;
  float k_1 = c[0][0] + c[0][1] * MAX_BATT;
  float k_2 = c[1][0] + c[1][1] * MAX_BATT;
  float k_3 = c[2][0] + c[2][1] * MAX_BATT;
  return (-k_2 + sqrtf(powf(k_2, 2) - 4 * k_3 * (k_1 - MAX_PWM))) / (2 * k_3);
}

static float max_esc_speed(float c0, float c1) {
  /* QuadcopterConstants.hpp:394-405 */
  int ESC_PERIOD_MAX = 2000;
  return (ESC_PERIOD_MAX - c0) / c1;
}

int ora_type_from_id(unsigned id) {
  /* QuadcopterConstants.hpp:297-332 */
  switch (id) {
    case 3: case 4: case 10: return 1;
    case 2: case 5: case 6: case 7: case 9: case 12: case 15: case 17: return 2;
    case 13: case 14: case 18: case 19: return 4;
    case 1: case 16: case 20: case 21: case 22: case 24: case 26: return 5;
    default: return 0;
  }
}

int ora_params_from_type(ora_params *p, int type) {
  /* float members as in QuadcopterConstants.hpp:334-349, widened to double the
   * way Simulator/Rappids_Simulator/main.cpp:152-164,203-209 does. */
  float mass, ixx, izz, arm, kf, tau_per_thrust, dx, dy, dz;
  float t_const = 0, m_inertia = 0, min_speed = 0, max_speed = 10000; /* :42-45 */
  float yaw = 0, pitch = 0, roll = 0;
  switch (type) {
    case 1: { /* QC_TYPE_CF_STANDARD :54-90 */
      mass = 38e-3; ixx = 16e-6f; izz = 29e-6f; arm = 46e-3f;
      kf = (float)3.58e-8f; tau_per_thrust = 0.0006;
      float c[3][2] = {{-86.19993685f, 22.87189816f},
                       {0.30208677f, -0.07345602f},
                       {-1.59346434e-05f, 1.53209239e-05f}};
      max_speed = max_cf_speed(c);
      dx = 0.0f; dy = 0.0f; dz = 0.0f;
      break;
    }
    case 2: { /* QC_TYPE_CF_BIGMOTORSPROPS :91-124 */
      mass = 39e-3; ixx = 30e-6f; izz = 60e-6f; arm = 48e-3f;
      kf = (float)4.14e-8f; tau_per_thrust = 0.001;
      float c[3][2] = {{-379.31113434f, 84.84738207f},
                       {0.65309704f, -0.13852527f},
                       {-1.34462353e-04f, 3.57662798e-05f}};
      max_speed = max_cf_speed(c);
      dx = 0.0206185f; dy = 0.0216621f; dz = 0.0f;
      break;
    }
    case 4: /* QC_TYPE_CF_LARGEQUAD :157-195 */
      mass = 0.760; ixx = 0.004406f; izz = 0.008611f; arm = 0.166f;
      kf = 7.64e-6f; tau_per_thrust = 0.0140f;
      max_speed = max_esc_speed(972.0f, 0.742f);
      dx = 0.1286181f; dy = 0.1286181f; dz = 0.1286181f;
      break;
    case 5: /* QC_TYPE_CF_MINIQUAD :196-235 */
      mass = 0.142; ixx = 92.7e-6f; izz = 158.57e-6f; arm = 58e-3f;
      kf = 4.32e-8f; tau_per_thrust = 0.00808f;
      max_speed = max_esc_speed(999.0f, 0.14f);
      dx = 0.0f; dy = 0.0f; dz = 0.0f;
      break;
    default:
      return -1; /* INVALID / FEEDTHROUGH: valid == false :125-156,237-266 */
  }
  /* main.cpp:152-164: note k_torque is a float*float product, then widened */
  const double inertia[9] = {ixx, 0, 0, 0, ixx, 0, 0, 0, izz};
  const double com_error[3] = {0, 0, 0}; /* main.cpp:171 */
  const double drag[3] = {dx, dy, dz};   /* main.cpp:207-209 */
  const float ktau = tau_per_thrust * kf;
  ora_params_init(p, mass, inertia, arm, com_error, min_speed, max_speed, kf,
                  ktau, t_const, m_inertia, drag, yaw, pitch, roll);
  return 0;
}

void ora_state_init(ora_state *s) {
  /* SimulationObject6DOF.hpp:14-19; Motor.cpp:18; default engine seed 1 */
  memset(s, 0, sizeof(*s));
  s->att[0] = 1.0;
  s->rng = 1u;
}

/* ------------------------------------------------------------------------ */
/* IMU noise: libstdc++ <random> as used at Quadcopter_T.hpp:122-123          */

uint32_t ora_minstd_next(uint32_t *state) {
  /* std::minstd_rand0 = linear_congruential_engine<uint_fast32_t,16807,0,
   * 2147483647> */
  *state = (uint32_t)(((uint64_t)(*state) * 16807u) % 2147483647u);
  return *state;
}

double ora_canonical(uint32_t *state) {
  /* std::generate_canonical<double,53>(minstd_rand0): bits/random.tcc.
   * R = max-min+1 = 2147483646, k = 2 calls; sum and tmp are double, the
   * second tmp is (double)(R*R) from a long double product. */
  const double R1 = 2147483646.0;
  const double R2 = 4611686009837453312.0; /* (double)(2147483646.0L^2) */
  double sum = 0.0;
  sum += (double)(ora_minstd_next(state) - 1u) * 1.0;
  sum += (double)(ora_minstd_next(state) - 1u) * R1;
  double ret = sum / R2;
  if (ret >= 1.0) ret = nextafter(1.0, 0.0);
  return ret;
}

void ora_normal_pair(uint32_t *state, double *first, double *second) {
  /* std::normal_distribution<double>(0,1)::operator(): Marsaglia polar; the
   * first call returns y*mult and saves x*mult for the second call. */
  double x, y, r2;
  do {
    x = 2.0 * ora_canonical(state) - 1.0;
    y = 2.0 * ora_canonical(state) - 1.0;
    r2 = x * x + y * y;
  } while (r2 > 1.0 || r2 == 0.0);
  const double mult = sqrt(-2 * log(r2) / r2);
  *first = y * mult * 1.0 + 0.0;
  *second = x * mult * 1.0 + 0.0;
}

/* ------------------------------------------------------------------------ */
/* Motor::Run  (Components/Components/Simulation/Motor.cpp:39-84)            */

double ora_motor_run(const ora_params *p, int m, double speed, double speed_cmd,
                     double dt, double thrust[3], double torque[3],
                     double ang_mom[3], double *power) {
  const double *rot_axis = p->motor_rot_axis[m];
  const double *thrust_axis = p->motor_thrust_axis[m];
  const double old_speed = speed;                       /* :46 */
  if (speed_cmd < 0) speed_cmd = 0;                     /* :48-50 */
  double c;
  if (p->motor_time_const == 0) c = 0;                  /* :54-58 */
  else c = exp(-dt / p->motor_time_const);
  speed = c * speed + (1 - c) * speed_cmd;              /* :60 */
  if (speed > p->motor_max_speed) speed = p->motor_max_speed;       /* :62-66 */
  else if (speed < p->motor_min_speed) speed = p->motor_min_speed;
  const double am = speed * p->motor_inertia;           /* :68 */
  const double th = p->k_thrust * speed * fabs(speed);  /* :70 */
  const double aero = -p->k_torque * speed * fabs(speed); /* :73 */
  for (int k = 0; k < 3; k++) {
    ang_mom[k] = am * rot_axis[k];
    thrust[k] = th * thrust_axis[k];
    torque[k] = 0.0 + aero * rot_axis[k];               /* :71-73 */
  }
  double arm_torque[3];
  v3_cross(p->motor_pos[m], thrust, arm_torque);        /* :76 */
  const double ang_acc = (speed - old_speed) / dt;      /* :78 */
  const double spin_up = ang_acc * p->motor_inertia;    /* :79 */
  for (int k = 0; k < 3; k++) {
    torque[k] = torque[k] + arm_torque[k];
    torque[k] = torque[k] - spin_up * rot_axis[k];
  }
  if (power)                                            /* :81 */
    *power = speed * sqrt(torque[0] * torque[0] + torque[1] * torque[1] +
                          torque[2] * torque[2]);
  return speed;
}

/* ------------------------------------------------------------------------ */
/* Quadcopter_T<logicType>::Run (Quadcopter_T.cpp:85-203), dt >= 1e-6 part    */

void ora_quad_step(const ora_params *p, ora_state *s, const float motor_cmd[4],
                   const double ext_force[3], const double ext_torque[3],
                   double dt, int logic_tick, float gyro[3], float acc_meas[3],
                   double acc_world[3]) {
  ora_quad_step_normals(p, s, motor_cmd, ext_force, ext_torque, dt, logic_tick, NULL, gyro, acc_meas, acc_world);
}

/* The same step with the six N(0,1) values of a logic tick supplied by the caller in DRAW order (normals6[0] is what the
 * reference's first draw would be: it lands on gyro z, :167-169) -- for noise policies that are not the libstdc++ stream
 * (agrifly_oracle_counter.c); s->rng is then neither read nor advanced.  normals6 == NULL: drawn from s->rng, i.e.
 * ora_quad_step. */
void ora_quad_step_normals(const ora_params *p, ora_state *s, const float motor_cmd[4],
                           const double ext_force[3], const double ext_torque[3],
                           double dt, int logic_tick, const double *normals6, float gyro[3], float acc_meas[3],
                           double acc_world[3]) {
  static const double zero3[3] = {0, 0, 0};
  if (!ext_force) ext_force = zero3;
  if (!ext_torque) ext_torque = zero3;

  double total_force_b[3] = {0, 0, 0};                  /* :93-94 */
  double total_torque_b[3] = {0, 0, 0};
  double motor_L[4][3];
  for (int i = 0; i < 4; i++) {                         /* :97-104 */
    double thrust[3], torque[3];
    s->motor_speed[i] = ora_motor_run(p, i, s->motor_speed[i],
                                      (double)motor_cmd[i], dt, thrust, torque,
                                      motor_L[i], 0);
    for (int k = 0; k < 3; k++) {
      total_force_b[k] = total_force_b[k] + thrust[k];
      total_torque_b[k] = total_torque_b[k] + torque[k];
    }
  }

  double ext_torque_b[3];                               /* :106 */
  ora_rotate_inv(s->att, ext_torque, ext_torque_b);
  for (int k = 0; k < 3; k++) total_torque_b[k] = total_torque_b[k] + ext_torque_b[k];

  double ang_mom[3];                                    /* :113-117 */
  m33_mul_v3(p->inertia, s->ang_vel, ang_mom);
  for (int i = 0; i < 4; i++)
    for (int k = 0; k < 3; k++) ang_mom[k] = ang_mom[k] + motor_L[i][k];

  double gyro_term[3], net_torque[3], ang_acc[3];       /* :119-120 */
  v3_cross(s->ang_vel, ang_mom, gyro_term);
  for (int k = 0; k < 3; k++) net_torque[k] = total_torque_b[k] - gyro_term[k];
  m33_mul_v3(p->inertia_inv, net_torque, ang_acc);

  double vel_b[3];                                      /* :123-128 */
  ora_rotate_inv(s->att, s->vel, vel_b);
  for (int k = 0; k < 3; k++)
    total_force_b[k] = total_force_b[k] + p->lin_drag[k] * (-vel_b[k]);

  double acc[3] = {0, 0, -9.81};                        /* :131-132 */
  double force_w[3];
  ora_rotate(s->att, total_force_b, force_w);
  for (int k = 0; k < 3; k++)
    acc[k] = acc[k] + (force_w[k] + ext_force[k]) / p->mass;

  double newpos[3], newvel[3], newatt[4], newangvel[3], rotvec[3], dq[4];
  for (int k = 0; k < 3; k++) {                         /* :140-143 */
    newpos[k] = (s->pos[k] + dt * s->vel[k]) + dt * (dt * (0.5 * acc[k]));
    newvel[k] = s->vel[k] + dt * acc[k];
    rotvec[k] = dt * s->ang_vel[k];
    newangvel[k] = s->ang_vel[k] + dt * ang_acc[k];
  }
  ora_rot_from_rotvec(rotvec, dq);
  ora_rot_mul(s->att, dq, newatt);

  if ((newpos[2] <= 0) && (newvel[2] < 0)) {            /* :146-151 */
    newpos[2] = 0;
    newvel[2] = 0;
    acc[2] = 0;
    newangvel[0] = 0; newangvel[1] = 0; newangvel[2] = 0;
  }

  for (int k = 0; k < 3; k++) {                         /* :153-156 */
    s->pos[k] = newpos[k];
    s->vel[k] = newvel[k];
    s->ang_vel[k] = newangvel[k];
  }
  for (int k = 0; k < 4; k++) s->att[k] = newatt[k];
  if (acc_world) { acc_world[0] = acc[0]; acc_world[1] = acc[1]; acc_world[2] = acc[2]; }

  if (logic_tick) {                                     /* :159-183 */
    /* g++ evaluates the three ctor arguments right to left:
     * z <- 1st draw, y <- 2nd, x <- 3rd; six draws = three polar pairs. */
    double n[6];
    if (normals6) {
      for (int k = 0; k < 6; k++) n[k] = normals6[k];
    } else {
      ora_normal_pair(&s->rng, &n[0], &n[1]);
      ora_normal_pair(&s->rng, &n[2], &n[3]);
      ora_normal_pair(&s->rng, &n[4], &n[5]);
    }

    float w_f[3] = {(float)s->ang_vel[0], (float)s->ang_vel[1], (float)s->ang_vel[2]}; /* :165 */
    float g[3];
    m33f_mul_v3f(p->R_imu_inv, w_f, g);                 /* :166 */
    const float sg = (float)p->sigma_gyro;              /* :170 */
    g[0] = g[0] + sg * (float)n[2];                     /* :167-170 */
    g[1] = g[1] + sg * (float)n[1];
    g[2] = g[2] + sg * (float)n[0];

    double proper[3] = {acc[0] + 0, acc[1] + 0, acc[2] + 9.81}; /* :174 */
    double proper_b[3];
    ora_rotate_inv(s->att, proper, proper_b);
    float a_f[3] = {(float)proper_b[0], (float)proper_b[1], (float)proper_b[2]};
    float a[3];
    m33f_mul_v3f(p->R_imu_inv, a_f, a);                 /* :175 */
    const float sa = (float)p->sigma_acc;               /* :179 */
    a[0] = a[0] + sa * (float)n[5];                     /* :176-179 */
    a[1] = a[1] + sa * (float)n[4];
    a[2] = a[2] + sa * (float)n[3];
    if (gyro) { gyro[0] = g[0]; gyro[1] = g[1]; gyro[2] = g[2]; }
    if (acc_meas) { acc_meas[0] = a[0]; acc_meas[1] = a[1]; acc_meas[2] = a[2]; }
  }
}

/* ------------------------------------------------------------------------ */
/* Clock (Timer.hpp:27-54, ManualTimer.hpp:29-40)                            */

void ora_clock_init(ora_clock *c, double logic_period) {
  c->now_us = 0;          /* ManualTimer ctor */
  c->integ_reset_us = 0;  /* Timer ctor -> Reset() */
  c->logic_reset_us = 0;
  c->logic_period = logic_period;
}

double ora_clock_run(ora_clock *c, int *tick) {
  *tick = 0;
  /* Quadcopter_T.cpp:87-91; Timer::GetSeconds<double> Timer.hpp:36-38 */
  const double dt = (double)((double)(c->now_us - c->integ_reset_us) * 1e-6);
  if (dt < 1e-6) return 0.0;
  c->integ_reset_us = c->now_us;
  /* Quadcopter_T.cpp:159-160; AdjustTimeBySeconds Timer.hpp:27-33 with a
   * negative argument: _lastResetTime_usec += uint64_t(x * -1e6) */
  const double el = (double)((double)(c->now_us - c->logic_reset_us) * 1e-6);
  if (el > c->logic_period) {
    c->logic_reset_us += (uint64_t)((-c->logic_period) * -1e6);
    *tick = 1;
  }
  return dt;
}

void ora_clock_advance(ora_clock *c, uint64_t dt_us) { c->now_us += dt_us; }

/* ------------------------------------------------------------------------ */
/* Batched SoA driver                                                        */

static int g_batch_threads = 1;
/* threads ora_step_batch spreads the vehicles over (OpenMP); 1 = the scalar port */
void ora_set_batch_threads(int n) { g_batch_threads = n > 0 ? n : 1; }
int ora_get_batch_threads(void) { return g_batch_threads; }

void ora_step_batch(int64_t n, int n_steps, const ora_params *table,
                    const uint8_t *types, double *pos, double *vel,
                    double *att, double *ang_vel, double *motor_speed,
                    uint32_t *rng, const float *motor_cmd,
                    const double *ext_force, const double *ext_torque,
                    double dt, const uint8_t *tick_per_step, float *gyro,
                    float *acc) {
#pragma omp parallel for schedule(static) num_threads(g_batch_threads) if (g_batch_threads > 1)
  for (int64_t i = 0; i < n; i++) {
    const ora_params *p = &table[types ? types[i] : 0];
    ora_state s;
    for (int k = 0; k < 3; k++) {
      s.pos[k] = pos[k * n + i];
      s.vel[k] = vel[k * n + i];
      s.ang_vel[k] = ang_vel[k * n + i];
    }
    for (int k = 0; k < 4; k++) {
      s.att[k] = att[k * n + i];
      s.motor_speed[k] = motor_speed[k * n + i];
    }
    s.rng = rng ? rng[i] : 1u;
    float cmd[4] = {motor_cmd[0 * n + i], motor_cmd[1 * n + i],
                    motor_cmd[2 * n + i], motor_cmd[3 * n + i]};
    double fe[3] = {0, 0, 0}, te[3] = {0, 0, 0};
    if (ext_force) for (int k = 0; k < 3; k++) fe[k] = ext_force[k * n + i];
    if (ext_torque) for (int k = 0; k < 3; k++) te[k] = ext_torque[k * n + i];
    float g[3], a[3];
    for (int st = 0; st < n_steps; st++) {
      const int tick = tick_per_step ? tick_per_step[st] : 0;
      ora_quad_step(p, &s, cmd, fe, te, dt, tick, g, a, 0);
      if (tick) {
        for (int k = 0; k < 3; k++) {
          if (gyro) gyro[k * n + i] = g[k];
          if (acc) acc[k * n + i] = a[k];
        }
      }
    }
    for (int k = 0; k < 3; k++) {
      pos[k * n + i] = s.pos[k];
      vel[k * n + i] = s.vel[k];
      ang_vel[k * n + i] = s.ang_vel[k];
    }
    for (int k = 0; k < 4; k++) {
      att[k * n + i] = s.att[k];
      motor_speed[k * n + i] = s.motor_speed[k];
    }
    if (rng) rng[i] = s.rng;
  }
}
