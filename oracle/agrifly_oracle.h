/*
 * agrifly_oracle.h -- CPU restatement of agri-fly's vehicle-step hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the reported CPU baseline.
 * The shipped path is the HIP engine behind include/agrifly_engine.h and it
 * never falls back to this code.
 *
 * PARITY STATUS (see DESIGN.md "Oracle"):
 *   - rigid body, motors, Vec3/Rotation math:
 *     **parity unpinned**.  The reference ships no tests or golden vectors and
 *     its sources for this path include <Eigen/Dense>, which this image lacks,
 *     so the reference cannot be built here without stand-ins.  The functions
 *     below restate the reference source line by line (citations given) in
 *     the same operation order, in double, no FMA contraction.  Informational
 *     (tests/test_reference_anchors.py): flown in the reference's config-1
 *     loop they reproduce, to all nine printed digits, the positions SURVEY.md
 *     Appendix B records for the unmodified reference after 1 s and after
 *     10 s of flight (10 000 steps).
 *   - clock / logic-gate cadence (a6): pinned against the reference's own
 *     Timer/ManualTimer headers, which compile stand-alone (oracle/_ref/
 *     timer_probe, fixture tests/golden/timer_cadence.json).
 *   - IMU noise stream (a7): pinned against libstdc++'s
 *     std::default_random_engine + std::normal_distribution<double>, the
 *     third-party code the reference calls (fixture tests/golden/rng_kat.json).
 *
 * All paths cited are relative to /root/reference.
 */
#ifndef AGRIFLY_ORACLE_H
#define AGRIFLY_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Per-vehicle constant record == the arguments of the Quadcopter_T ctor
 * (Components/Components/Simulation/Quadcopter_T.hpp:24-32) after the ctor
 * body has expanded them (Quadcopter_T.cpp:8-83). */
typedef struct ora_params {
  double mass;              /* _mass                      Quadcopter_T.cpp:21 */
  double inertia[9];        /* _inertiaMatrix, row major  Quadcopter_T.cpp:19 */
  double inertia_inv[9];    /* _inertiaMatrixInv          Quadcopter_T.cpp:20 */
  double motor_pos[4][3];   /* Motor::_position           Quadcopter_T.cpp:47-65 */
  double motor_rot_axis[4][3];    /* Motor::_rotAxis      Quadcopter_T.cpp:45-65 */
  double motor_thrust_axis[4][3]; /* Motor::_thrustAxis   Motor.cpp:32-36 */
  double motor_min_speed;   /* Motor::_minSpeed  */
  double motor_max_speed;   /* Motor::_maxSpeed  */
  double k_thrust;          /* Motor::_thrustFromSpeedSqr */
  double k_torque;          /* Motor::_torqueFromSpeedSqr */
  double motor_time_const;  /* Motor::_timeConstant */
  double motor_inertia;     /* Motor::_inertia */
  double lin_drag[3];       /* _linDragCoeffB             Quadcopter_T.cpp:24 */
  float  R_imu_inv[9];      /* _R_inverse (float)         Quadcopter_T.cpp:78-80 */
  double sigma_acc;         /* ACCELEROMETER_NOISE_STD_DEV Quadcopter_T.cpp:5 */
  double sigma_gyro;        /* RATE_GYRO_NOISE_STD_DEV    Quadcopter_T.cpp:6 */
} ora_params;

/* Mutable per-vehicle state: SimulationObject6DOF.hpp:77-80 + Motor::_speed
 * (Motor.hpp:55) + the libstdc++ engine word (Quadcopter_T.hpp:122). */
typedef struct ora_state {
  double pos[3];
  double vel[3];
  double att[4];     /* scalar first, Rotation.hpp:46-51 */
  double ang_vel[3];
  double motor_speed[4];
  uint32_t rng;      /* minstd_rand0 state; default seed 1 */
} ora_state;

/* Fill p the way the Quadcopter_T ctor does (Quadcopter_T.cpp:8-83):
 * 4 motors at arm/sqrt(2)*(+-1,+-1,0)+comError, spin +z,-z,+z,-z, handedness
 * CW,CCW,CW,CCW; I^-1; IMU mount R^-1 from yaw/pitch/roll (float). */
void ora_params_init(ora_params *p, double mass, const double inertia[9],
                     double arm_length, const double com_error[3],
                     double motor_min_speed, double motor_max_speed,
                     double k_thrust, double k_torque, double motor_time_const,
                     double motor_inertia, const double lin_drag[3],
                     float imu_yaw, float imu_pitch, float imu_roll);

/* Vehicle-type table (Components/Components/Logic/QuadcopterConstants.hpp:
 * 31-274) narrowed to what the loops feed the ctor (Simulator/
 * Rappids_Simulator/main.cpp:147-218).  type: 1 STANDARD, 2 BIGMOTORSPROPS,
 * 4 LARGEQUAD, 5 MINIQUAD.  Returns 0, or -1 for an invalid type. */
int ora_params_from_type(ora_params *p, int quadcopter_type);
int ora_type_from_id(unsigned id); /* QuadcopterConstants.hpp:297-332 */

void ora_state_init(ora_state *s); /* SimulationObject6DOF.hpp:14-19 */

/* One Quadcopter_T::Run() body for dt >= 1e-6 (Quadcopter_T.cpp:85-203).
 * motor_cmd are the float commands of Quadcopter_T.hpp:100.  If logic_tick
 * is non-zero the IMU synthesis of :163-183 runs and writes gyro[3], acc[3]
 * (floats, noise from s->rng in g++'s right-to-left draw order).  Returns the
 * step acceleration through acc_world[3] if not NULL. */
void ora_quad_step(const ora_params *p, ora_state *s, const float motor_cmd[4],
                   const double ext_force[3], const double ext_torque[3],
                   double dt, int logic_tick, float gyro[3], float acc[3],
                   double acc_world[3]);
/* the same step with the tick's six N(0,1) values supplied in draw order (NULL: drawn from s->rng = ora_quad_step) */
void ora_quad_step_normals(const ora_params *p, ora_state *s, const float motor_cmd[4],
                           const double ext_force[3], const double ext_torque[3],
                           double dt, int logic_tick, const double *normals6, float gyro[3], float acc_meas[3],
                           double acc_world[3]);

/* Motor::Run (Motor.cpp:39-84) for one motor; returns new speed, writes
 * thrust[3], torque[3], ang_mom[3] (body frame) and instantaneous power. */
double ora_motor_run(const ora_params *p, int motor, double speed,
                     double speed_cmd, double dt, double thrust[3],
                     double torque[3], double ang_mom[3], double *power);

/* std::minstd_rand0 step and libstdc++ normal_distribution<double>(0,1)
 * pair draw (bits/random.tcc generate_canonical + normal_distribution). */
uint32_t ora_minstd_next(uint32_t *state);
double ora_canonical(uint32_t *state);
void ora_normal_pair(uint32_t *state, double *first, double *second);

/* Rotation<double> helpers (Common/Common/Math/Rotation.hpp). */
void ora_rot_matrix(const double q[4], double R[9]);            /* :196-220 */
void ora_rot_mul(const double a[4], const double b[4], double out[4]); /* :124-131 */
void ora_rot_from_rotvec(const double r[3], double out[4]);     /* :84-97  */
void ora_rot_from_euler_ypr(double y, double p, double r, double out[4]); /* :99-110 */
void ora_rot_to_euler_ypr(const double q[4], double ypr[3]);    /* :163-169 */
void ora_rotate(const double q[4], const double v[3], double out[3]);     /* :236-245 */
void ora_rotate_inv(const double q[4], const double v[3], double out[3]); /* :68 + :236 */

/* Clock semantics (Common/Common/Time/Timer.hpp:27-54,
 * ManualTimer.hpp:29-40, Quadcopter_T.cpp:87-91,159-160). */
typedef struct ora_clock {
  uint64_t now_us;            /* ManualTimer::_currentTime */
  uint64_t integ_reset_us;    /* _integrationTimer._lastResetTime_usec */
  uint64_t logic_reset_us;    /* _timerOnboardLogic._lastResetTime_usec */
  double logic_period;        /* _onboardLogicPeriod */
} ora_clock;
void ora_clock_init(ora_clock *c, double logic_period);
/* The timing part of one Run(): returns dt in seconds (0 => early return,
 * nothing else happens) and sets *tick when the logic gate fires. */
double ora_clock_run(ora_clock *c, int *tick);
void ora_clock_advance(ora_clock *c, uint64_t dt_us);

/* Batched SoA wrappers used by the tests / bench through ctypes.  Arrays are
 * planar: pos[3*n] = x[0..n) y[0..n) z[0..n) etc.  types[i] indexes table. */
/* threads ora_step_batch spreads the vehicles over (OpenMP); default 1 */
void ora_set_batch_threads(int n);
int ora_get_batch_threads(void);
void ora_step_batch(int64_t n, int n_steps, const ora_params *table,
                    const uint8_t *types, double *pos, double *vel,
                    double *att, double *ang_vel, double *motor_speed,
                    uint32_t *rng, const float *motor_cmd,
                    const double *ext_force, const double *ext_torque,
                    double dt, const uint8_t *tick_per_step, float *gyro,
                    float *acc);

#ifdef __cplusplus
}
#endif
#endif
