/*
 * agrifly_oracle_logic.h -- CPU restatement of the onboard rates-control path
 * (SURVEY.md 8f row f1): IMU low-pass -> angular-velocity estimate ->
 * angular-velocity controller -> mixer -> propeller speeds, i.e. what
 * Onboard::QuadcopterLogic does per tick in FS_EXTERNAL_RATES_CONTROL.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as agrifly_oracle.h).
 * PARITY STATUS: the second-order low-pass is pinned against the reference's
 * own Common/Common/Math/LowPassFilterSecondOrder.hpp (stand-alone header,
 * oracle/_ref/lpf_probe, tests/golden/lpf_kat.json); controller, mixer and the
 * call sequence are **parity unpinned** restatements (their headers need Eigen).
 */
#ifndef AGRIFLY_ORACLE_LOGIC_H
#define AGRIFLY_ORACLE_LOGIC_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ora_lpf2 {      /* LowPassFilterSecondOrder<float, float> */
  float a1, a2, b0, b1, b2;
  float xm0, xm1, ym0, ym1;
} ora_lpf2;
void ora_lpf2_init(ora_lpf2 *f, float sampling_period, float cutoff_rad_s, float init_value);
float ora_lpf2_apply(ora_lpf2 *f, float input);

typedef struct ora_logic_params { /* QuadcopterLogic::Initialise, QuadcopterLogic.cpp:97-150 */
  float mass;            /* _mass */
  float inertia[9];      /* consts.inertiaMatrix (float), QuadcopterConstants.hpp:269-271 */
  float tc_xy, tc_z;     /* angVelControl_timeConst_xy / _z */
  float d, kt, kf;       /* QuadcopterMixer::SetParameters, QuadcopterMixer.hpp:36-52 */
  float max_thrust, min_thrust, max_cmd_total_thrust;
  float R[9];            /* _R = FromEulerYPR(IMU ypr).GetRotationMatrix() */
  float onboard_period;  /* float(onboardLogicPeriod), Quadcopter_T.cpp:18 */
  float gyro_cutoff;     /* 200 rad/s, QuadcopterLogic.cpp:103 */
} ora_logic_params;

typedef struct ora_logic_state {
  ora_lpf2 gyro_lpf[3];      /* _imuRateGyro.lowPass (Vec3f sample = 3 scalar filters) */
  float ang_vel_est[3];      /* KalmanFilter6DOF::_angVel */
  int imu_initialized;       /* KalmanFilter6DOF::_IMUInitialized */
  int have_rates_cmd;        /* _state == FS_EXTERNAL_RATES_CONTROL */
  float thrust_norm;         /* _radioMessage.msg.floats[0] */
  float des_ang_vel[3];      /* floats[1..3] */
  float motor_speed_cmd[4];  /* _desMotorSpeeds */
  float motor_force_cmd[4];  /* _desMotorForcesForTelemetry */
} ora_logic_state;

int ora_logic_params_from_type(ora_logic_params *p, int quadcopter_type, float onboard_period);
void ora_logic_init(const ora_logic_params *p, ora_logic_state *s);
void ora_logic_set_rates_cmd(ora_logic_state *s, float thrust_norm, const float des_ang_vel[3]);
/* one tick: SetIMUMeasurementRateGyro(gyro) ... Run() in rates mode */
void ora_logic_tick(const ora_logic_params *p, ora_logic_state *s, const float gyro[3]);

#ifdef __cplusplus
}
#endif
#endif
