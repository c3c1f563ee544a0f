/*
 * agrifly_oracle_logic.c -- see agrifly_oracle_logic.h.  TEST INFRASTRUCTURE.
 * float arithmetic throughout, as in the reference's onboard code.
 * Citations: file:line under /root/reference.
 */
#include "agrifly_oracle_logic.h"

#include <math.h>
#include <string.h>

/* ---- Common/Common/Math/LowPassFilterSecondOrder.hpp -------------------- */
void ora_lpf2_init(ora_lpf2 *f, float samplingPeriod, float wc, float initValue) {
  /* :22-49, TYPE_RATE = float */
  const float dt = samplingPeriod;
  const float sqrt2 = (float)sqrt(2.0);
  f->a1 = (dt * dt * wc * wc - 2 * sqrt2 * dt * wc + 4) / (dt * dt * wc * wc + 2 * sqrt2 * dt * wc + 4);
  f->a2 = 2 * (dt * dt * wc * wc - 4) / (dt * dt * wc * wc + 2 * sqrt2 * dt * wc + 4);
  f->b0 = dt * dt * wc * wc / (dt * dt * wc * wc + 2 * sqrt2 * dt * wc + 4);
  f->b1 = dt * dt * wc * wc / (dt * dt * wc * wc + 2 * sqrt2 * dt * wc + 4);
  f->b2 = 2 * dt * dt * wc * wc / (dt * dt * wc * wc + 2 * sqrt2 * dt * wc + 4);
  f->xm0 = f->xm1 = f->ym0 = f->ym1 = initValue;
}

float ora_lpf2_apply(ora_lpf2 *f, float input) {
  /* :51-64, literal (the doubled coefficient multiplies the current input) */
  float output = f->b2 * input;
  output += +f->b0 * f->xm0 + f->b1 * f->xm1;
  output += -f->a1 * f->ym0 - f->a2 * f->ym1;
  f->xm0 = f->xm1;
  f->xm1 = input;
  f->ym0 = f->ym1;
  f->ym1 = output;
  return output;
}

/* ---- constants ---------------------------------------------------------- */
static float max_cf_speed(float c[3][2]) { /* QuadcopterConstants.hpp:370-392 */
  int MAX_PWM = 255;
  float MAX_BATT = 4.1;
  float k_1 = c[0][0] + c[0][1] * MAX_BATT, k_2 = c[1][0] + c[1][1] * MAX_BATT, k_3 = c[2][0] + c[2][1] * MAX_BATT;
  return (-k_2 + sqrtf(powf(k_2, 2) - 4 * k_3 * (k_1 - MAX_PWM))) / (2 * k_3);
}

int ora_logic_params_from_type(ora_logic_params *p, int type, float onboard_period) {
  /* QuadcopterConstants.hpp:31-274 + QuadcopterLogic.cpp:97-150 */
  float ixx, izz, arm, kf, ktt, max_speed, max_thrust, min_thrust = 0.0f, max_cmd = -1;
  float tc_xy = 0.03f, tc_z = 0.5f; /* defaults :36-39 */
  memset(p, 0, sizeof(*p));
  switch (type) {
    case 1: {
      float c[3][2] = {{-86.19993685f, 22.87189816f}, {0.30208677f, -0.07345602f}, {-1.59346434e-05f, 1.53209239e-05f}};
      p->mass = 38e-3; ixx = 16e-6f; izz = 29e-6f; arm = 46e-3f; kf = (float)3.58e-8f; ktt = 0.0006;
      max_speed = max_cf_speed(c);
      max_thrust = kf * powf(max_speed, 2);
      max_cmd = 0.9f * max_thrust * 4;
      tc_xy = 0.04f;
      break;
    }
    case 2: {
      float c[3][2] = {{-379.31113434f, 84.84738207f}, {0.65309704f, -0.13852527f}, {-1.34462353e-04f, 3.57662798e-05f}};
      p->mass = 39e-3; ixx = 30e-6f; izz = 60e-6f; arm = 48e-3f; kf = (float)4.14e-8f; ktt = 0.001;
      max_speed = max_cf_speed(c);
      max_thrust = kf * powf(max_speed, 2);
      max_cmd = 0.8f * max_thrust * 4;
      break;
    }
    case 4:
      p->mass = 0.760; ixx = 0.004406f; izz = 0.008611f; arm = 0.166f; kf = 7.64e-6f; ktt = 0.0140f;
      max_speed = (2000 - 972.0f) / 0.742f;
      max_thrust = kf * powf(max_speed, 2);
      tc_xy = 0.0457f; tc_z = 0.2545f;
      break;
    case 5:
      p->mass = 0.142; ixx = 92.7e-6f; izz = 158.57e-6f; arm = 58e-3f; kf = 4.32e-8f; ktt = 0.00808f;
      max_speed = (2000 - 999.0f) / 0.14f;
      max_thrust = kf * powf(max_speed, 2);
      min_thrust = 0.03f;
      max_cmd = 0.7f * (max_thrust * 4);
      tc_xy = 0.04f; tc_z = tc_xy * 5;
      break;
    default:
      return -1;
  }
  p->inertia[0] = ixx; p->inertia[4] = ixx; p->inertia[8] = izz;
  p->tc_xy = tc_xy; p->tc_z = tc_z;
  /* QuadcopterMixer::SetParameters, QuadcopterMixer.hpp:36-52; prop0SpinDir = 1 */
  p->d = arm / sqrtf(2.0f);
  p->kt = 1 * ktt;
  p->kf = kf;
  p->max_thrust = max_thrust;
  p->min_thrust = min_thrust;
  p->max_cmd_total_thrust = (max_cmd < 0) ? 4 * max_thrust * 0.8f : max_cmd;
  /* _R: IMU angles are zero for every shipped type => identity */
  p->R[0] = p->R[4] = p->R[8] = 1.0f;
  p->onboard_period = onboard_period;
  p->gyro_cutoff = 200.0f;
  return 0;
}

void ora_logic_init(const ora_logic_params *p, ora_logic_state *s) {
  memset(s, 0, sizeof(*s));
  for (int k = 0; k < 3; k++) ora_lpf2_init(&s->gyro_lpf[k], p->onboard_period, p->gyro_cutoff, 0.0f); /* :38,133 */
}

void ora_logic_set_rates_cmd(ora_logic_state *s, float thrust_norm, const float w[3]) {
  /* SetRadioMessage + ParseIncomingCommunications (QuadcopterLogic.cpp:275-303) for externalRatesCmd */
  s->have_rates_cmd = 1;
  s->thrust_norm = thrust_norm;
  s->des_ang_vel[0] = w[0]; s->des_ang_vel[1] = w[1]; s->des_ang_vel[2] = w[2];
}

void ora_logic_tick(const ora_logic_params *p, ora_logic_state *s, const float gyro[3]) {
  /* SetIMUMeasurementRateGyro, QuadcopterLogic.hpp:40-45 (bias = 0, :145) */
  float raw[3];
  for (int i = 0; i < 3; i++) {
    float acc = 0.0f;
    for (int j = 0; j < 3; j++) acc += p->R[3 * i + j] * gyro[j];
    raw[i] = acc;
  }
  float filt[3];
  for (int k = 0; k < 3; k++) filt[k] = ora_lpf2_apply(&s->gyro_lpf[k], raw[k] - 0.0f);

  /* Run() -> UpdateEstimator -> KalmanFilter6DOF::Predict (KalmanFilter6DOF.cpp:70-147):
   * the first call only initialises; afterwards _angVel = measGyro (:115) */
  if (!s->imu_initialized) {
    s->imu_initialized = 1;
  } else {
    s->ang_vel_est[0] = filt[0]; s->ang_vel_est[1] = filt[1]; s->ang_vel_est[2] = filt[2];
  }

  if (!s->have_rates_cmd) { /* FS_IDLE: QuadcopterLogic.cpp:213-217 */
    for (int i = 0; i < 4; i++) { s->motor_speed_cmd[i] = 0; s->motor_force_cmd[i] = 0; }
    return;
  }

  /* RunControllerExternalRatesControl, QuadcopterLogic.cpp:528-541 */
  const float *w = s->ang_vel_est;
  /* GetDesiredTorques, QuadcopterAngularVelocityController.hpp:25-38 */
  const float ex = s->des_ang_vel[0] - w[0], ey = s->des_ang_vel[1] - w[1], ez = s->des_ang_vel[2] - w[2];
  const float aa[3] = {ex / p->tc_xy, ey / p->tc_xy, ez / p->tc_z};
  float Iw[3], Ia[3];
  for (int i = 0; i < 3; i++) {
    float a = 0.0f, b = 0.0f;
    for (int j = 0; j < 3; j++) { a += p->inertia[3 * i + j] * w[j]; b += p->inertia[3 * i + j] * aa[j]; }
    Iw[i] = a; Ia[i] = b;
  }
  const float nl[3] = {w[1] * Iw[2] - w[2] * Iw[1], w[2] * Iw[0] - w[0] * Iw[2], w[0] * Iw[1] - w[1] * Iw[0]};
  const float t[3] = {Ia[0] + nl[0], Ia[1] + nl[1], Ia[2] + nl[2]};

  /* QuadcopterMixer::GetMotorForces, QuadcopterMixer.hpp:63-86 */
  const float totF = s->thrust_norm * p->mass;
  const float desF = totF > p->max_cmd_total_thrust ? p->max_cmd_total_thrust : totF;
  float *F = s->motor_force_cmd;
  F[0] = (-t[0] / p->d - t[1] / p->d - t[2] / p->kt + desF) / 4.0f;
  F[1] = (-t[0] / p->d + t[1] / p->d + t[2] / p->kt + desF) / 4.0f;
  F[2] = (+t[0] / p->d + t[1] / p->d - t[2] / p->kt + desF) / 4.0f;
  F[3] = (+t[0] / p->d - t[1] / p->d + t[2] / p->kt + desF) / 4.0f;
  for (int i = 0; i < 4; i++) {
    if (F[i] < p->min_thrust) F[i] = p->min_thrust;
    else if (F[i] > p->max_thrust) F[i] = p->max_thrust;
  }
  /* PropellerSpeedsFromThrust, QuadcopterMixer.hpp:88-99 (correction factors 1) */
  for (int i = 0; i < 4; i++) {
    if (F[i] <= 0) { s->motor_speed_cmd[i] = 0; continue; }
    s->motor_speed_cmd[i] = sqrtf(F[i] / (1.0f * p->kf));
  }
}
