// tests/cpp/test_facade.cpp -- drives the C++ facade the way the reference's
// loops drive Simulation::Quadcopter: a recording logicType (every method the
// duck-typed concept needs, Quadcopter_T.cpp:163-199), BaseTimer-driven dt,
// fixed motor commands.  Prints a JSON trace that tests/test_gpu_facade.py
// compares with the oracle run through the same scenario.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

#include "agrifly/Quadcopter_T.hpp"

struct TapLogic {
  float cmd[4];
  Vec3f gyro, acc;
  int runs, order_ok, stage;
  TapLogic(BaseTimer *, float) : gyro(0, 0, 0), acc(0, 0, 0), runs(0), order_ok(1), stage(0) {
    for (int i = 0; i < 4; i++) cmd[i] = 0;
  }
  void Initialise(int, uint8_t) {}
  void GetEstimate(Vec3f &, Vec3f &, Rotationf &, Vec3f &) const {}
  int AddRangingTargetId(uint8_t, Vec3f) { return 0; }
  template <class M> void SetRadioMessage(M const) {}
  template <class P> void GetTelemetryDataPackets(P &, P &) {}
  Vec3f GetAccelerometer() const { return acc; }
  Vec3f GetRateGyro() const { return gyro; }
  // the reference calls these in exactly this order every tick
  void SetBatteryMeasurement(float, float) { order_ok &= (stage == 0); stage = 1; }
  void SetIMUMeasurementRateGyro(float x, float y, float z) { order_ok &= (stage == 1); stage = 2; gyro = Vec3f(x, y, z); }
  void SetIMUMeasurementAccelerometer(float x, float y, float z) { order_ok &= (stage == 2); stage = 3; acc = Vec3f(x, y, z); }
  void SetIMUMeasurementTemperature(float t) { order_ok &= (stage == 3 && t == 25.0f); stage = 4; }
  void Run() { order_ok &= (stage == 4); stage = 0; runs++; }
  float GetMotorSpeedCmd(unsigned i) const { return cmd[i]; }
  uint8_t GetNextUWBRangingTarget() const { return 0; }
  void SetUWBMeasurement(float, uint8_t, bool) {}
};

int main(int argc, char **argv) {
  const int precision = (argc > 1 && !std::strcmp(argv[1], "f64")) ? AFE_F64 : AFE_F32;
  const uint64_t dt_us = argc > 2 ? (uint64_t)atoll(argv[2]) : 1000;
  const double period = argc > 3 ? atof(argv[3]) : 0.0005;
  const int runs = argc > 4 ? atoi(argv[4]) : 4;

  afe_vehicle_params c;
  if (afe_params_from_type(afe_type_from_id(1), &c) != AFE_OK) return 2;  // id 1 -> MINIQUAD (main.cpp:147-150)
  agrifly::Matrix33 I;
  for (int k = 0; k < 9; k++) I.m[k] = c.inertia[k];
  ManualTimer simTimer;
  agrifly::Quadcopter_T<TapLogic> quad(&simTimer, c.mass, I, c.arm_length, Vec3d(0, 0, 0), c.motor_min_speed,
                                       c.motor_max_speed, c.prop_thrust_from_speed_sqr,
                                       c.prop_torque_from_speed_sqr, c.motor_time_const, c.motor_inertia,
                                       Vec3d(0.1, 0.1, 0.1), 1, 5, period, precision);
  quad.SetPosition(Vec3d(0, 0, 1));
  quad.SetAttitude(Rotationd::FromEulerYPR(0.3, 0.1, -0.2));
  quad.SetVelocity(Vec3d(1, -2, 0.5));
  quad.SetAngularVelocity(Vec3d(0.5, -0.4, 0.3));
  const double wh = std::sqrt(c.mass * 9.81 / (4 * c.prop_thrust_from_speed_sqr));
  quad.Logic().cmd[0] = float(wh * 1.02);
  quad.Logic().cmd[1] = float(wh * 0.99);
  quad.Logic().cmd[2] = float(wh * 1.01);
  quad.Logic().cmd[3] = float(wh * 0.98);

  std::printf("{\"precision\": %d, \"dt_us\": %llu, \"period\": %.17g, \"trace\": [\n", precision,
              (unsigned long long)dt_us, period);
  for (int s = 0; s < runs; s++) {
    quad.Run();  // first call: dt == 0, nothing happens (Quadcopter_T.cpp:88-90)
    simTimer.AdvanceMicroSeconds(dt_us);
    Vec3d p = quad.GetPosition(), v = quad.GetVelocity(), w = quad.GetAngularVelocity();
    Rotationd a = quad.GetAttitude();
    const TapLogic &L = quad.Logic();
    std::printf(" {\"pos\": [%.17g, %.17g, %.17g], \"vel\": [%.17g, %.17g, %.17g], "
                "\"att\": [%.17g, %.17g, %.17g, %.17g], \"ang_vel\": [%.17g, %.17g, %.17g], "
                "\"gyro\": [%.9g, %.9g, %.9g], \"acc\": [%.9g, %.9g, %.9g], \"runs\": %d, "
                "\"order_ok\": %d, \"motor_force0\": %.17g}%s\n",
                p.x, p.y, p.z, v.x, v.y, v.z, a[0], a[1], a[2], a[3], w.x, w.y, w.z, L.gyro.x, L.gyro.y,
                L.gyro.z, L.acc.x, L.acc.y, L.acc.z, L.runs, L.order_ok, quad.GetMotorForce(0),
                s + 1 < runs ? "," : "");
  }
  std::printf("]}\n");
  return 0;
}
