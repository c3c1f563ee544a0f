// agrifly/Quadcopter_T.hpp -- C++ host facade over the C ABI (agrifly_engine.h).
//
// Re-creates, for the vehicle-step path only, the two seams of agri-fly:
//   upper: the Vehicle API the loops hold -- Simulation::SimulationObject6DOF
//          (Components/Components/Simulation/SimulationObject6DOF.hpp:12-85)
//          and Simulation::Quadcopter_T<logicType> (Quadcopter_T.hpp:20-134):
//          same constructor arguments, Run(), Get/Set{Position,Velocity,
//          Attitude,AngularVelocity}, SetExternalForce/Torque, GetMotorForce,
//          SetCommandRadioMsg, GetTelemetryDataPackets, GetAccelerometer,
//          GetRateGyro, GetEstimate, AddUWBRadioTarget;
//   lower: the logicType concept -- the per-tick call order of
//          Quadcopter_T.cpp:163-189 is reproduced exactly.
// The physics between those seams runs on the GPU.  No exceptions escape in
// the reference (void functions, assert only); here a failing engine call
// aborts with the engine's message by default, or goes to the host's handler
// (agrifly::SetErrorHandler) and comes back as agrifly::last_status().
//
// Two shapes:
//   agrifly::Quadcopter_T<logicType>  one vehicle = one 1-vehicle ensemble; IS-A
//       Simulation::SimulationObject6DOF, so `std::shared_ptr<Simulation::SimulationObject6DOF>`
//       holders (AIFS_ROS/hiperlab_rostools/src/Simulator/main.cpp:83) take it unchanged;
//       behaves like the reference object (BaseTimer-driven dt, strict-> gate, state in the base's
//       _pos/_vel/_att/_angVel members, UWB radio hand-off).  It is a VIEW for drop-in use: every
//       Run() is a kernel launch plus a few small copies and synchronisations, orders of magnitude
//       slower than the CPU reference at N = 1 -- the batched Fleet / C ABI is the real interface.
//   agrifly::Fleet<logicType>         N vehicles in ONE ensemble, N host logic
//       objects: the `for (v : vehicles) v->Run();` of AIFS_ROS/.../Simulator/
//       main.cpp:323-325 as one launch + one batched IMU/command exchange.
#pragma once

#include <stdint.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../agrifly_engine.h"
#include "SimulationObject6DOF.hpp"   // the tree's own header under AGRIFLY_USE_REFERENCE_TYPES

namespace agrifly {

// Failure policy.  The reference's interface has nowhere to return an error to (void functions, assert only), so
// by default a failing engine call ends the process with the engine's message -- what its assert would do.  A host
// that wants the error instead installs a handler: it is called with the status, the failing call and the engine's
// message, and the facade call then returns to its caller having done nothing further (state unchanged; last_status()
// keeps the code).  The handler may also throw, which unwinds out of the facade call.
typedef void (*ErrorHandler)(int status, const char *what, const char *message);
inline ErrorHandler &error_handler() { static ErrorHandler h = 0; return h; }
inline int &last_status() { static int s = AFE_OK; return s; }
inline void SetErrorHandler(ErrorHandler h) { error_handler() = h; }

inline bool check(afe_engine *e, int rc, const char *what) {
  if (rc == AFE_OK) return true;
  last_status() = rc;
  const char *msg = e ? afe_last_error(e) : "";
  if (error_handler()) {
    error_handler()(rc, what, msg);
    return false;
  }
  std::fprintf(stderr, "agrifly: %s failed: %s (%s)\n", what, afe_status_string(rc), msg);
  std::abort();
}

// The ctor arguments of Quadcopter_T (Quadcopter_T.hpp:24-32) as a record.
template <class Matrix33>
inline afe_vehicle_params MakeParams(double mass, const Matrix33 &inertiaMatrix, double armLength,
                                     Vec3d centreOfMassError, double motorMinSpeed, double motorMaxSpeed,
                                     double propThrustFromSpeedSqr, double propTorqueFromSpeedSqr,
                                     double motorTimeConst, double motorInertia, Vec3d linDragCoeffB,
                                     float imuYaw = 0, float imuPitch = 0, float imuRoll = 0) {
  afe_vehicle_params p;
  p.mass = mass;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) p.inertia[3 * i + j] = inertiaMatrix(i, j);
  p.arm_length = armLength;
  for (int k = 0; k < 3; k++) { p.com_error[k] = centreOfMassError[k]; p.lin_drag_coeff_b[k] = linDragCoeffB[k]; }
  p.motor_min_speed = motorMinSpeed;
  p.motor_max_speed = motorMaxSpeed;
  p.prop_thrust_from_speed_sqr = propThrustFromSpeedSqr;
  p.prop_torque_from_speed_sqr = propTorqueFromSpeedSqr;
  p.motor_time_const = motorTimeConst;
  p.motor_inertia = motorInertia;
  p.imu_yaw = imuYaw; p.imu_pitch = imuPitch; p.imu_roll = imuRoll;
  return p;
}

struct Matrix33 {  // tiny stand-in where no Eigen is around: anything with (i,j) works
  double m[9];
  double operator()(int i, int j) const { return m[3 * i + j]; }
  double &operator()(int i, int j) { return m[3 * i + j]; }
};

// ---------------------------------------------------------------------------
template <class logicType>
class Fleet {
 public:
  // quadcopterType / ids are forwarded to logicType::Initialise exactly as
  // Quadcopter_T.cpp:82 does; `precision` AFE_F32 (production) or AFE_F64.
  Fleet(BaseTimer *const masterTimer, int64_t nVehicles, const std::vector<afe_vehicle_params> &typeTable,
        const std::vector<uint8_t> &typeIndex, double onboardLogicPeriod, int precision = AFE_F32,
        int device = -1, float battVoltage = 0.0f)
      : _master(masterTimer), _integrationTimer(masterTimer), _n(nVehicles), _period(onboardLogicPeriod),
        _battVoltage(battVoltage), _battCurrent(-1.0f) {  // Quadcopter_T.cpp:72-73
    // the onboard logic runs on the host at every tick (Run() below): up to a few thousand vehicles the state arena
    // lives in host-visible memory, so that a Run() and the accessors after it cost bus latency, not transfer calls
    if (nVehicles <= 16384) check(0, afe_create_host_visible(&_e, nVehicles, precision, device, 0), "afe_create_host_visible");
    else check(0, afe_create(&_e, nVehicles, precision, device, 0), "afe_create");
    check(_e, afe_set_type_table(_e, typeTable.data(), (int)typeTable.size()), "afe_set_type_table");
    if (!typeIndex.empty()) check(_e, afe_set_vehicle_types(_e, 0, nVehicles, typeIndex.data()), "afe_set_vehicle_types");
    check(_e, afe_set_logic_period(_e, onboardLogicPeriod), "afe_set_logic_period");
    _logic.reserve((size_t)nVehicles);
    for (int64_t i = 0; i < nVehicles; i++) _logic.emplace_back(masterTimer, float(onboardLogicPeriod));
    _cmd.assign((size_t)(4 * nVehicles), 0.0f);
    _gyro.assign((size_t)(3 * nVehicles), 0.0f);
    _acc.assign((size_t)(3 * nVehicles), 0.0f);
  }
  ~Fleet() { if (_e) afe_destroy(_e); }
  Fleet(const Fleet &) = delete;
  Fleet &operator=(const Fleet &) = delete;

  afe_engine *engine() { return _e; }
  int64_t size() const { return _n; }
  logicType &logic(int64_t i) { return _logic[(size_t)i]; }
  uint64_t ticks() { uint64_t t = 0; check(_e, afe_logic_ticks(_e, &t), "afe_logic_ticks"); return t; }

  // `for (v : vehicles) v->Run();` -- one Quadcopter_T::Run() per vehicle
  void Run() {
    const uint64_t dt_us = _integrationTimer.GetMicroSeconds();
    const double dt = (double)((double)dt_us * 1e-6);  // Timer.hpp:36-38
    if (dt < 1e-6) return;                              // Quadcopter_T.cpp:88-90
    _integrationTimer.Reset();
    uint64_t before = 0, after = 0;
    check(_e, afe_logic_ticks(_e, &before), "afe_logic_ticks");
    check(_e, afe_step(_e, dt_us, 1), "afe_step");
    check(_e, afe_logic_ticks(_e, &after), "afe_logic_ticks");
    if (after == before) return;                        // gate did not fire (:159)
    check(_e, afe_get_imu(_e, 0, _n, _gyro.data(), _acc.data()), "afe_get_imu");
    const float TEMP_MEAS = 25;                         // :182
    for (int64_t i = 0; i < _n; i++) {                  // :163-189, same order
      logicType &L = _logic[(size_t)i];
      L.SetBatteryMeasurement(_battVoltage, _battCurrent);
      L.SetIMUMeasurementRateGyro(_gyro[i], _gyro[_n + i], _gyro[2 * _n + i]);
      L.SetIMUMeasurementAccelerometer(_acc[i], _acc[_n + i], _acc[2 * _n + i]);
      L.SetIMUMeasurementTemperature(TEMP_MEAS);
      L.Run();
      for (unsigned m = 0; m < 4; m++) _cmd[(size_t)(m * _n + i)] = L.GetMotorSpeedCmd(m);
    }
    check(_e, afe_set_motor_cmds(_e, 0, _n, _cmd.data()), "afe_set_motor_cmds");
  }

  // per-vehicle accessors (SimulationObject6DOF.hpp:26-56)
  Vec3d GetPosition(int64_t i) { double v[3]; get(i, v, 0, 0, 0, 0); return Vec3d(v[0], v[1], v[2]); }
  Vec3d GetVelocity(int64_t i) { double v[3]; get(i, 0, v, 0, 0, 0); return Vec3d(v[0], v[1], v[2]); }
  Rotationd GetAttitude(int64_t i) { double q[4]; get(i, 0, 0, q, 0, 0); return Rotationd(q[0], q[1], q[2], q[3]); }
  Vec3d GetAngularVelocity(int64_t i) { double v[3]; get(i, 0, 0, 0, v, 0); return Vec3d(v[0], v[1], v[2]); }
  void SetPosition(int64_t i, Vec3d in) { double v[3] = {in.x, in.y, in.z}; set(i, v, 0, 0, 0, 0); }
  void SetVelocity(int64_t i, Vec3d in) { double v[3] = {in.x, in.y, in.z}; set(i, 0, v, 0, 0, 0); }
  void SetAttitude(int64_t i, Rotationd in) { double q[4] = {in[0], in[1], in[2], in[3]}; set(i, 0, 0, q, 0, 0); }
  void SetAngularVelocity(int64_t i, Vec3d in) { double v[3] = {in.x, in.y, in.z}; set(i, 0, 0, 0, v, 0); }
  void SetExternalForce(int64_t i, Vec3d in) {  // Quadcopter_T.hpp:45
    double v[3] = {in.x, in.y, in.z};
    check(_e, afe_set_external_force(_e, i, 1, v), "afe_set_external_force");
  }
  void SetExternalTorque(int64_t i, Vec3d in) {  // Quadcopter_T.hpp:49
    double v[3] = {in.x, in.y, in.z};
    check(_e, afe_set_external_torque(_e, i, 1, v), "afe_set_external_torque");
  }
  double GetMotorSpeed(int64_t i, unsigned m) { double w[4]; get(i, 0, 0, 0, 0, w); return w[m]; }

 private:
  void get(int64_t i, double *p, double *v, double *q, double *w, double *m) {
    check(_e, afe_get_state(_e, i, 1, p, v, q, w, m), "afe_get_state");
  }
  void set(int64_t i, const double *p, const double *v, const double *q, const double *w, const double *m) {
    check(_e, afe_set_state(_e, i, 1, p, v, q, w, m), "afe_set_state");
  }

  afe_engine *_e = 0;
  BaseTimer *const _master;
  Timer _integrationTimer;
  int64_t _n;
  double _period;
  float _battVoltage, _battCurrent;
  std::vector<logicType> _logic;
  std::vector<float> _cmd, _gyro, _acc;
};

// ---------------------------------------------------------------------------
// Drop-in for Simulation::Quadcopter_T<logicType> (one vehicle).
template <class logicType>
class Quadcopter_T : public Simulation::SimulationObject6DOF {
 public:
  // Same argument list as Quadcopter_T.hpp:24-32.  inertiaMatrix: anything with
  // operator()(i,j) (Eigen::Matrix<double,3,3> in the agri-fly tree).
  // quadcopterType is passed through to logicType::Initialise as an int-like
  // value; lowBatteryThreshold / IMU mount angles come with the type in the
  // reference (Quadcopter_T.cpp:71-77) and are explicit optional arguments here.
  template <class Matrix33T, class QuadcopterTypeT>
  Quadcopter_T(BaseTimer *const masterTimer, double mass, const Matrix33T &inertiaMatrix, double armLength,
               Vec3d centreOfMassError, double motorMinSpeed, double motorMaxSpeed,
               double propThrustFromSpeedSqr, double propTorqueFromSpeedSqr, double motorTimeConst,
               double motorInertia, Vec3d linDragCoeffB, uint8_t id, QuadcopterTypeT quadcopterType,
               double onboardLogicPeriod, int precision = AFE_F32, float lowBatteryThreshold = 0.0f,
               float imuYaw = 0, float imuPitch = 0, float imuRoll = 0)
      : Simulation::SimulationObject6DOF(masterTimer),
        _fleet(masterTimer, 1,
               std::vector<afe_vehicle_params>(1, MakeParams(mass, inertiaMatrix, armLength, centreOfMassError,
                                                             motorMinSpeed, motorMaxSpeed, propThrustFromSpeedSqr,
                                                             propTorqueFromSpeedSqr, motorTimeConst, motorInertia,
                                                             linDragCoeffB, imuYaw, imuPitch, imuRoll)),
               std::vector<uint8_t>(), onboardLogicPeriod, precision, -1,
               1.2f * lowBatteryThreshold),  // Quadcopter_T.cpp:72
        _kThrust(propThrustFromSpeedSqr) {
    _radio.reset(new Simulation::UWBRadio(masterTimer, id));   // Quadcopter_T.cpp:68
    _fleet.logic(0).Initialise(quadcopterType, id);              // Quadcopter_T.cpp:82
    remember();
  }
  virtual ~Quadcopter_T() {}

  // The reference's Run() works on the base's _pos/_vel/_att/_angVel members (set through the base's
  // non-virtual setters) and writes them back (Quadcopter_T.cpp:140-156).  Here: members that changed
  // since the last read-back go to the device, the step runs there, the new state comes back.
  virtual void Run() {
    push();
    const uint64_t before = _fleet.ticks();
    _fleet.Run();
    pull();
    if (_fleet.ticks() != before && _radio) {   // the UWB hand-off of a logic tick, Quadcopter_T.cpp:191-199
      _radio->SetPosition(_pos);
      _radio->SetNextRangingTarget(_fleet.logic(0).GetNextUWBRangingTarget());
      if (_radio->GetHaveNewMeasurement()) {
        Simulation::UWBRadio::RangingMeasurement meas = _radio->GetMeasurement();
        _fleet.logic(0).SetUWBMeasurement(meas.range, meas.responderId, meas.failure);
      }
    }
  }

  // Quadcopter_T.hpp:39-42: z component of motor i's thrust = k_f w |w|
  double GetMotorForce(unsigned i) {
    const double w = _fleet.GetMotorSpeed(0, i);
    return _kThrust * w * (w < 0 ? -w : w);
  }
  void SetExternalForce(Vec3d in) { _fleet.SetExternalForce(0, in); }
  void SetExternalTorque(Vec3d in) { _fleet.SetExternalTorque(0, in); }

  // pass-throughs to the onboard logic (Quadcopter_T.hpp:53-83)
  void GetEstimate(Vec3f &pos, Vec3f &vel, Rotationf &att, Vec3f &angVel) { _fleet.logic(0).GetEstimate(pos, vel, att, angVel); }
  virtual void AddUWBRadioTarget(uint8_t id, Vec3f pos) { _fleet.logic(0).AddRangingTargetId(id, pos); }
  virtual void SetCommandRadioMsg(RadioTypes::RadioMessageDecoded::RawMessage const raw) {
#ifdef AGRIFLY_USE_REFERENCE_TYPES
    RadioTypes::RadioMessageDecoded msg = RadioTypes::RadioMessageDecoded(raw.raw);  // Quadcopter_T.hpp:64-66
    _fleet.logic(0).SetRadioMessage(msg);
#else
    _fleet.logic(0).SetRadioMessage(raw);
#endif
  }
  virtual void GetTelemetryDataPackets(TelemetryPacket::data_packet_t &dataPacket1,
                                       TelemetryPacket::data_packet_t &dataPacket2) {
    _fleet.logic(0).GetTelemetryDataPackets(dataPacket1, dataPacket2);
  }
  virtual void GetAccelerometer(Vec3d &acc) { acc = Vec3d(_fleet.logic(0).GetAccelerometer()); }
  virtual void GetRateGyro(Vec3d &rateGyro) { rateGyro = Vec3d(_fleet.logic(0).GetRateGyro()); }

  logicType &Logic() { return _fleet.logic(0); }
  afe_engine *engine() { return _fleet.engine(); }

 private:
  static bool same(const double *a, const double *b, int n) { return std::memcmp(a, b, sizeof(double) * (size_t)n) == 0; }
  void members(double p[3], double v[3], double q[4], double w[3]) const {
    for (int k = 0; k < 3; k++) { p[k] = _pos[k]; v[k] = _vel[k]; w[k] = _angVel[k]; }
    for (unsigned k = 0; k < 4; k++) q[k] = _att[k];
  }
  void remember() { members(_sp, _sv, _sq, _sw); }
  void push() {   // only what the caller changed through the base's setters
    double p[3], v[3], q[4], w[3];
    members(p, v, q, w);
    const bool dp = !same(p, _sp, 3), dv = !same(v, _sv, 3), dq = !same(q, _sq, 4), dw = !same(w, _sw, 3);
    if (dp || dv || dq || dw)
      check(_fleet.engine(), afe_set_state(_fleet.engine(), 0, 1, dp ? p : 0, dv ? v : 0, dq ? q : 0, dw ? w : 0, 0), "afe_set_state");
  }
  void pull() {
    double p[3], v[3], q[4], w[3];
    check(_fleet.engine(), afe_get_state(_fleet.engine(), 0, 1, p, v, q, w, 0), "afe_get_state");
    _pos = Vec3d(p[0], p[1], p[2]);
    _vel = Vec3d(v[0], v[1], v[2]);
    _att = Rotationd(q[0], q[1], q[2], q[3]);
    _angVel = Vec3d(w[0], w[1], w[2]);
    remember();
  }

  Fleet<logicType> _fleet;
  double _kThrust;
  double _sp[3], _sv[3], _sq[4], _sw[3];   // the state as of the last read-back
};

}  // namespace agrifly
