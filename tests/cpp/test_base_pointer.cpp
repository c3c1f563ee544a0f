// tests/cpp/test_base_pointer.cpp -- the facade held the way the reference's loops hold a vehicle:
//   std::shared_ptr<Simulation::SimulationObject6DOF> vehicle     (AIFS_ROS/.../Simulator/main.cpp:83)
// Everything goes through the base class: the non-virtual state setters / getters that work on the
// base's members, the virtual Run / SetCommandRadioMsg / GetTelemetryDataPackets / GetAccelerometer /
// GetRateGyro / AddUWBRadioTarget / GetRadio, and a ranging exchange between two vehicles' radios
// (the radios are what a ranging network -- the tree's Simulation::UWBNetwork -- works on).  Prints a JSON trace; tests/test_gpu_facade.py checks it against the direct
// (derived-class) trace and the oracle.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "agrifly/Quadcopter_T.hpp"

struct TapLogic {
  float cmd[4];
  Vec3f gyro, acc;
  int runs, n_radio, n_uwb;
  uint8_t last_radio_type, target, my_id, uwb_responder;
  float uwb_range;
  TapLogic(BaseTimer *, float) : gyro(0, 0, 0), acc(0, 0, 0), runs(0), n_radio(0), n_uwb(0), last_radio_type(0), target(0),
                                 my_id(0), uwb_responder(0), uwb_range(0) {
    for (int i = 0; i < 4; i++) cmd[i] = 0;
  }
  void Initialise(int, uint8_t id) { my_id = id; }
  void GetEstimate(Vec3f &, Vec3f &, Rotationf &, Vec3f &) const {}
  int AddRangingTargetId(uint8_t id, Vec3f) { target = id; return 0; }
  void SetRadioMessage(RadioTypes::RadioMessageDecoded::RawMessage const m) { n_radio++; last_radio_type = m.raw[0]; }
  void GetTelemetryDataPackets(TelemetryPacket::data_packet_t &a, TelemetryPacket::data_packet_t &b) {
    std::memset(&a, 0, sizeof(a));
    std::memset(&b, 0, sizeof(b));
    a.type = 0; b.type = 1;
    a.packetNumber = b.packetNumber = (uint8_t)runs;
  }
  Vec3f GetAccelerometer() const { return acc; }
  Vec3f GetRateGyro() const { return gyro; }
  void SetBatteryMeasurement(float, float) {}
  void SetIMUMeasurementRateGyro(float x, float y, float z) { gyro = Vec3f(x, y, z); }
  void SetIMUMeasurementAccelerometer(float x, float y, float z) { acc = Vec3f(x, y, z); }
  void SetIMUMeasurementTemperature(float) {}
  void Run() { runs++; }
  float GetMotorSpeedCmd(unsigned i) const { return cmd[i]; }
  uint8_t GetNextUWBRangingTarget() const { return target; }
  void SetUWBMeasurement(float range, uint8_t responder, bool) { n_uwb++; uwb_range = range; uwb_responder = responder; }
};

typedef agrifly::Quadcopter_T<TapLogic> Quad;

// Test driver for the radios the facade hands out: a ranging service with the timing of a two-phase exchange.
// Every `period` seconds it does ONE thing -- either it latches the first radio that asks for a range, or it
// answers the latched request: true distance between the two radios plus one draw of the engine's UWB noise
// stream (afe_uwb_draw), broadcast to every radio.  Inside the agri-fly tree the tree's own network class drives
// the same radios; this is only what the test needs to see the hand-off work.
struct RangingService {
  typedef std::shared_ptr<Simulation::UWBRadio> Radio;
  std::vector<Radio> radios;
  Timer since;
  double period;
  afe_uwb_network *noise;
  int from, to;      // latched request (radio ids), 0 = none
  RangingService(BaseTimer *t, double period_s, double sigma, double p_outlier, double sigma_outlier)
      : since(t), period(period_s), noise(0), from(0), to(0) {
    afe_uwb_create(&noise);
    afe_uwb_set_noise(noise, sigma, p_outlier, sigma_outlier);
  }
  ~RangingService() { afe_uwb_destroy(noise); }
  Radio find(int id) const {
    for (size_t k = 0; k < radios.size(); k++) if (radios[k]->GetId() == id) return radios[k];
    return Radio();
  }
  void Poll() {
    for (size_t k = 0; k < radios.size(); k++) radios[k]->Run();
    if (since.GetSeconds<double>() < period) return;
    if (from && to) {
      const Radio a = find(from), b = find(to);
      double n = 0;
      uint8_t outlier = 0;
      afe_uwb_draw(noise, 1, &n, &outlier);
      Simulation::UWBRadio::RangingMeasurement m;
      m.range = outlier ? (float)n : (float)((a->GetPosition() - b->GetPosition()).GetNorm2() + n);
      m.haveNew = true; m.failure = false; m.responderId = (uint8_t)to;
      for (size_t k = 0; k < radios.size(); k++) radios[k]->SetMeasurement(m);
      from = to = 0;
      return;
    }
    for (size_t k = 0; k < radios.size() && !from; k++)
      if (radios[k]->GetNextRangingTargetId()) { from = radios[k]->GetId(); to = radios[k]->GetNextRangingTargetId(); }
    since.Reset();
  }
};

// a radio's position is a default-constructed (NaN) Vec3d until the first logic tick hands it the vehicle's
static std::string num(double v) {
  char buf[40];
  if (!std::isfinite(v)) return "null";
  std::snprintf(buf, sizeof(buf), "%.17g", v);
  return buf;
}

static std::shared_ptr<Quad> make(BaseTimer *t, uint8_t id, int precision, double period) {
  afe_vehicle_params c;
  afe_params_from_type(afe_type_from_id(1), &c);
  agrifly::Matrix33 I;
  for (int k = 0; k < 9; k++) I.m[k] = c.inertia[k];
  return std::make_shared<Quad>(t, c.mass, I, c.arm_length, Vec3d(0, 0, 0), c.motor_min_speed, c.motor_max_speed,
                                c.prop_thrust_from_speed_sqr, c.prop_torque_from_speed_sqr, c.motor_time_const,
                                c.motor_inertia, Vec3d(0.1, 0.1, 0.1), id, 5, period, precision);
}

static int g_handled = 0, g_last = 0;
static void on_error(int status, const char *, const char *) { g_handled++; g_last = status; }

int main(int argc, char **argv) {
  // a failing engine call reaches the host's handler instead of ending the process: a vehicle that cannot exist
  // (negative mass -> AFE_ERR_INVALID_ARG from afe_set_type_table, and every later call on it is refused)
  agrifly::SetErrorHandler(on_error);
  {
    ManualTimer t0;
    afe_vehicle_params c;
    afe_params_from_type(5, &c);
    agrifly::Matrix33 I;
    for (int k = 0; k < 9; k++) I.m[k] = c.inertia[k];
    Quad bad(&t0, -1.0, I, c.arm_length, Vec3d(0, 0, 0), c.motor_min_speed, c.motor_max_speed, c.prop_thrust_from_speed_sqr,
             c.prop_torque_from_speed_sqr, c.motor_time_const, c.motor_inertia, Vec3d(0, 0, 0), 1, 5, 0.002, AFE_F32);
    t0.AdvanceMicroSeconds(1000);
    bad.Run();
  }
  const int handled = g_handled, handled_status = g_last;
  agrifly::SetErrorHandler(0);
  const int precision = (argc > 1 && !std::strcmp(argv[1], "f64")) ? AFE_F64 : AFE_F32;
  const uint64_t dt_us = argc > 2 ? (uint64_t)atoll(argv[2]) : 1000;
  const double period = argc > 3 ? atof(argv[3]) : 0.0005;
  const int runs = argc > 4 ? atoi(argv[4]) : 4;

  ManualTimer simTimer;
  std::shared_ptr<Quad> q1 = make(&simTimer, 1, precision, period), q2 = make(&simTimer, 2, precision, period);
  // what the loops hold
  std::vector<std::shared_ptr<Simulation::SimulationObject6DOF> > vehicles;
  vehicles.push_back(q1);
  vehicles.push_back(q2);
  afe_vehicle_params c;
  afe_params_from_type(5, &c);
  const double wh = std::sqrt(c.mass * 9.81 / (4 * c.prop_thrust_from_speed_sqr));
  for (size_t k = 0; k < vehicles.size(); k++) {
    Simulation::SimulationObject6DOF &v = *vehicles[k];
    v.SetPosition(Vec3d(0 + 3.0 * k, 0 - 1.0 * k, 1 + 0.5 * k));
    v.SetAttitude(Rotationd::FromEulerYPR(0.3, 0.1, -0.2));
    v.SetVelocity(Vec3d(1, -2, 0.5));
    v.SetAngularVelocity(Vec3d(0.5, -0.4, 0.3));
    Quad &q = k == 0 ? *q1 : *q2;
    q.Logic().cmd[0] = float(wh * 1.02); q.Logic().cmd[1] = float(wh * 0.99);
    q.Logic().cmd[2] = float(wh * 1.01); q.Logic().cmd[3] = float(wh * 0.98);
  }
  vehicles[0]->AddUWBRadioTarget(2, Vec3f(0, 0, 0));   // vehicle 1 wants to range to radio 2
  RangingService net(&simTimer, 0.0015, 0.05, 0.0, 3.0);
  for (size_t k = 0; k < vehicles.size(); k++) net.radios.push_back(vehicles[k]->GetRadio());
  RadioTypes::RadioMessageDecoded::RawMessage raw;
  const float w0[3] = {0.1f, -0.2f, 0.3f};
  afe_radio_create_rates_command(0, 9.81f, w0, raw.raw);

  std::printf("{\"precision\": %d, \"handled_errors\": %d, \"handled_status\": %d, \"trace\": [\n", precision, handled, handled_status);
  for (int s = 0; s < runs; s++) {
    for (size_t k = 0; k < vehicles.size(); k++) vehicles[k]->Run();   // main.cpp:323-325
    net.Poll();
    simTimer.AdvanceMicroSeconds(dt_us);
    if (s == 1) vehicles[0]->SetCommandRadioMsg(raw);
    if (s == 2) vehicles[1]->SetVelocity(Vec3d(0, 0, 0));              // a setter between two Run()s must take effect
    Vec3d g, a;
    vehicles[0]->GetRateGyro(g);
    vehicles[0]->GetAccelerometer(a);
    TelemetryPacket::data_packet_t t1, t2;
    vehicles[0]->GetTelemetryDataPackets(t1, t2);
    std::printf(" {");
    for (size_t k = 0; k < vehicles.size(); k++) {
      const Vec3d p = vehicles[k]->GetPosition(), v = vehicles[k]->GetVelocity(), w = vehicles[k]->GetAngularVelocity();
      const Rotationd q = vehicles[k]->GetAttitude();
      std::printf("\"pos%zu\": [%.17g, %.17g, %.17g], \"vel%zu\": [%.17g, %.17g, %.17g], \"att%zu\": [%.17g, %.17g, %.17g, %.17g], "
                  "\"ang_vel%zu\": [%.17g, %.17g, %.17g], ", k, p.x, p.y, p.z, k, v.x, v.y, v.z, k, q[0], q[1], q[2], q[3], k, w.x, w.y, w.z);
    }
    std::printf("\"gyro\": [%.9g, %.9g, %.9g], \"acc\": [%.9g, %.9g, %.9g], \"runs\": %d, \"n_radio\": %d, \"radio_type\": %d, "
                "\"telemetry\": [%d, %d, %d], \"n_uwb\": [%d, %d], \"uwb_range\": %.9g, \"uwb_responder\": %d, "
                "\"radio_pos\": [%s, %s, %s]}%s\n",
                g.x, g.y, g.z, a.x, a.y, a.z, q1->Logic().runs, q1->Logic().n_radio, q1->Logic().last_radio_type, t1.type, t2.type,
                t1.packetNumber, q1->Logic().n_uwb, q2->Logic().n_uwb, q1->Logic().uwb_range, q1->Logic().uwb_responder,
                num(vehicles[1]->GetRadio()->GetPosition().x).c_str(), num(vehicles[1]->GetRadio()->GetPosition().y).c_str(),
                num(vehicles[1]->GetRadio()->GetPosition().z).c_str(), s + 1 < runs ? "," : "");
  }
  std::printf("]}\n");
  return 0;
}
