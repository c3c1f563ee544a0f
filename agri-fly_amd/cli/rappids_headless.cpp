// rappids_headless -- the Rappids_Simulator loop (Simulator/Rappids_Simulator/main.cpp:116-749) for a
// whole ensemble, headless, over the C ABI of the MI355X engine.
//
// What is the reference's:  the loop order and cadences (:140-142,174-178,330,391-392,451-476,611-673,
//   737-739), the Timer / strict-'>' gates, the 16-bit radio uplink and its 30 ms CommunicationsDelay,
//   the CSV log with exactly the columns of :266-270 written exactly as :676-733 writes them.
// What runs on the GPU:     quad->Run() for every vehicle -- physics, motors, IMU synthesis and the
//   rates-control slice of Onboard::QuadcopterLogic (afe_set_rates_logic) -- one launch per step.
// What is stubbed, and says so in the log header comment line when --verbose:
//   * AirSim / Unity (:183-184,300-321,331-389,397-438): absent; with --scene orchard the depth image
//     comes from the engine's own depth camera and the RAPPIDS planner runs on the GPU (afe_rappids_plan).
//   * MocapStateEstimator (:221-224,451-457,468-469,647-649) and QuadcopterController::Run (:625-627): the tree's
//     Offboard/ sources are not part of this repository, so both are restated -- cli/mocap_estimator.hpp
//     (200 Hz truth pose in, prediction 30 ms ahead out, driven by the commands in flight) and HoverController
//     below (float, QuadcopterController.cpp:11-74); inside the agri-fly tree the reference's own classes drop
//     into the same calls.  --estimator truth hands the controller the true state instead.
//   * telemetry (:459-466,659-664): m1..m4 are the commanded motor forces k_f cmd^2 put through the
//     reference's telemetry quantisation (afe_telemetry_encode / _decode); panic is always 0.
//
//   rappids_headless [--vehicles N] [--seconds T] [--dt-us 2000] [--precision f32|f64]
//                    [--seeds reference|decorrelated] [--log-vehicle i] [--digits D] [--out simulation.csv]
//                    [--estimator mocap|truth] [--print-seconds]
// --print-seconds: the logged vehicle's true state on stdout after every whole second of Run() calls
//   ("t=1.000 pos=x y z vel=... q=... f0=..."), the line format SURVEY.md Appendix B quotes for the reference.
// Defaults are the reference's own: 1 vehicle, dt = 1/500 s, 8 s, 6 significant digits (ofstream default).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <sstream>
#include <string>
#include <vector>

#include "agrifly/Wire.hpp"
#include "hover_controller.hpp"
#include "mocap_estimator.hpp"

namespace {

template <typename Real>
std::string toCSV(const Vec3<Real> v, int digits) {   // main.cpp:55-60
  std::stringstream ss;
  ss << std::setprecision(digits) << v.x << "," << v.y << "," << v.z << ",";
  return ss.str();
}

void die(afe_engine *e, int rc, const char *what) {
  if (rc == AFE_OK) return;
  std::fprintf(stderr, "rappids_headless: %s failed: %s (%s)\n", what, afe_status_string(rc), e ? afe_last_error(e) : "");
  std::exit(1);
}

}  // namespace

int main(int argc, char **argv) {
  int64_t nVehicles = 1, logVehicle = 0;
  double endTime = 8.0;                  // main.cpp:142
  uint64_t dt_us_arg = 0;
  int precision = AFE_F32, seeds = AFE_SEED_REFERENCE, digits = 6;
  std::string outPath = "simulation.csv";
  bool useMocapEstimator = true, printSeconds = false;
  for (int a = 1; a < argc; a++) {
    const std::string k = argv[a];
    const char *v = a + 1 < argc ? argv[a + 1] : "";
    if (k == "--vehicles") { nVehicles = atoll(v); a++; }
    else if (k == "--seconds") { endTime = atof(v); a++; }
    else if (k == "--dt-us") { dt_us_arg = (uint64_t)atoll(v); a++; }
    else if (k == "--precision") { precision = std::strcmp(v, "f64") ? AFE_F32 : AFE_F64; a++; }
    else if (k == "--seeds") { seeds = std::strcmp(v, "decorrelated") ? AFE_SEED_REFERENCE : AFE_SEED_DECORRELATED; a++; }
    else if (k == "--log-vehicle") { logVehicle = atoll(v); a++; }
    else if (k == "--digits") { digits = atoi(v); a++; }
    else if (k == "--out") { outPath = v; a++; }
    else if (k == "--estimator") { useMocapEstimator = std::strcmp(v, "truth") != 0; a++; }
    else if (k == "--print-seconds") { printSeconds = true; }
    else { std::fprintf(stderr, "rappids_headless: unknown option %s\n", k.c_str()); return 2; }
  }
  if (nVehicles < 1 || logVehicle < 0 || logVehicle >= nVehicles || digits < 1) return 2;

  // Basic timing, main.cpp:139-144
  const double dt = dt_us_arg ? dt_us_arg * 1e-6 : 1.0 / 500.0;
  ManualTimer simTimer;
  // the vehicle(s): id 1 -> QC_TYPE_CF_MINIQUAD (main.cpp:146-164,211-218)
  uint8_t vehicleId = 1;
  afe_vehicle_params vehConsts;
  die(0, afe_params_from_type(afe_type_from_id(vehicleId), &vehConsts), "afe_params_from_type");
  afe_rates_logic_params logicConsts;
  die(0, afe_rates_logic_params_from_type(afe_type_from_id(vehicleId), &logicConsts), "afe_rates_logic_params_from_type");
  double const periodMocapSystem = 1.0 / 200.0;        // :174
  double const periodOffboardMainLoop = 1.0 / 100.0;   // :175
  double const periodTelemetryLoop = 1.0 / 100.0;      // :176
  double const periodOnboardLogic = 1.0 / 500.0;       // :177
  double const timeDelayOffboardControlLoopTrue = 0.03;   // :178

  afe_engine *quad = 0;
  die(0, afe_create(&quad, nVehicles, precision, -1, 0), "afe_create");
  die(quad, afe_set_type_table(quad, &vehConsts, 1), "afe_set_type_table");
  die(quad, afe_set_logic_period(quad, periodOnboardLogic), "afe_set_logic_period");
  die(quad, afe_set_imu_noise(quad, 1, 0.1, 0.2, seeds), "afe_set_imu_noise");       // Quadcopter_T.cpp:5-6
  die(quad, afe_set_rates_logic(quad, &logicConsts, 1), "afe_set_rates_logic");
  // quad->SetPosition(initErrPos); quad->SetAttitude(initErrAtt): origin, identity (:279-280) = the engine's initial state

  std::vector<agrifly_cli::MocapEstimator> est;                              // :221-224, one per vehicle
  for (int64_t i = 0; i < nVehicles; i++) est.push_back(agrifly_cli::MocapEstimator(&simTimer, timeDelayOffboardControlLoopTrue));
  double const timeDelayOffboardControlLoopEstimate = 0.03;                  // :179
  agrifly_cli::HoverController ctrl;
  Vec3d desiredPosition(0, 0, 3.5);   // :240
  Vec3d desiredVelocity(0, 0, 0);
  double desYawAngleDeg = 0;

  std::ofstream logfile(outPath.c_str());
  if (!logfile) { std::fprintf(stderr, "rappids_headless: cannot open %s\n", outPath.c_str()); return 1; }
  logfile << std::setprecision(digits);
  logfile << "t,posx,posy,posz,velx,vely,velz,attY,attP,attR,angvelx,angvely,angvelz,m1,m2,m3,m4,"
             "estposx,estposy,estposz,estvelx,estvely,estvelz,esty,estp,estr,estangx,estangy,estangz,"
             "desposx,desposy,desposz,desvelx,desvely,desvelz,panic,r1,r2,r3,r4\n";      // :266-270
  std::printf("Starting simulation\n");

  Timer t(&simTimer);
  Timer integrationTimer(&simTimer);       // SimulationObject::_integrationTimer
  typedef std::vector<agrifly::RawRadioMessage> RadioBatch;   // one uplink packet per vehicle
  agrifly::DelayLine<RadioBatch> cmdRadioChannel(&simTimer, timeDelayOffboardControlLoopTrue);   // :282
  Timer timerPrint(&simTimer), timerMocap(&simTimer), timerOffboardMainLoop(&simTimer), timerTelemetryLoop(&simTimer);
  float lastRadioCommand[4] = {0, 0, 0, 0};
  std::vector<double> pos(3 * nVehicles), vel(3 * nVehicles), att(4 * nVehicles), angVel(3 * nVehicles);
  std::vector<float> cmds(4 * nVehicles), gyro(3 * nVehicles), acc(3 * nVehicles);
  const int64_t N = nVehicles, L = logVehicle;

  long steps = 0;
  const long stepsPerSecond = (long)(1.0 / dt);
  double p1[3], v1[3], q1[4], w1[3], m1[4];
  agrifly_cli::Estimate estLogged;
  estLogged.att = Rotationd::Identity();
  while (t.GetSeconds<double>() < endTime) {                                  // :330
    {   // quad->Run(), Quadcopter_T.cpp:85-91: dt from the integration timer, nothing on the first call
      const uint64_t run_us = integrationTimer.GetMicroSeconds();
      if ((double)((double)run_us * 1e-6) >= 1e-6) {
        integrationTimer.Reset();
        die(quad, afe_step(quad, run_us, 1), "afe_step");
      }
    }
    simTimer.AdvanceMicroSeconds(uint64_t(dt * 1e6));                          // :392

    if (timerPrint.GetSeconds<double>() >= 1) {                                // :446-449
      timerPrint.AdjustTimeBySeconds(-1);
      std::printf("Current sim time = %.1fs\n", t.GetSeconds<double>());
    }
    steps++;
    if (printSeconds && steps % stepsPerSecond == 0) {
      die(quad, afe_get_state(quad, L, 1, p1, v1, q1, w1, m1), "afe_get_state");
      std::printf("t=%.3f pos=%.9g %.9g %.9g vel=%.6g %.6g %.6g q=%.9g %.6g %.6g %.6g w=%.4g %.4g %.4g f0=%.6g\n",
                  t.GetSeconds<double>(), p1[0], p1[1], p1[2], v1[0], v1[1], v1[2], q1[0], q1[1], q1[2], q1[3], w1[0], w1[1], w1[2],
                  vehConsts.prop_thrust_from_speed_sqr * m1[0] * std::fabs(m1[0]));
    }
    if (timerMocap.GetSeconds<double>() > periodMocapSystem) {                 // :451-457
      timerMocap.AdjustTimeBySeconds(-periodMocapSystem);
      if (useMocapEstimator) {
        die(quad, afe_get_state(quad, 0, N, pos.data(), vel.data(), att.data(), angVel.data(), 0), "afe_get_state");
        for (int64_t i = 0; i < N; i++)
          est[(size_t)i].Measure(Vec3d(pos[i], pos[N + i], pos[2 * N + i]), Rotationd(att[i], att[N + i], att[2 * N + i], att[3 * N + i]));
      }
    }
    if (timerTelemetryLoop.GetSeconds<double>() > periodTelemetryLoop)         // :459-466 (no consumer)
      timerTelemetryLoop.AdjustTimeBySeconds(-periodTelemetryLoop);

    if (timerOffboardMainLoop.GetSeconds<double>() > periodOffboardMainLoop) { // :471
      timerOffboardMainLoop.AdjustTimeBySeconds(-periodOffboardMainLoop);      // :476
      die(quad, afe_get_state(quad, 0, N, pos.data(), vel.data(), att.data(), angVel.data(), 0), "afe_get_state");
      RadioBatch batch((size_t)N);
      for (int64_t i = 0; i < N; i++) {                                        // :611-638 for every vehicle
        Vec3d cmdAngVel;
        double cmdThrust;
        agrifly_cli::Estimate estState;                                        // :468-469
        if (useMocapEstimator) estState = est[(size_t)i].Predict(timeDelayOffboardControlLoopEstimate);
        else {
          estState.pos = Vec3d(pos[i], pos[N + i], pos[2 * N + i]); estState.vel = Vec3d(vel[i], vel[N + i], vel[2 * N + i]);
          estState.att = Rotationd(att[i], att[N + i], att[2 * N + i], att[3 * N + i]);
          estState.angVel = Vec3d(angVel[i], angVel[N + i], angVel[2 * N + i]);
        }
        ctrl.Run(estState.pos, estState.vel, estState.att, desiredPosition, Vec3d(0, 0, 0),
                 Vec3d(0, 0, 0), desYawAngleDeg * M_PI / 180.0, cmdAngVel, cmdThrust);
        if (useMocapEstimator)                                                 // :647-649
          est[(size_t)i].Announce(cmdAngVel, (estState.att * Vec3d(0, 0, 1) * cmdThrust - Vec3d(0, 0, 9.81)));
        if (i == L) estLogged = useMocapEstimator ? est[(size_t)i].Predict(0) : estState;   // :699-704
        batch[(size_t)i] = agrifly::MakeRatesCommand(0, float(cmdThrust), Vec3f(cmdAngVel));   // CreateRatesCommand
        if (i == L) {
          lastRadioCommand[0] = cmdThrust;                                     // :643-646
          lastRadioCommand[1] = cmdAngVel.x;
          lastRadioCommand[2] = cmdAngVel.y;
          lastRadioCommand[3] = cmdAngVel.z;
        }
      }
      // telemetry of the logged vehicle, :659-664
      die(quad, afe_get_motor_cmds(quad, 0, N, cmds.data()), "afe_get_motor_cmds");
      die(quad, afe_get_imu(quad, 0, N, gyro.data(), acc.data()), "afe_get_imu");
      afe_telemetry_packet tp, dataPacket;
      std::memset(&tp, 0, sizeof(tp));
      std::memset(&dataPacket, 0, sizeof(dataPacket));
      tp.type = 0;
      for (int k = 0; k < 3; k++) { tp.accel[k] = acc[k * N + L]; tp.gyro[k] = gyro[k * N + L]; tp.position[k] = float(pos[k * N + L]); }
      for (int m = 0; m < 4; m++) tp.motor_forces[m] = float(vehConsts.prop_thrust_from_speed_sqr) * cmds[m * N + L] * cmds[m * N + L];
      uint8_t wire[AFE_TELEMETRY_PACKET_SIZE];
      afe_telemetry_encode(&tp, wire);
      afe_telemetry_decode(wire, &dataPacket);

      cmdRadioChannel.AddMessage(batch);                                       // :673

      // the log row, :676-733
      const Vec3d p(pos[L], pos[N + L], pos[2 * N + L]), v(vel[L], vel[N + L], vel[2 * N + L]);
      const Vec3d w(angVel[L], angVel[N + L], angVel[2 * N + L]);
      const Rotationd q(att[L], att[N + L], att[2 * N + L], att[3 * N + L]);
      logfile << t.GetSeconds<double>() << ",";
      logfile << toCSV(p, digits);
      logfile << toCSV(v, digits);
      logfile << toCSV(q.ToEulerYPR(), digits);
      logfile << toCSV(w, digits);
      for (int m = 0; m < 4; m++) logfile << dataPacket.motor_forces[m] << ",";
      // estimator state, :694-716: GetPrediction(0), narrowed to float
      logfile << toCSV(Vec3f(estLogged.pos), digits);
      logfile << toCSV(Vec3f(estLogged.vel), digits);
      logfile << toCSV(Rotationf(estLogged.att).ToEulerYPR(), digits);
      logfile << toCSV(Vec3f(estLogged.angVel), digits);
      logfile << toCSV(desiredPosition, digits);
      logfile << toCSV(desiredVelocity, digits);
      logfile << int(dataPacket.panic_reason) << ",";
      for (int i = 0; i < 4; i++) logfile << double(lastRadioCommand[i]) << ",";
      logfile << "\n";
    }

    if (cmdRadioChannel.HaveNewMessage()) {                                    // :737-739
      const RadioBatch msg = cmdRadioChannel.GetMessage();
      die(quad, afe_set_commands_from_radio(quad, 0, N, msg[0].raw), "afe_set_commands_from_radio");
    }
  }
  die(quad, afe_sync(quad), "afe_sync");
  uint64_t ticks = 0, now = 0;
  afe_logic_ticks(quad, &ticks);
  afe_time_us(quad, &now);
  std::printf("Done. %lld vehicle(s), engine clock %.6f s, %llu logic ticks\n", (long long)N, now * 1e-6, (unsigned long long)ticks);
  logfile.close();
  afe_destroy(quad);
  return 0;
}
