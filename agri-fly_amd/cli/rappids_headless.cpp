// rappids_headless -- the Rappids_Simulator loop (Simulator/Rappids_Simulator/main.cpp:116-749) for a
// whole ensemble, headless, over the C ABI of the MI355X engine.
//
// What is the reference's:  the loop order and cadences (:140-142,174-178,330,391-392,451-476,611-673,
//   737-739), the Timer / strict-'>' gates, the 16-bit radio uplink and its 30 ms CommunicationsDelay,
//   the CSV log with exactly the columns of :266-270 written exactly as :676-733 writes them.
// What runs on the GPU:     quad->Run() for every vehicle -- physics, motors, IMU synthesis and the
//   rates-control slice of Onboard::QuadcopterLogic (afe_set_rates_logic) -- one launch per step.
// What is stubbed, and says so in the log header comment line when --verbose:
//   * AirSim / Unity (:183-184,300-321,331-389,397-438): absent.  With --scene <mesh> the depth image of every
//     vehicle comes from the engine's own depth camera at the reference's 30 Hz (afe_render_depth_engine, poses
//     read from the state slabs, images stay in HBM) and DepthImagePlanner runs on the GPU for all vehicles at once
//     (afe_rappids_plan_device; a candidate count instead of the 50 ms wall-clock budget of :502): the flight
//     branch of the loop, :478-608 -- plan on a ready image, keep the trajectory's frame, track it with
//     RunTracking (:629-632), hover at (0, 0, 2) until the first plan (:553-556).  Without --scene only the
//     take-off branch runs (hover at (0, 0, 3.5)), whatever the time.
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
//                    [--scene mesh.f32 [--goal x y z] [--start-flight T] [--candidates K] [--traj-log file]
//                     [--image-log file] [--hover z] [--line-up dy]]
// --scene: n x 9 little-endian float32 (v0, v1, v2 per triangle), world frame = the simulation's (z up).
// --goal: goalWorld (:241; default 120 0 3.5); --start-flight: startFlightTime (:141; default 5 s);
// --candidates: candidates per plan (default 256); --hover z: fly at height z before and after take-off instead
// of the reference's 3.5 m / 2 m; --line-up dy: vehicle i starts, hovers and aims dy * i metres further along y.
// --traj-log: PlannedTrajectory.csv of the logged vehicle, the reference's columns (:534-551) followed by what the
//   plan was made from (time, winning candidate, planner inputs, the pose its image was rendered at).
// --image-log: time, an FNV-1a checksum and the pose of every depth image of the logged vehicle (development aid: what
//   tools/experiments/flight_repro.py compares between runs).
// --print-seconds: the logged vehicle's true state on stdout after every whole second of Run() calls
//   ("t=1.000 pos=x y z vel=... q=... f0=..."), the line format SURVEY.md Appendix B quotes for the reference.
// Defaults are the reference's own: 1 vehicle, dt = 1/500 s, 8 s, 6 significant digits (ofstream default).
#include <chrono>
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
#include "planned_trajectory.hpp"

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
  std::string scenePath, trajLogPath, imageLogPath;
  double goalArg[3] = {120.0, 0.0, 3.5};   // main.cpp:241
  double startFlightTime = 5.0;            // :141
  double hoverArg = -1, lineUp = 0;
  int nCandidates = 256;
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
    else if (k == "--scene") { scenePath = v; a++; }
    else if (k == "--traj-log") { trajLogPath = v; a++; }
    else if (k == "--image-log") { imageLogPath = v; a++; }      // development aid: time, pose and a checksum of every depth image of the logged vehicle
    else if (k == "--goal" && a + 3 < argc) { for (int c = 0; c < 3; c++) goalArg[c] = atof(argv[a + 1 + c]); a += 3; }
    else if (k == "--start-flight") { startFlightTime = atof(v); a++; }
    else if (k == "--candidates") { nCandidates = atoi(v); a++; }
    else if (k == "--hover") { hoverArg = atof(v); a++; }
    else if (k == "--line-up") { lineUp = atof(v); a++; }
    else { std::fprintf(stderr, "rappids_headless: unknown option %s\n", k.c_str()); return 2; }
  }
  if (nVehicles < 1 || logVehicle < 0 || logVehicle >= nVehicles || digits < 1 || nCandidates < 1) return 2;

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
  // the offboard loop below visits the ensemble every few steps (mocap at 200 Hz, commands at 100 Hz, the log every step):
  // small ensembles keep their state in host-visible memory, so that those visits are host copies beside a resident grid
  if (nVehicles <= 4096) die(0, afe_create_host_visible(&quad, nVehicles, precision, -1, 0), "afe_create_host_visible");
  else die(0, afe_create(&quad, nVehicles, precision, -1, 0), "afe_create");
  die(quad, afe_set_type_table(quad, &vehConsts, 1), "afe_set_type_table");
  die(quad, afe_set_logic_period(quad, periodOnboardLogic), "afe_set_logic_period");
  die(quad, afe_set_imu_noise(quad, 1, 0.1, 0.2, seeds), "afe_set_imu_noise");       // Quadcopter_T.cpp:5-6
  die(quad, afe_set_rates_logic(quad, &logicConsts, 1), "afe_set_rates_logic");
  // quad->SetPosition(initErrPos); quad->SetAttitude(initErrAtt): origin, identity (:279-280) = the engine's initial state

  if (lineUp != 0) {   // vehicle i starts dy * i further along y (quad->SetPosition, :279)
    std::vector<double> p0(3 * nVehicles, 0.0), zero3(3 * nVehicles, 0.0), q0(4 * nVehicles, 0.0);
    for (int64_t i = 0; i < nVehicles; i++) { p0[nVehicles + i] = lineUp * (double)i; q0[i] = 1.0; }
    die(quad, afe_set_state(quad, 0, nVehicles, p0.data(), zero3.data(), q0.data(), zero3.data(), 0), "afe_set_state");
  }

  // the depth camera and the planner, main.cpp:119-125,166-170 (only with --scene)
  afe_scene *scene = 0;
  afe_camera cam;
  double depthCamAtt[4];
  afe_planner_config planCfg;
  void *depthImages = 0;                 // [nVehicles][240][320] uint16 in HBM
  std::vector<double> candidateSamples;
  die(0, afe_camera_default(&cam, 320, 240), "afe_camera_default");
  die(0, afe_camera_default_mount(depthCamAtt), "afe_camera_default_mount");
  const Rotationd camAtt(depthCamAtt[0], depthCamAtt[1], depthCamAtt[2], depthCamAtt[3]);
  if (!scenePath.empty()) {
    std::ifstream f(scenePath.c_str(), std::ios::binary | std::ios::ate);
    if (!f) { std::fprintf(stderr, "rappids_headless: cannot open %s\n", scenePath.c_str()); return 1; }
    const std::streamsize bytes = f.tellg();
    if (bytes <= 0 || bytes % 36) { std::fprintf(stderr, "rappids_headless: %s is not n x 9 float32\n", scenePath.c_str()); return 1; }
    std::vector<float> tris((size_t)bytes / 4);
    f.seekg(0);
    f.read((char *)tris.data(), bytes);
    die(0, afe_scene_create(-1, tris.data(), (int64_t)(bytes / 36), &scene), "afe_scene_create");
    // physicalVehicleRadius = 2 armLength, vehicleRadiusPlanning = 3 armLength, minCollisionDist 0.5 (:167-169)
    die(0, afe_planner_default_config(&planCfg, cam.width, cam.height, cam.depth_scale, cam.focal_length, 2 * vehConsts.arm_length,
                                      3 * vehConsts.arm_length, 0.5), "afe_planner_default_config");
    planCfg.cost_type = 1;               // ExplorationCost::GetTrajCost with the goal in the camera frame (:93-107)
    candidateSamples.resize((size_t)nCandidates * 4);
    die(0, afe_planner_samples(0, cam.width, cam.height, nCandidates, candidateSamples.data()), "afe_planner_samples");
    die(0, afe_device_alloc(-1, (uint64_t)nVehicles * cam.width * cam.height * 2, &depthImages), "afe_device_alloc");
  }

  std::vector<agrifly_cli::MocapEstimator> est;                              // :221-224, one per vehicle
  for (int64_t i = 0; i < nVehicles; i++) est.push_back(agrifly_cli::MocapEstimator(&simTimer, timeDelayOffboardControlLoopTrue));
  double const timeDelayOffboardControlLoopEstimate = 0.03;                  // :179
  agrifly_cli::HoverController ctrl;
  Vec3d desiredPosition(0, 0, hoverArg > 0 ? hoverArg : 3.5);   // :240
  Vec3d desiredVelocity(0, 0, 0);
  double desYawAngleDeg = 0;
  const Vec3d goalWorld(goalArg[0], goalArg[1], goalArg[2]);
  const Vec3d noTrajectoryYet(0, 0, hoverArg > 0 ? hoverArg : 2.0);          // :553-556
  // per vehicle: what main.cpp keeps in trajPlanned, _traj, trajAtt, trajOffset, trackTrajTime, previousThrust, imageReady
  struct Flight {
    bool planned, imageReady;
    agrifly_cli::PlannedTrajectory traj;
    Rotationd trajAtt;
    Vec3d trajOffset;
    double trajStart, previousThrust;
    double imagePose[7];
    Flight() : planned(false), imageReady(false), trajAtt(Rotationd::Identity()), trajOffset(0, 0, 0), trajStart(0), previousThrust(9.81) {}
  };
  std::vector<Flight> flight((size_t)nVehicles);
  bool requestNewImage = true;                                               // :199
  double const periodBetweenImages = 1 / 30.0;                               // :200-201
  int plannedTrajCount = 0;
  std::ofstream imageLog;
  if (!imageLogPath.empty()) { imageLog.open(imageLogPath.c_str()); imageLog << std::setprecision(17); }
  std::ofstream trajLog;
  if (!trajLogPath.empty()) { trajLog.open(trajLogPath.c_str()); trajLog << std::setprecision(17); }

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
  Timer timerRequestNewImage(&simTimer);
  float lastRadioCommand[4] = {0, 0, 0, 0};
  std::vector<double> pos(3 * nVehicles), vel(3 * nVehicles), att(4 * nVehicles), angVel(3 * nVehicles);
  std::vector<float> cmds(4 * nVehicles), gyro(3 * nVehicles), acc(3 * nVehicles);
  const int64_t N = nVehicles, L = logVehicle;

  long steps = 0;
  const long stepsPerSecond = (long)(1.0 / dt);
  double p1[3], v1[3], q1[4], w1[3], m1[4];
  agrifly_cli::Estimate estLogged;
  estLogged.att = Rotationd::Identity();
  Vec3d desiredPositionLogged(desiredPosition), desiredVelocityLogged(desiredVelocity);
  const std::chrono::steady_clock::time_point wall0 = std::chrono::steady_clock::now();   // (host wall clock of the loop itself: process start and HIP initialisation are outside)
  while (t.GetSeconds<double>() < endTime) {                                  // :330
    if (scene && requestNewImage) {                                            // :331-389: one DepthVis image per vehicle
      die(quad, afe_render_depth_engine(quad, scene, &cam, 0, nVehicles, depthCamAtt, depthImages, 1, 0), "afe_render_depth_engine");
      double rp[3], rv[3], rq[4], rw[3];
      die(quad, afe_get_state(quad, logVehicle, 1, rp, rv, rq, rw, 0), "afe_get_state");
      for (int64_t i = 0; i < nVehicles; i++) flight[(size_t)i].imageReady = true;
      for (int c = 0; c < 3; c++) flight[(size_t)logVehicle].imagePose[c] = rp[c];
      for (int c = 0; c < 4; c++) flight[(size_t)logVehicle].imagePose[3 + c] = rq[c];
      if (imageLog.is_open()) {
        std::vector<uint16_t> img((size_t)cam.width * cam.height);
        die(0, afe_device_download(img.data(), (const char *)depthImages + (size_t)logVehicle * img.size() * 2, img.size() * 2), "afe_device_download");
        uint64_t h = 1469598103934665603ull;      // FNV-1a
        for (uint16_t px : img) { h ^= px; h *= 1099511628211ull; }
        imageLog << t.GetSeconds<double>() << "," << h;
        for (int c = 0; c < 3; c++) imageLog << "," << rp[c];
        for (int c = 0; c < 4; c++) imageLog << "," << rq[c];
        imageLog << "\n";
      }
      requestNewImage = false;
    }
    {   // quad->Run(), Quadcopter_T.cpp:85-91: dt from the integration timer, nothing on the first call
      const uint64_t run_us = integrationTimer.GetMicroSeconds();
      if ((double)((double)run_us * 1e-6) >= 1e-6) {
        integrationTimer.Reset();
        die(quad, afe_step(quad, run_us, 1), "afe_step");
      }
    }
    simTimer.AdvanceMicroSeconds(uint64_t(dt * 1e6));                          // :392

    if (timerRequestNewImage.GetSeconds<double>() > periodBetweenImages) {     // :440-444
      requestNewImage = true;
      timerRequestNewImage.AdjustTimeBySeconds(-periodBetweenImages);
    }
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
      std::vector<agrifly_cli::Estimate> estStates((size_t)N);                 // :468-469
      for (int64_t i = 0; i < N; i++) {
        agrifly_cli::Estimate &estState = estStates[(size_t)i];
        if (useMocapEstimator) estState = est[(size_t)i].Predict(timeDelayOffboardControlLoopEstimate);
        else {
          estState.pos = Vec3d(pos[i], pos[N + i], pos[2 * N + i]); estState.vel = Vec3d(vel[i], vel[N + i], vel[2 * N + i]);
          estState.att = Rotationd(att[i], att[N + i], att[2 * N + i], att[3 * N + i]);
          estState.angVel = Vec3d(angVel[i], angVel[N + i], angVel[2 * N + i]);
        }
      }
      const bool flying = scene && t.GetSeconds<double>() > startFlightTime;   // :478
      if (flying) {                                                            // :479-552, every vehicle with a ready image at once
        std::vector<int32_t> who;
        for (int64_t i = 0; i < N; i++) if (flight[(size_t)i].imageReady) who.push_back((int32_t)i);
        const int64_t n = (int64_t)who.size();
        if (n > 0) {
          std::vector<double> vel0(3 * n), acc0(3 * n), grav(3 * n), goalCam(3 * n);
          for (int64_t k = 0; k < n; k++) {
            const int64_t i = who[(size_t)k];
            const agrifly_cli::Estimate &es = estStates[(size_t)i];
            const Vec3d offset(0, lineUp * (double)i, 0);
            const Rotationd toCam = camAtt.Inverse() * es.att.Inverse();
            const Vec3d v = toCam * es.vel;                                    // :490-495
            const Vec3d a = toCam * (Vec3d(0, 0, 1) * flight[(size_t)i].previousThrust - Vec3d(0, 0, 9.81));
            const Vec3d g = toCam * Vec3d(0, 0, -9.81);
            const Vec3d G = toCam * ((goalWorld + offset) - es.pos);           // ExplorationCost::GetTrajCost, :100-102
            for (int c = 0; c < 3; c++) { vel0[c * n + k] = v[c]; acc0[c * n + k] = a[c]; grav[c * n + k] = g[c]; goalCam[c * n + k] = G[c]; }
          }
          std::vector<afe_plan_output> plans((size_t)n);
          die(0, afe_rappids_plan_device(-1, &planCfg, n, depthImages, N, who.data(), vel0.data(), acc0.data(), grav.data(), goalCam.data(),
                                         candidateSamples.data(), 1, 0, nCandidates, plans.data(), 0, 0), "afe_rappids_plan_device");
          for (int64_t k = 0; k < n; k++) {
            if (!plans[(size_t)k].found) continue;                             // the image stays ready: the next tick plans on it again
            const int64_t i = who[(size_t)k];
            Flight &fl = flight[(size_t)i];
            const agrifly_cli::Estimate &es = estStates[(size_t)i];
            for (int q = 0; q < 6; q++) for (int c = 0; c < 3; c++) fl.traj.c[q][c] = plans[(size_t)k].coeffs[q][c];
            fl.traj.duration = plans[(size_t)k].tf;
            fl.traj.gravity = Vec3d(grav[k], grav[n + k], grav[2 * n + k]);
            fl.trajAtt = es.att * camAtt;                                      // :520
            fl.trajOffset = es.pos;
            fl.trajStart = t.GetSeconds<double>();                             // trackTrajTime.Reset()
            fl.planned = true;
            fl.imageReady = false;
            if (i == L) {
              plannedTrajCount++;
              if (trajLog.is_open()) {                                         // :534-551
                trajLog << plannedTrajCount << ",";
                for (int q = 0; q < 6; q++) trajLog << fl.traj.c[q][0] << "," << fl.traj.c[q][1] << "," << fl.traj.c[q][2] << ",";
                const Vec3d ypr = fl.trajAtt.ToEulerYPR();
                trajLog << ypr.x << "," << ypr.y << "," << ypr.z << ",";
                trajLog << fl.trajOffset.x << "," << fl.trajOffset.y << "," << fl.trajOffset.z << ",";
                trajLog << 0.0 << "," << fl.traj.duration;
                trajLog << "," << t.GetSeconds<double>() << "," << plans[(size_t)k].best_index;
                for (int c = 0; c < 3; c++) trajLog << "," << vel0[c * n + k];
                for (int c = 0; c < 3; c++) trajLog << "," << acc0[c * n + k];
                for (int c = 0; c < 3; c++) trajLog << "," << grav[c * n + k];
                for (int c = 0; c < 3; c++) trajLog << "," << goalCam[c * n + k];
                for (int c = 0; c < 7; c++) trajLog << "," << fl.imagePose[c];
                trajLog << "\n";
              }
            }
          }
        }
      }
      for (int64_t i = 0; i < N; i++) {                                        // :554-638 for every vehicle
        Vec3d cmdAngVel;
        double cmdThrust;
        const agrifly_cli::Estimate &estState = estStates[(size_t)i];
        Flight &fl = flight[(size_t)i];
        const Vec3d offset(0, lineUp * (double)i, 0);
        Vec3d wantPos = desiredPosition + offset, wantVel(0, 0, 0);
        if (!flying) {                                                         // :623-627
          ctrl.Run(estState.pos, estState.vel, estState.att, wantPos, Vec3d(0, 0, 0),
                   Vec3d(0, 0, 0), desYawAngleDeg * M_PI / 180.0, cmdAngVel, cmdThrust);
        } else {
          Vec3d wantAcc(0, 0, 0), wantAngVel(0, 0, 0);
          double wantThrust = 9.81;    // (the reference reads an uninitialised double here until the first plan)
          if (!fl.planned) wantPos = noTrajectoryYet + offset;                 // :553-556
          else {                                                               // :558-607
            double traj_t = t.GetSeconds<double>() - fl.trajStart;
            Vec3d trajPos, trajVel, trajAcc;
            if (traj_t < fl.traj.duration) {
              traj_t += 0.04;
              trajPos = fl.traj.Position(traj_t);
              trajVel = fl.traj.Velocity(traj_t);
              trajAcc = fl.traj.Acceleration(traj_t);
            } else {
              trajPos = fl.traj.Position(fl.traj.duration);
              trajVel = Vec3d(0, 0, 0);
              trajAcc = Vec3d(0, 0, 0);
            }
            if (trajPos.z < 0) {       // never backwards, behind the camera
              trajPos.z = 0;
              if (trajVel.z < 0) trajVel.z = 0;
              if (trajAcc.z < 0) trajAcc.z = 0;
            }
            wantPos = fl.trajAtt * trajPos + fl.trajOffset;
            wantVel = fl.trajAtt * trajVel;
            wantAcc = fl.trajAtt * trajAcc;
            wantThrust = fl.traj.Thrust(traj_t);
            wantAngVel = estState.att.Inverse() * (fl.trajAtt * fl.traj.Omega(traj_t, 0.02));
          }
          Rotationf cmdAtt;
          ctrl.RunTracking(estState.pos, estState.vel, estState.att, wantPos, wantVel, wantAcc, desYawAngleDeg * M_PI / 180.0,
                           wantThrust, wantAngVel, cmdAngVel, cmdThrust, cmdAtt);
        }
        fl.previousThrust = cmdThrust;                                         // :640
        if (i == L) { desiredPositionLogged = wantPos; desiredVelocityLogged = wantVel; }
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
      logfile << toCSV(desiredPositionLogged, digits);
      logfile << toCSV(desiredVelocityLogged, digits);
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
  const double loopWall = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count();
  std::printf("Loop wall time %.6f s for %ld steps\n", loopWall, steps);
  uint64_t ticks = 0, now = 0;
  afe_logic_ticks(quad, &ticks);
  afe_time_us(quad, &now);
  std::printf("Done. %lld vehicle(s), engine clock %.6f s, %llu logic ticks\n", (long long)N, now * 1e-6, (unsigned long long)ticks);
  logfile.close();
  if (scene) {
    std::printf("%d trajectories planned for vehicle %lld\n", plannedTrajCount, (long long)L);
    afe_device_free(depthImages);
    afe_scene_destroy(scene);
  }
  afe_destroy(quad);
  return 0;
}
