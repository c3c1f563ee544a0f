// host_loop_probe.cpp -- what one `vehicle->Run(); vehicle->GetPosition();` costs through the C ABI when the host is in
// the loop of every step (the shape of Simulator/Rappids_Simulator/main.cpp:330-392 with the onboard logic on the host,
// Quadcopter_T.cpp:159-189): per step afe_step(1) + afe_get_state, and on every logic tick afe_get_imu +
// afe_set_motor_cmds.  Device arena vs host-visible arena (afe_create_host_visible), launches vs the resident grid.
//   g++ -O2 -std=c++11 -Iinclude tools/host_loop_probe.cpp -o tools/host_loop_probe.bin -Lagri-fly_amd/lib -lagrifly_engine \
//       -Wl,-rpath,$PWD/agri-fly_amd/lib -Wl,-rpath,/opt/rocm/lib && tools/host_loop_probe.bin
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "agrifly_engine.h"

#define CK(x) do { int rc_ = (x); if (rc_) { std::fprintf(stderr, "%s -> %d (%s)\n", #x, rc_, e ? afe_last_error(e) : ""); std::exit(1); } } while (0)

static double g_part[4];   // where a step's time goes: afe_step, afe_get_state (includes the wait), afe_get_imu, afe_set_motor_cmds
static inline double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static double loop(bool host_visible, int mode, int64_t n, int precision, int steps, double *checksum) {
  afe_engine *e = nullptr;
  CK(host_visible ? afe_create_host_visible(&e, n, precision, -1, 0) : afe_create(&e, n, precision, -1, 0));
  afe_vehicle_params P;
  CK(afe_params_from_type(5, &P));
  CK(afe_set_type_table(e, &P, 1));
  CK(afe_set_logic_period(e, 1.0 / 500));
  CK(afe_set_imu_noise(e, 1, 0.1, 0.2, AFE_SEED_REFERENCE));
  CK(afe_set_step_mode(e, mode));
  std::vector<double> pos(3 * n), vel(3 * n), att(4 * n), w(3 * n);
  std::vector<float> gyro(3 * n), acc(3 * n), cmd(4 * n, 2839.27f);
  for (int64_t i = 0; i < n; i++) { pos[2 * n + i] = 3.5; att[i] = 1.0; }
  CK(afe_set_state(e, 0, n, pos.data(), vel.data(), att.data(), w.data(), nullptr));
  CK(afe_set_motor_cmds(e, 0, n, cmd.data()));
  double best = 1e30;
  for (int rep = 0; rep < 4; rep++) {
    const auto t0 = std::chrono::steady_clock::now();
    double part[4] = {0, 0, 0, 0};
    for (int s = 0; s < steps; s++) {
      uint64_t before = 0, after = 0;
      CK(afe_logic_ticks(e, &before));
      const double a0 = now_us();
      CK(afe_step(e, 1000, 1));
      const double a1 = now_us();
      CK(afe_logic_ticks(e, &after));
      CK(afe_get_state(e, 0, n, pos.data(), vel.data(), att.data(), w.data(), nullptr));
      const double a2 = now_us();
      part[0] += a1 - a0; part[1] += a2 - a1;
      if (after != before) {
        CK(afe_get_imu(e, 0, n, gyro.data(), acc.data()));
        const double a3 = now_us();
        for (int64_t i = 0; i < 4 * n; i++) cmd[(size_t)i] = 2839.27f + 0.01f * gyro[(size_t)(i % n)];   // stands for the onboard logic
        CK(afe_set_motor_cmds(e, 0, n, cmd.data()));
        part[2] += a3 - a2; part[3] += now_us() - a3;
      }
    }
    for (int k = 0; k < 4; k++) g_part[k] = part[k] / steps;
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / steps;
    if (us < best) best = us;
  }
  *checksum = pos[2 * n] + vel[2 * n] + att[0] + gyro[0];
  CK(afe_destroy(e));
  return best;
}

int main(int argc, char **argv) {
  const int steps = argc > 1 ? std::atoi(argv[1]) : 4000;
  const char *modes[] = {"launches", "resident grid"};
  for (int precision = 0; precision < 2; precision++)
    for (int64_t n : {(int64_t)1, (int64_t)64, (int64_t)1024, (int64_t)4096}) {
      double sums[4];
      int k = 0;
      std::printf("%s, %5lld vehicle(s):", precision ? "fp64" : "fp32", (long long)n);
      for (int hv = 0; hv < 2; hv++)
        for (int m = 0; m < 2; m++) {
          const double us = loop(hv != 0, m ? AFE_STEP_PERSISTENT : AFE_STEP_LAUNCH, n, precision, steps, &sums[k]);
          std::printf("  %s/%s %.2f us [step %.2f, get_state %.2f, get_imu %.2f, set_cmds %.2f]", hv ? "host-visible" : "device arena", modes[m], us,
                      g_part[0], g_part[1], g_part[2], g_part[3]);
          k++;
        }
      bool same = sums[0] == sums[1] && sums[1] == sums[2] && sums[2] == sums[3];
      std::printf("  per step%s\n", same ? "" : "  (CHECKSUMS DIFFER)");
      std::fflush(stdout);
    }
  return 0;
}
