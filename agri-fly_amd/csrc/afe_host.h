// afe_host.h -- host-only helpers of the engine (no HIP): parameter expansion,
// vehicle-type table, clock / logic-gate planning.  Kept free of GPU calls so
// the CPU test job can exercise them through the C ABI without a device.
#pragma once
#include <stdint.h>

#include <cstdlib>
#include <cstring>

#include "../../include/agrifly_engine.h"
#include "afe_device.h"

namespace afe {

// Environment hooks.  The RELEASE library reads six variables, all documented in INTEGRATION.md section 3:
//   AFE_PERSIST_AQL, AFE_FORCE_STEP_MODE, AFE_FORCE_SPLIT, AFE_FORCE_HOST_ARENA, AFE_PERSIST_DEBUG, AFE_GRID_LOG
// (plain std::getenv at their call sites).  Every other variable is a measurement aid of the kernel lab (tools/) and is
// read only by a library built with -DAFE_DEV_HOOKS (make EXTRA=-DAFE_DEV_HOOKS); afe_has_dev_hooks() says which one
// is loaded.  In the release build this folds to nullptr and the code behind each hook is dead.
inline const char *afe_dev_env(const char *name) {
#ifdef AFE_DEV_HOOKS
  return std::getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

// Fault injection for the failure paths of the resident grid's own queue (tests/test_gpu_fault_paths.py), -DAFE_DEV_HOOKS
// builds only: AFE_FAULT=<name>[,<name>...] makes the named step fail the way the runtime could --
//   queue_create   hsa_queue_create refuses            kernel_symbol  the kernel descriptor is not found in the loaded code
//   kernarg_size   the code object declares another kernel-argument size than the host packs
//   park_timeout   a parked grid is reported as still running when the wait ends
//   sync_answer    the host never sees the grid's answer to a sync request (nor its park): afe_sync runs into its patience
// In the release build this folds to false.
inline bool afe_fault(const char *name) {
  const char *s = afe_dev_env("AFE_FAULT");
  if (!s) return false;
  const size_t n = std::strlen(name);
  for (const char *p = s; (p = std::strstr(p, name)) != nullptr; p += n)
    if ((p == s || p[-1] == ',') && (p[n] == 0 || p[n] == ',')) return true;
  return false;
}
// how long afe_sync waits for a resident grid that neither answers nor parks (seconds; AFE_SYNC_PATIENCE_S in a dev-hooks build)
inline int afe_sync_patience_s() {
  const char *s = afe_dev_env("AFE_SYNC_PATIENCE_S");
  return s && *s ? std::atoi(s) : 20;
}

// Expanded constants in double (what the Quadcopter_T ctor computes once).
struct HostParams {
  double mass;
  double I[9], Iinv[9];
  double mp[4][3];
  double kf, ktau, tau_m, Jm, wmin, wmax;
  double drag[3];
  float Rimu[9];
};

// Quadcopter_T ctor body, reference Quadcopter_T.cpp:20,45-65,75-80.
// Returns AFE_OK or AFE_ERR_INVALID_ARG (the reference asserts instead:
// Motor.cpp:27-30).
int expand_params(const afe_vehicle_params &in, HostParams &out, const char **why);

// Narrow to the kernel record for a given dt (c_lag = exp(-dt/tau_m)).
template <typename R>
void to_device_params(const HostParams &h, double dt, DevParams<R> &d);

// QuadcopterLogic::Initialise (QuadcopterLogic.cpp:97-150) for the rates slice:
// mixer constants, IMU mount matrix, gyro low-pass coefficients, all in float.
int expand_logic(const afe_rates_logic_params &in, float onboard_period, DevLogic &out, const char **why);

// Timer semantics, reference Common/Common/Time/Timer.hpp:27-54.
inline double us_to_seconds(uint64_t us) { return (double)((double)us * 1e-6); }
inline uint64_t period_to_us(double period_s) { return (uint64_t)((-period_s) * -1e6); }
// one step of dt_us: returns true when the gate of Quadcopter_T.cpp:159 fires
inline bool gate_step(double period_s, uint64_t &elapsed_us, uint64_t dt_us) {
  elapsed_us += dt_us;
  if (us_to_seconds(elapsed_us) > period_s) {
    elapsed_us -= period_to_us(period_s);
    return true;
  }
  return false;
}

}  // namespace afe
