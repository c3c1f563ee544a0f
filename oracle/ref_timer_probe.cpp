// oracle/ref_timer_probe.cpp -- TEST INFRASTRUCTURE.
// Drives the REFERENCE's own clock classes (compiled from where they lie:
// Common/Common/Time/{BaseTimer,ManualTimer,Timer}.hpp and
// Components/Components/Simulation/CommunicationsDelay.hpp, none of which need
// Eigen) through the exact sequence of calls Quadcopter_T::Run makes
// (Components/Components/Simulation/Quadcopter_T.cpp:87-91,159-160) inside the
// Rappids_Simulator loop (Simulator/Rappids_Simulator/main.cpp:391-392), and
// prints what the restated clock in agrifly_oracle.c must reproduce.
//
// usage: timer_probe <loop_dt_seconds> <logic_period_seconds> <n_runs>
// prints one JSON object.
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
#include "Common/Time/ManualTimer.hpp"
#include "Common/Time/Timer.hpp"
#include "Components/Simulation/CommunicationsDelay.hpp"

int main(int argc, char **argv) {
  if (argc < 4) return 2;
  const double loopDt = atof(argv[1]);
  const double period = atof(argv[2]);
  const int nRuns = atoi(argv[3]);

  ManualTimer simTimer;
  Timer integrationTimer(&simTimer);   // SimulationObject::_integrationTimer
  Timer timerOnboardLogic(&simTimer);  // Quadcopter_T::_timerOnboardLogic
  Simulation::CommunicationsDelay<int> radio(&simTimer, 0.03);  // main.cpp:282

  printf("{\"loop_dt\": %.17g, \"period\": %.17g, \"advance_us\": %llu,\n",
         loopDt, period, (unsigned long long) uint64_t(loopDt * 1e6));
  printf(" \"dt\": [");
  int *ticks = (int*) calloc(nRuns, sizeof(int));
  int *delivered = (int*) calloc(nRuns, sizeof(int));
  for (int s = 0; s < nRuns; s++) {
    // ---- Quadcopter_T::Run() timing skeleton ----
    const double dt = integrationTimer.GetSeconds<double>();
    if (dt < 1e-6) {
      printf("%s0", s ? ", " : "");
    } else {
      integrationTimer.Reset();
      printf("%s%.17g", s ? ", " : "", dt);
      if (timerOnboardLogic.GetSeconds<double>() > period) {
        timerOnboardLogic.AdjustTimeBySeconds(-period);
        ticks[s] = 1;
      }
    }
    // ---- main loop ----
    simTimer.AdvanceMicroSeconds(uint64_t(loopDt * 1e6));
    // one message enqueued every 10 runs, tagged with the run index
    if (s % 10 == 0) radio.AddMessage(s);
    delivered[s] = -1;
    if (radio.HaveNewMessage()) delivered[s] = radio.GetMessage();
  }
  printf("],\n \"tick\": [");
  for (int s = 0; s < nRuns; s++) printf("%s%d", s ? ", " : "", ticks[s]);
  printf("],\n \"radio_delivered\": [");
  for (int s = 0; s < nRuns; s++) printf("%s%d", s ? ", " : "", delivered[s]);
  printf("]}\n");
  return 0;
}
