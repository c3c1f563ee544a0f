// oracle/ref_lpf_probe.cpp -- TEST INFRASTRUCTURE.
// Known answers from the REFERENCE's own second-order low-pass, compiled from
// where it lies (Common/Common/Math/LowPassFilterSecondOrder.hpp needs only
// <math.h>), instantiated the way the onboard logic does
// (LowPassFilterSecondOrder<float, float>, Components/Components/Logic/
// QuadcopterLogic.hpp:322) with its gyro / accelerometer settings
// (QuadcopterLogic.cpp:102-103,130-134).
// usage: lpf_probe <period> <cutoff> <n>   -> JSON {coefficient-free: outputs}
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include "Common/Math/LowPassFilterSecondOrder.hpp"

int main(int argc, char **argv) {
  if (argc < 4) return 2;
  const float period = (float) atof(argv[1]);
  const float cutoff = (float) atof(argv[2]);
  const int n = atoi(argv[3]);
  LowPassFilterSecondOrder<float, float> f;
  f.Initialise(period, cutoff, 0.0f);
  printf("{\"period\": %.9g, \"cutoff\": %.9g, \"input\": [", period, cutoff);
  // deterministic test signal: step + chirp + alternating spikes
  float *in = (float*) malloc(sizeof(float) * n);
  for (int k = 0; k < n; k++) {
    in[k] = (k < 5 ? 0.0f : 1.0f) + 0.5f * sinf(0.013f * k * k) + ((k % 7) == 0 ? -2.25f : 0.125f);
    printf("%s%.9g", k ? ", " : "", in[k]);
  }
  printf("], \"output\": [");
  for (int k = 0; k < n; k++) printf("%s%.9g", k ? ", " : "", f.Apply(in[k]));
  printf("], \"final_value\": %.9g}\n", f.GetValue());
  return 0;
}
