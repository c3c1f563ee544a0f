// oracle/ref_uwb_probe.cpp -- TEST INFRASTRUCTURE.
// Known-answer generator for the UWB ranging noise stream.  The reference draws
// it from libstdc++ (Components/Components/Simulation/UWBNetwork.cpp:4-6:
// file-scope std::mt19937, std::uniform_real_distribution<double_t>(0,1),
// std::normal_distribution<double>(0,1); rng.seed(0) at :19), a third-party
// library that IS present in this image.  This probe declares the same three
// objects and runs the statement sequence of UWBNetwork::Run's completion branch
// (:66-71) with unit geometry, so the restated stream in
// agrifly_oracle_world.c is pinned against the real thing.  (UWBNetwork.cpp
// itself cannot be compiled here: Vec3.hpp -> Matrix.hpp -> <Eigen/Dense>.)
//
// usage: uwb_probe <n> <noiseStdDev> <outlierProbability> <outlierStdDev>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

std::mt19937 rng;
std::uniform_real_distribution<double_t> distUniform(0, 1);
std::normal_distribution<double> distNormal(0, 1);

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 64;
  const double addNoiseStdDev = argc > 2 ? atof(argv[2]) : 0.05;
  const double outlierProbability = argc > 3 ? atof(argv[3]) : 0.1;
  const double outlierStdDev = argc > 4 ? atof(argv[4]) : 3.0;
  {  // raw pieces of the stream from the same seed
    std::mt19937 g;
    g.seed(0);
    printf("{\"raw\": [");
    for (int i = 0; i < 8; i++) printf("%s%lu", i ? ", " : "", (unsigned long)g());
    printf("],\n");
    std::mt19937 g2;
    g2.seed(0);
    printf(" \"canonical\": [");
    for (int i = 0; i < 8; i++) printf("%s%.17g", i ? ", " : "", std::generate_canonical<double, 53>(g2));
    printf("],\n");
    std::mt19937 g3;
    g3.seed(0);
    std::normal_distribution<double> nd(0, 1);
    printf(" \"normals\": [");
    for (int i = 0; i < 9; i++) printf("%s%.17g", i ? ", " : "", nd(g3));   // odd count: exercises the cached value
    printf("],\n");
  }
  rng.seed(0);  // UWBNetwork.cpp:19
  printf(" \"noise_std\": %.17g, \"outlier_prob\": %.17g, \"outlier_std\": %.17g,\n", addNoiseStdDev, outlierProbability, outlierStdDev);
  printf(" \"transactions\": [");
  for (int k = 0; k < n; k++) {
    // a fixed, known geometry: the true range of transaction k is 1 + k/8 metres
    const double trueRange = 1.0 + k / 8.0;
    float range;
    int outlier;
    if (distUniform(rng) < outlierProbability) {   // :67
      range = distNormal(rng) * outlierStdDev;     // :68
      outlier = 1;
    } else {
      double measNoise = distNormal(rng) * addNoiseStdDev;   // :70
      range = trueRange + measNoise;                          // :71
      outlier = 0;
    }
    printf("%s[%d, %.9g]", k ? ", " : "", outlier, range);
  }
  printf("]}\n");
  return 0;
}
