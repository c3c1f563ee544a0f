// oracle/ref_rng_probe.cpp -- TEST INFRASTRUCTURE.
// Known-answer generator for the IMU noise stream.  The reference draws its
// noise from libstdc++ (Components/Components/Simulation/Quadcopter_T.hpp:
// 122-123: std::default_random_engine + std::normal_distribution<double>),
// i.e. from a third-party library that IS present in this image, so the
// restated generator in agrifly_oracle.c is pinned against the real thing.
// The second block reproduces the *shape* of the reference's call site
// (Quadcopter_T.cpp:167-169,176-178: three draws as constructor arguments)
// to record the argument evaluation order of this compiler.
//
// usage: rng_probe <n_normals> [seed]      prints one JSON object.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <type_traits>

struct Triple {  // same shape as Vec3f(Real xin, Real yin, Real zin)
  float x, y, z;
  Triple(float xin, float yin, float zin) : x(xin), y(yin), z(zin) {}
};

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 60;
  const bool seeded = argc > 2;
  std::default_random_engine gen;
  if (seeded) gen.seed((unsigned long) atol(argv[2]));
  std::normal_distribution<double> normal(0, 1);

  printf("{\"engine_is_minstd_rand0\": %d, \"min\": %lu, \"max\": %lu,\n",
         (int) std::is_same<std::default_random_engine, std::minstd_rand0>::value,
         (unsigned long) gen.min(), (unsigned long) gen.max());

  {  // raw engine outputs
    std::default_random_engine g2;
    if (seeded) g2.seed((unsigned long) atol(argv[2]));
    printf(" \"raw\": [");
    for (int i = 0; i < 16; i++) printf("%s%lu", i ? ", " : "", (unsigned long) g2());
    printf("],\n");
  }
  {  // generate_canonical<double,53>
    std::default_random_engine g3;
    if (seeded) g3.seed((unsigned long) atol(argv[2]));
    printf(" \"canonical\": [");
    for (int i = 0; i < 16; i++)
      printf("%s%.17g", i ? ", " : "", std::generate_canonical<double, 53>(g3));
    printf("],\n");
  }
  printf(" \"normals\": [");
  for (int i = 0; i < n; i++) printf("%s%.17g", i ? ", " : "", normal(gen));
  printf("],\n");

  // engine word after the n draws (n even => no cached normal pending)
  {
    std::default_random_engine probe = gen;
    printf(" \"next_raw_after\": %lu,\n", (unsigned long) probe());
  }

  // call-site shape: which draw lands in which component
  std::default_random_engine g4;
  std::normal_distribution<double> n4(0, 1);
  Triple gyro = Triple(float(n4(g4)), float(n4(g4)), float(n4(g4)));
  Triple acc = Triple(float(n4(g4)), float(n4(g4)), float(n4(g4)));
  printf(" \"ctor_order_gyro_xyz\": [%.9g, %.9g, %.9g],\n", gyro.x, gyro.y, gyro.z);
  printf(" \"ctor_order_acc_xyz\": [%.9g, %.9g, %.9g]}\n", acc.x, acc.y, acc.z);
  return 0;
}
