"""AddressSanitizer + UBSan build of the CPU checker (the GPU pool cannot run
sanitizers; SURVEY.md section 5 / 7.1b ask for them on the CPU code)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r'''
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "agrifly_oracle.h"
#include "agrifly_oracle_logic.h"
int main(void) {
  enum { N = 257, STEPS = 400 };
  ora_params table[4];
  int types[4] = {1, 2, 4, 5};
  for (int k = 0; k < 4; k++) if (ora_params_from_type(&table[k], types[k])) return 2;
  double *pos = calloc(3 * N, 8), *vel = calloc(3 * N, 8), *att = calloc(4 * N, 8), *w = calloc(3 * N, 8);
  double *ms = calloc(4 * N, 8), *fe = calloc(3 * N, 8), *te = calloc(3 * N, 8);
  float *cmd = calloc(4 * N, 4), *gyro = calloc(3 * N, 4), *acc = calloc(3 * N, 4);
  uint32_t *rng = calloc(N, 4);
  uint8_t *ty = calloc(N, 1), ticks[STEPS];
  for (int i = 0; i < N; i++) {
    att[i] = 1.0; pos[2 * N + i] = (i % 3) ? 5.0 : 0.0; rng[i] = 1u + i; ty[i] = i % 4;
    for (int m = 0; m < 4; m++) cmd[m * N + i] = 1000.0f + i;
    fe[i] = 0.01 * i;
  }
  for (int s = 0; s < STEPS; s++) ticks[s] = s & 1;
  ora_step_batch(N, STEPS, table, ty, pos, vel, att, w, ms, rng, cmd, fe, te, 1e-3, ticks, gyro, acc);
  ora_logic_params lp; ora_logic_state ls;
  if (ora_logic_params_from_type(&lp, 5, 0.002f)) return 3;
  ora_logic_init(&lp, &ls);
  float wdes[3] = {0.1f, -0.2f, 0.05f};
  ora_logic_set_rates_cmd(&ls, 9.81f, wdes);
  for (int s = 0; s < 1000; s++) { float g[3] = {gyro[0], gyro[N], gyro[2 * N]}; ora_logic_tick(&lp, &ls, g); }
  ora_clock c; ora_clock_init(&c, 0.002); int t;
  for (int s = 0; s < 1000; s++) { ora_clock_run(&c, &t); ora_clock_advance(&c, 1000); }
  double sum = 0; for (int i = 0; i < 3 * N; i++) sum += pos[i];
  printf("ok %.6f %.3f\n", sum, ls.motor_speed_cmd[0]);
  free(pos); free(vel); free(att); free(w); free(ms); free(fe); free(te); free(cmd); free(gyro); free(acc); free(rng); free(ty);
  return 0;
}
'''


def test_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    src = tmp_path / "driver.c"
    src.write_text(DRIVER)
    exe = tmp_path / "driver"
    ora = os.path.join(ROOT, "oracle")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-ffp-contract=off", "-I", ora, str(src), os.path.join(ora, "agrifly_oracle.c"),
                           os.path.join(ora, "agrifly_oracle_logic.c"), "-lm", "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.startswith("ok ")


def test_host_side_engine_code_is_clean_under_ubsan(tmp_path):
    """the engine's pure-host entry points (type table, tick planner) with g++ sanitizers"""
    src = tmp_path / "host.cpp"
    src.write_text(r'''
#include <cstdio>
#include <cstdint>
#include "agrifly_engine.h"
int main() {
  afe_vehicle_params p; afe_rates_logic_params l;
  for (int t = 0; t < 8; t++) { afe_params_from_type(t, &p); afe_rates_logic_params_from_type(t, &l); }
  uint64_t el = 0; uint8_t ticks[4096];
  const uint64_t dts[6] = {0, 1, 500, 1000, 2000, 3333};
  for (int k = 0; k < 6; k++) afe_plan_ticks(0.002, &el, dts[k], 4096, ticks);
  for (unsigned id = 0; id < 64; id++) afe_type_from_id(id);
  std::printf("ok\n");
  return 0;
}
''')
    exe = tmp_path / "host"
    csrc = os.path.join(ROOT, "agri-fly_amd", "csrc")
    # afe_params.cpp has no HIP calls: build it alone with the host compiler and sanitizers
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", "-I", os.path.join(ROOT, "include"), str(src),
                           os.path.join(csrc, "afe_params.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
