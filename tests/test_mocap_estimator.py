"""The headless program's C++ estimator (agri-fly_amd/cli/mocap_estimator.hpp) against the test side's Python
restatement (tests/offboard_reference.py) on a synthetic flight -- irregular measurement times, commands announced
through the 30 ms pipe, a jump that the 6-sigma gate rejects ten times before the forced reset -- compared BIT FOR
BIT, and the C++ side built with AddressSanitizer + UBSan.  (Both restate Offboard::MocapStateEstimator; flown around
the engine they reproduce the reference's config-1 positions, tests/test_gpu_headless.py.)  CPU only."""
import math
import os
import subprocess

from tests.offboard_reference import Clock, MocapStateEstimator, q_from_rotvec, q_mul

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r'''
#include <cstdio>
#include "mocap_estimator.hpp"
static Vec3d path(double t, bool jumped) { return Vec3d(0.3 * t + (jumped ? 4.0 : 0.0), 0.1 * t * t, 1.0 + 0.5 * t); }
int main() {
  ManualTimer clock;
  agrifly_cli::MocapEstimator est(&clock, 0.03);
  Rotationd att = Rotationd::Identity();
  unsigned long long now = 0;
  for (int it = 0; it < 1500; it++) {
    const unsigned long long step = 700 + 37 * (unsigned long long)(it % 11);      // irregular loop period, microseconds
    clock.AdvanceMicroSeconds(step);
    now += step;
    const double t = now * 1e-6;
    att = att * Rotationd::FromRotationVector(Vec3d(0.2, -0.1, 0.4) * (step * 1e-6));
    if (it % 5 == 0) est.Measure(path(t, it >= 900), att);
    if (it % 9 == 0) est.Announce(Vec3d(0.2, -0.1, 0.4 + 0.001 * it), Vec3d(0.0, 0.2, 0.01 * (it % 7)));
    if (it % 3 == 0) {
      const agrifly_cli::Estimate e = est.Predict(0.03);
      std::printf("%.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %u\n", e.pos.x, e.pos.y, e.pos.z,
                  e.vel.x, e.vel.y, e.vel.z, e.att[0], e.att[1], e.att[2], e.att[3], e.angVel.x, e.angVel.y, e.angVel.z, est.Rejected());
    }
  }
  return 0;
}
'''


def _python_side():
    clock = Clock()
    est = MocapStateEstimator(clock, 0.03)
    att = (1.0, 0.0, 0.0, 0.0)
    rows = []
    for it in range(1500):
        step = 700 + 37 * (it % 11)
        clock.us += step
        t = clock.us * 1e-6
        att = q_mul(att, q_from_rotvec(tuple(c * (step * 1e-6) for c in (0.2, -0.1, 0.4))))
        if it % 5 == 0:
            est.update((0.3 * t + (4.0 if it >= 900 else 0.0), 0.1 * t * t, 1.0 + 0.5 * t), att)
        if it % 9 == 0:
            est.set_predicted((0.2, -0.1, 0.4 + 0.001 * it), (0.0, 0.2, 0.01 * (it % 7)))
        if it % 3 == 0:
            p, v, q, w = est.prediction(0.03)
            rows.append(tuple(p) + tuple(v) + tuple(q) + tuple(w) + (est.n_rejected,))
    return rows


def test_cpp_estimator_equals_the_python_restatement_bit_for_bit(tmp_path):
    src = tmp_path / "driver.cpp"
    src.write_text(DRIVER)
    exe = tmp_path / "driver"
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "agri-fly_amd", "cli"),
                           str(src), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert out.returncode == 0, out.stderr[-2000:]
    cpp = [tuple(float(x) for x in line.split()) for line in out.stdout.strip().split("\n")]
    py = _python_side()
    assert len(cpp) == len(py) == 500
    for k, (a, b) in enumerate(zip(cpp, py)):
        assert all((x == y) or (math.isnan(x) and math.isnan(y)) for x, y in zip(a, b)), (k, a, b)
    assert py[-1][-1] == 10                    # the jump: ten rejections, then the reset took it
    assert abs(py[-1][0] - (0.3 * 1.3 + 4.0)) < 0.5


CONTROLLER_DRIVER = r'''
#include <cstdio>
#include <cstdlib>
#include "hover_controller.hpp"
int main(int argc, char **argv) {
  agrifly_cli::HoverController ctrl;
  double in[10];
  while (std::scanf("%lf %lf %lf %lf %lf %lf %lf %lf %lf %lf", in, in + 1, in + 2, in + 3, in + 4, in + 5, in + 6, in + 7, in + 8, in + 9) == 10) {
    Vec3d w; double thr;
    ctrl.Run(Vec3d(in[0], in[1], in[2]), Vec3d(in[3], in[4], in[5]), Rotationd(in[6], in[7], in[8], in[9]), Vec3d(0, 0, 3.5),
             Vec3d(0, 0, 0), Vec3d(0, 0, 0), 0.0, w, thr);
    std::printf("%.9g %.9g %.9g %.9g\n", thr, w.x, w.y, w.z);
  }
  return 0;
}
'''


def test_cpp_controller_equals_the_numpy_restatement_bit_for_bit(tmp_path):
    """agri-fly_amd/cli/hover_controller.hpp against tests/offboard_stub.py (both restate QuadcopterController::Run in
    float): 2 000 random states from hover-like to violently tilted and far away (thrust saturation, the tilt limit,
    the small-angle branches), every float of every answer equal."""
    import numpy as np
    from tests.offboard_stub import OffboardHover
    src = tmp_path / "ctrl.cpp"
    src.write_text(CONTROLLER_DRIVER)
    exe = tmp_path / "ctrl"
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "agri-fly_amd", "cli"),
                           str(src), "-o", str(exe)])
    rng = np.random.default_rng(12)
    n = 2000
    scale = np.where(np.arange(n) % 4 == 0, 30.0, 1.0)
    pos = rng.normal(0, 2.0, (3, n)) * scale + np.array([[0], [0], [3.5]])
    vel = rng.normal(0, 1.0, (3, n)) * scale
    q = rng.normal(size=(4, n)) * np.array([[1.0], [0.3], [0.3], [0.3]])
    q[:, ::7] = np.array([[1.0], [0.0], [0.0], [0.0]]) + rng.normal(0, 1e-4, (4, len(q[0, ::7])))
    q /= np.linalg.norm(q, axis=0)
    pos[:, 5], vel[:, 5], q[:, 5] = [0, 0, 3.5], [0, 0, 0], [1, 0, 0, 0]           # exactly at the set point, level
    text = "".join(" ".join("%.17g" % x for x in np.concatenate([pos[:, i], vel[:, i], q[:, i]])) + "\n" for i in range(n))
    out = subprocess.run([str(exe)], input=text, capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert out.returncode == 0, out.stderr[-2000:]
    cpp = np.array([[float(x) for x in line.split()] for line in out.stdout.strip().split("\n")], np.float32)
    stub = OffboardHover(1)
    mism = 0
    for i in range(n):
        th, w = stub.controller(pos[:, i].reshape(3, 1), vel[:, i].reshape(3, 1), q[:, i].reshape(4, 1))
        got = np.array([th[0], w[0, 0], w[1, 0], w[2, 0]], np.float32)
        mism += not np.array_equal(got, cpp[i])
    assert mism == 0


TRACKING_DRIVER = r'''
#include <cstdio>
#include "hover_controller.hpp"
#include "planned_trajectory.hpp"
int main() {
  agrifly_cli::HoverController ctrl;
  double in[45];
  for (;;) {
    int got = 0;
    for (int k = 0; k < 45; k++) got += std::scanf("%lf", in + k) == 1;
    if (got != 45) break;
    agrifly_cli::PlannedTrajectory tr;
    for (int q = 0; q < 6; q++) for (int a = 0; a < 3; a++) tr.c[q][a] = in[3 * q + a];
    tr.gravity = Vec3d(in[18], in[19], in[20]);
    const double t = in[21];
    const Vec3d p = tr.Position(t), v = tr.Velocity(t), a = tr.Acceleration(t), w = tr.Omega(t, 0.02);
    std::printf("%.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g", p.x, p.y, p.z, v.x, v.y, v.z, a.x, a.y, a.z,
                tr.Thrust(t), w.x, w.y, w.z);
    Vec3d cmdW; double cmdT; Rotationf cmdAtt;
    ctrl.RunTracking(Vec3d(in[22], in[23], in[24]), Vec3d(in[25], in[26], in[27]), Rotationd(in[28], in[29], in[30], in[31]),
                     Vec3d(in[32], in[33], in[34]), Vec3d(in[35], in[36], in[37]), Vec3d(in[38], in[39], in[40]), 0.0, in[41],
                     Vec3d(in[42], in[43], in[44]), cmdW, cmdT, cmdAtt);
    std::printf(" %.9g %.9g %.9g %.9g\n", cmdT, cmdW.x, cmdW.y, cmdW.z);
  }
  return 0;
}
'''


def test_cpp_tracking_equals_the_numpy_restatement(tmp_path):
    """agri-fly_amd/cli/planned_trajectory.hpp and HoverController::RunTracking against the numpy driver of the GPU
    system tests (tests/orchard_flight.py: poly_eval, traj_omega, run_tracking -- the restatements of
    RapidTrajectoryGenerator's evaluation and QuadcopterController::RunTracking the in-loop tests fly with), on
    1 000 random trajectories and states.  The polynomials agree exactly; the angles go through numpy's own
    arccos on one side and glibc's on the other, hence an ulp or two of slack there.  ASan + UBSan on the C++ side."""
    import numpy as np
    from tests.orchard_flight import poly_eval, run_tracking, traj_omega
    src = tmp_path / "track.cpp"
    src.write_text(TRACKING_DRIVER)
    exe = tmp_path / "track"
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "agri-fly_amd", "cli"),
                           str(src), "-o", str(exe)])
    rng = np.random.default_rng(31)
    n = 1000
    coeffs = rng.normal(0, 1.0, (n, 6, 3)) * np.array([0.05, 0.1, 0.3, 0.5, 1.0, 1.0])[None, :, None]
    grav = np.tile(np.array([[0.0], [9.81], [0.0]]), (1, n)) + rng.normal(0, 0.5, (3, n))
    t = rng.uniform(0, 2.0, n)
    pos, vel = rng.normal(0, 2.0, (3, n)), rng.normal(0, 1.0, (3, n))
    q = rng.normal(size=(4, n)) * np.array([[1.0], [0.3], [0.3], [0.3]])
    q /= np.linalg.norm(q, axis=0)
    ref_pos, ref_vel, ref_acc = pos + rng.normal(0, 0.5, (3, n)), rng.normal(0, 1.0, (3, n)), rng.normal(0, 2.0, (3, n))
    ref_thrust, ref_w = rng.uniform(5, 15, n), rng.normal(0, 1.0, (3, n))
    ref_acc[:, 3] = -np.array([0.0, 0.0, 9.81]) - (ref_pos[:, 3] - pos[:, 3]) * 4 - (ref_vel[:, 3] - vel[:, 3]) * 2.8   # a vanishing proper acceleration
    text = ""
    for i in range(n):
        vals = np.concatenate([coeffs[i].ravel(), grav[:, i], [t[i]], pos[:, i], vel[:, i], q[:, i], ref_pos[:, i], ref_vel[:, i], ref_acc[:, i],
                               [ref_thrust[i]], ref_w[:, i]])
        text += " ".join("%.17g" % x for x in vals) + "\n"
    out = subprocess.run([str(exe)], input=text, capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert out.returncode == 0, out.stderr[-2000:]
    cpp = np.array([[float(x) for x in line.split()] for line in out.stdout.strip().split("\n")])
    assert cpp.shape == (n, 17)
    p, v, a = poly_eval(coeffs, t)
    assert np.array_equal(cpp[:, 0:3], p.T) and np.array_equal(cpp[:, 3:6], v.T) and np.array_equal(cpp[:, 6:9], a.T)
    thrust = np.sqrt(((a - grav) ** 2).sum(0))
    np.testing.assert_allclose(cpp[:, 9], thrust, rtol=4e-16)
    np.testing.assert_allclose(cpp[:, 10:13], traj_omega(coeffs, grav, t).T, rtol=1e-9, atol=1e-12)
    th, w = run_tracking(pos, vel, q, ref_pos, ref_vel, ref_acc, ref_thrust, ref_w)
    ok = np.ones(n, bool)
    ok[3] = False                                   # 0 / 0 in the thrust direction: both sides say NaN, in their own way
    # (the thrust: the C++ side adds the float correction to the double feed-forward like the reference, the numpy driver in float)
    np.testing.assert_allclose(cpp[ok, 13], th[ok], rtol=3e-7, atol=2e-6)
    np.testing.assert_allclose(cpp[ok, 14:17], w[:, ok].T, rtol=2e-5, atol=2e-5)
    assert np.median(np.abs(cpp[ok, 14:17] - w[:, ok].T)) < 1e-6      # float ulps of rates of a few rad/s
