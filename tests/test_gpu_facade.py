"""The C++ host facade (include/agrifly/Quadcopter_T.hpp) driven like the
reference's loops drive Simulation::Quadcopter, checked against the oracle put
through the same scenario (a recording logicType with fixed motor commands --
the shape of the survey's `tap` probe).  Needs an MI355X."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_facade")

# Informational anchor (NOT a pin): the trace the survey-stage probe build of the
# unmodified reference printed for exactly this scenario (SURVEY.md App. B `tap`).
SURVEY_TAP_POS = [[0.00099964788733547202, -0.001999295774670944, 1.0004949189436676],
                  [0.0019987612422235003, -0.0039961134100170645, 1.0009844644391153],
                  [0.0029975087793572184, -0.0059893873275107604, 1.0014734260346949]]
SURVEY_TAP_W = [[0.50008526860017721, -0.39989341424977853, 0.29999999999999999],
                [0.49400753851943502, -0.39987922844542068, 0.29787054305376592],
                [0.4879292003447171, -0.39986708573840679, 0.29574108610753186]]


def _oracle_trace(ora, dt_us, period, runs):
    base = ora.params_from_type(5)
    p = ora.params_init(base.mass, list(base.inertia), float(np.float32(58e-3)), [0, 0, 0],
                        base.motor_min_speed, base.motor_max_speed, base.k_thrust, base.k_torque,
                        0, 0, [0.1, 0.1, 0.1])
    b = ora.Batch(1, [p])
    b.pos[:, 0] = [0, 0, 1]
    b.vel[:, 0] = [1, -2, 0.5]
    b.ang_vel[:, 0] = [0.5, -0.4, 0.3]
    q = np.zeros(4)
    ora.lib().ora_rot_from_euler_ypr(0.3, 0.1, -0.2, q.ctypes.data_as(C.POINTER(C.c_double)))
    b.att[:, 0] = q
    wh = np.sqrt(base.mass * 9.81 / (4 * base.k_thrust))
    cmds = np.float32([wh * 1.02, wh * 0.99, wh * 1.01, wh * 0.98])
    dts, ticks = ora.clock_ticks(dt_us * 1e-6, period, runs)
    out, n_runs = [], 0
    for s in range(runs):
        if dts[s] > 0:
            b.step(dts[s], 1, ticks=[ticks[s]])
            if ticks[s]:
                n_runs += 1
                b.motor_cmd[:, 0] = cmds   # logic output takes effect on the NEXT step (:187-189)
        out.append(dict(pos=b.pos[:, 0].copy(), vel=b.vel[:, 0].copy(), att=b.att[:, 0].copy(),
                        ang_vel=b.ang_vel[:, 0].copy(), gyro=b.gyro[:, 0].copy(), acc=b.acc[:, 0].copy(),
                        runs=n_runs))
    return out


@pytest.mark.parametrize("precision,tol", [("f64", 1e-12), ("f32", 1e-5)])
@pytest.mark.parametrize("dt_us,period,runs", [(1000, 0.0005, 4), (2000, 1 / 500, 12), (1000, 1 / 500, 12)])
def test_facade_matches_oracle(ora, precision, tol, dt_us, period, runs):
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", os.path.dirname(EXE)])
    tr = json.loads(subprocess.check_output([EXE, precision, str(dt_us), repr(period), str(runs)]))["trace"]
    want = _oracle_trace(ora, dt_us, period, runs)
    assert len(tr) == runs
    for s, (got, ref) in enumerate(zip(tr, want)):
        assert got["order_ok"] == 1
        assert got["runs"] == ref["runs"], "logic tick count after Run() #%d" % s
        for k in ("pos", "vel", "att", "ang_vel"):
            err = np.max(np.abs(np.array(got[k]) - ref[k]) / np.maximum(np.abs(ref[k]), 1.0))
            assert err <= tol, (s, k, err)
        if ref["runs"]:
            np.testing.assert_allclose(got["gyro"], ref["gyro"], rtol=0, atol=max(tol, 2e-7) * 10)
            np.testing.assert_allclose(got["acc"], ref["acc"], rtol=0, atol=max(tol, 2e-7) * 100)


def test_facade_reproduces_survey_tap_trace():
    """informational anchor: Run() #2..#4 of the 1 ms / 0.5 ms-period scenario"""
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", os.path.dirname(EXE)])
    tr = json.loads(subprocess.check_output([EXE, "f64", "1000", "0.0005", "4"]))["trace"]
    for s in range(3):
        np.testing.assert_allclose(tr[s + 1]["pos"], SURVEY_TAP_POS[s], rtol=1e-13)
        np.testing.assert_allclose(tr[s + 1]["ang_vel"], SURVEY_TAP_W[s], rtol=1e-12)
    assert tr[0]["runs"] == 0 and tr[1]["runs"] == 1   # first Run() is the dt == 0 early return


BASE_EXE = os.path.join(ROOT, "tests", "cpp", "test_base_pointer")


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_facade_through_the_simulationobject6dof_base_pointer(ora, precision):
    """tests/cpp/test_base_pointer.cpp holds two vehicles as std::shared_ptr<Simulation::SimulationObject6DOF>
    (AIFS_ROS/.../Simulator/main.cpp:83) and uses nothing but the base class: the trace of vehicle 0 must be the
    one the derived class gives directly (same scenario as test_facade), setters between Run()s take effect,
    radio / telemetry / IMU virtuals reach the logic, and a ranging exchange over GetRadio() (the radios a
    Simulation::UWBNetwork works on) delivers the range the reference's completion branch would
    (UWBNetwork.cpp:66-71, noise stream seeded 0)."""
    for exe in (EXE, BASE_EXE):
        if not os.path.exists(exe):
            subprocess.check_call(["make", "-C", os.path.dirname(exe)])
    runs = 8
    direct = json.loads(subprocess.check_output([EXE, precision, "1000", "0.0005", str(runs)]))["trace"]
    whole = json.loads(subprocess.check_output([BASE_EXE, precision, "1000", "0.0005", str(runs)]))
    # a vehicle that cannot exist (negative mass) went to the host's error handler (agrifly::SetErrorHandler) instead of
    # ending the process: afe_set_type_table refused it, every later call on that engine was refused as well
    assert whole["handled_errors"] >= 2 and whole["handled_status"] in (1, 5)
    base = whole["trace"]
    assert len(base) == runs
    for s in range(runs):
        for k in ("pos", "vel", "att", "ang_vel"):
            assert base[s][k + "0"] == direct[s][k], (s, k)          # bit for bit the derived-class path
        assert base[s]["gyro"] == direct[s]["gyro"] and base[s]["acc"] == direct[s]["acc"]
        assert base[s]["runs"] == direct[s]["runs"]
        assert base[s]["telemetry"] == [0, 1, base[s]["runs"] % 256]
        assert base[s]["n_radio"] == (1 if s >= 1 else 0) and base[s]["radio_type"] == (5 if s >= 1 else 0)
    # SetVelocity(0) on vehicle 1 after Run #3: the base's getter shows it at once (same trace row), and the next
    # step starts from rest (only gravity / thrust / drag act)
    assert np.linalg.norm(base[1]["vel1"]) > 2.0 and base[2]["vel1"] == [0, 0, 0] and np.linalg.norm(base[3]["vel1"]) < 0.05
    # UWB: request seen at t = 2 ms, completed at t = 4 ms (0.0015 s period), delivered to the logic on the next tick
    assert [b["n_uwb"][0] for b in base] == [0, 0, 0, 0, 0, 1, 1, 1]
    assert [b["n_uwb"][1] for b in base][-1] == 1                    # everyone "hears" the measurement
    assert base[5]["uwb_responder"] == 2
    u = ora.UwbNetwork(0.05, 0.0, 3.0)
    want, outlier = u.range(base[4]["pos0"], base[4]["radio_pos"])
    assert outlier == 0
    assert np.float32(base[5]["uwb_range"]) == want
    assert base[4]["radio_pos"] == base[4]["pos1"]                   # the radio carries the vehicle's true position
