"""The restated onboard rates logic (oracle/agrifly_oracle_logic.c) and the
engine's host-side logic constants.  CPU only."""
import ctypes as C
import json
import os

import numpy as np
import pytest


def test_lowpass_matches_reference_header(ora, golden_dir):
    """pinned: the reference's own LowPassFilterSecondOrder.hpp compiled in place"""
    L = ora.logic_lib()
    cases = json.load(open(os.path.join(golden_dir, "lpf_kat.json")))["cases"]
    assert len(cases) >= 3
    for c in cases:
        f = ora.OraLpf2()
        L.ora_lpf2_init(C.byref(f), c["period"], c["cutoff"], 0.0)
        out = [L.ora_lpf2_apply(C.byref(f), x) for x in c["input"]]
        np.testing.assert_array_equal(np.float32(out), np.float32(c["output"]))
        assert np.float32(f.ym1) == np.float32(c["final_value"])


def test_lowpass_dc_gain_is_one(ora):
    L = ora.logic_lib()
    f = ora.OraLpf2()
    L.ora_lpf2_init(C.byref(f), 0.002, 200.0, 0.0)
    y = 0.0
    for _ in range(400):
        y = L.ora_lpf2_apply(C.byref(f), 1.0)
    assert abs(y - 1.0) < 1e-5


def test_logic_tables_agree_between_oracle_and_engine(ora, afa):
    for t in (1, 2, 4, 5):
        o = ora.logic_params_from_type(t, 1 / 500)
        e = afa.rates_logic_params_from_type(t)
        assert o.mass == e.mass and list(o.inertia) == list(e.inertia)
        assert o.tc_xy == e.ang_vel_time_const_xy and o.tc_z == e.ang_vel_time_const_z
        assert o.kf == e.prop_thrust_from_speed_sqr and o.kt == e.prop_torque_from_thrust * e.prop0_spin_dir
        assert o.d == np.float32(e.arm_length) / np.sqrt(np.float32(2.0))
        assert o.max_thrust == e.max_thrust_per_propeller and o.min_thrust == e.min_thrust_per_propeller
        want = e.max_cmd_total_thrust if e.max_cmd_total_thrust >= 0 else np.float32(4) * e.max_thrust_per_propeller * np.float32(0.8)
        assert o.max_cmd_total_thrust == np.float32(want)
        assert e.gyro_lowpass_cutoff == 200.0
    with pytest.raises(afa.AfeError):
        afa.rates_logic_params_from_type(3)


def test_hover_command_gives_hover_speeds(ora):
    """thrust_norm = g, zero rates, zero gyro -> four equal speeds with k_f w^2 = m g / 4"""
    L = ora.logic_lib()
    p = ora.logic_params_from_type(5, 1 / 500)
    s = ora.OraLogicState()
    L.ora_logic_init(C.byref(p), C.byref(s))
    g0 = (C.c_float * 3)(0, 0, 0)
    L.ora_logic_tick(C.byref(p), C.byref(s), g0)          # IDLE: zero commands
    assert list(s.motor_speed_cmd) == [0, 0, 0, 0] and s.imu_initialized == 1
    L.ora_logic_set_rates_cmd(C.byref(s), 9.81, (C.c_float * 3)(0, 0, 0))
    L.ora_logic_tick(C.byref(p), C.byref(s), g0)
    w = np.array(list(s.motor_speed_cmd))
    assert np.all(w == w[0])
    assert w[0] == pytest.approx(np.sqrt(0.142 * 9.81 / (4 * 4.32e-8)), rel=1e-6)


def test_rate_error_produces_restoring_torque(ora):
    """positive roll-rate error -> more thrust on the +y side (motors 2,3), less on -y (0,1)"""
    L = ora.logic_lib()
    p = ora.logic_params_from_type(5, 1 / 500)
    s = ora.OraLogicState()
    L.ora_logic_init(C.byref(p), C.byref(s))
    L.ora_logic_set_rates_cmd(C.byref(s), 9.81, (C.c_float * 3)(1.0, 0, 0))
    g0 = (C.c_float * 3)(0, 0, 0)
    L.ora_logic_tick(C.byref(p), C.byref(s), g0)
    L.ora_logic_tick(C.byref(p), C.byref(s), g0)
    F = list(s.motor_force_cmd)
    assert F[2] > F[1] and F[3] > F[0] and F[2] == pytest.approx(F[3]) and F[0] == pytest.approx(F[1])
    # torque about x recovered from the forces: d*(F2+F3-F0-F1) = I_xx * (1 rad/s / tc_xy)
    tx = p.d * (F[2] + F[3] - F[0] - F[1])
    assert tx == pytest.approx(p.inertia[0] * 1.0 / p.tc_xy, rel=1e-4)
    # yaw: positive yaw-rate error speeds up the -z spinning props (1,3)
    s2 = ora.OraLogicState()
    L.ora_logic_init(C.byref(p), C.byref(s2))
    L.ora_logic_set_rates_cmd(C.byref(s2), 9.81, (C.c_float * 3)(0, 0, 1.0))
    L.ora_logic_tick(C.byref(p), C.byref(s2), g0)
    F = list(s2.motor_force_cmd)
    assert F[1] > F[0] and F[3] > F[2]


def test_closed_loop_hover_is_stable_in_the_oracle(ora):
    """physics + IMU noise + restated logic: hover thrust, zero rates, 1 s"""
    n = 4
    b = ora.Batch(n, [ora.params_from_type(5)])
    b.pos[2] = 3.5
    b.rng[:] = 1 + np.arange(n)
    lp = [ora.logic_params_from_type(5, 1 / 500)]
    cl = ora.ClosedLoopBatch(b, lp, 1 / 500)
    cl.set_rates_cmd(np.full(n, 9.81, np.float32), np.zeros((3, n), np.float32))
    _, ticks = ora.clock_ticks(1e-3, 1 / 500, 1001)
    cl.step(1e-3, ticks[1:])
    assert np.all(np.abs(b.ang_vel) < 1.0)
    assert np.all(np.abs(np.linalg.norm(b.att, axis=0) - 1) < 1e-9)
    assert np.all(b.att[0] > 0.99)            # still upright: rates loop rejects the gyro noise
    # open-loop in z: the first tick happens 2 steps in, so it sags a little, then hovers
    assert np.all(np.abs(b.vel[2]) < 0.2)
