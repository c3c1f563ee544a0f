"""Pins of the oracle (and of the engine's host-side clock) against the
committed fixtures generated from the reference's own Timer headers and from
libstdc++ (tests/golden/make_golden.py).  CPU only."""
import ctypes as C
import json
import os

import numpy as np
import pytest


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def test_clock_matches_reference_timer_headers(ora, golden_dir):
    cases = _load(golden_dir, "timer_cadence.json")["cases"]
    assert len(cases) >= 6
    for c in cases:
        dts, ticks = ora.clock_ticks(c["loop_dt"], c["period"], len(c["tick"]))
        assert ticks == c["tick"], (c["loop_dt_expr"], c["period_expr"])
        # dt is bit-identical: (double)(uint64 us * 1e-6), Timer.hpp:36-38
        assert dts == c["dt"], (c["loop_dt_expr"], c["period_expr"])


def test_engine_tick_planner_matches_reference_timer_headers(afa, golden_dir):
    """afe_plan_ticks is the product's own clock; same fixture."""
    for c in _load(golden_dir, "timer_cadence.json")["cases"]:
        n_runs = len(c["tick"])
        # run 0 is the dt == 0 early return; runs 1.. are physics steps
        ticks, _ = afa.plan_ticks(c["period"], 0, c["advance_us"], n_runs - 1)
        assert [0] + ticks.tolist() == c["tick"], (c["loop_dt_expr"], c["period_expr"])


def test_survey_cadence_patterns(ora):
    # SURVEY.md G4 [probe]: cycle counters after each Run()
    def cycles(dt, per, n):
        return np.cumsum(ora.clock_ticks(dt, per, n)[1]).tolist()
    assert cycles(1 / 500, 1 / 500, 6) == [0, 0, 1, 2, 3, 4]
    assert cycles(1 / 1000, 1 / 500, 8) == [0, 0, 0, 1, 1, 2, 2, 3]
    assert cycles(1 / 1000, 1 / 1000, 6) == [0, 0, 1, 2, 3, 4]


def test_communications_delay_fixture_semantics(golden_dir):
    """CommunicationsDelay (reference CommunicationsDelay.hpp:18-33): a message
    enqueued at run s is released when now >= enqueue + uint64(0.03e6) us, one
    message per loop iteration (main.cpp:737-739)."""
    for c in _load(golden_dir, "timer_cadence.json")["cases"]:
        adv = c["advance_us"]
        queue, got = [], []
        now = 0
        for s in range(len(c["tick"])):
            now += adv
            if s % 10 == 0:
                queue.append((now + int(np.uint64(0.03 * 1e6)), s))
            if queue and now >= queue[0][0]:
                got.append(queue.pop(0)[1])
            else:
                got.append(-1)
        assert got == c["radio_delivered"]


def test_rng_matches_libstdcxx(ora, golden_dir):
    kat = _load(golden_dir, "rng_kat.json")
    L = ora.lib()
    for stream in kat["streams"]:
        assert stream["engine_is_minstd_rand0"] == 1
        seed = stream["seed"] % 2147483647 or 1
        s = C.c_uint32(seed)
        raw = [L.ora_minstd_next(C.byref(s)) for _ in range(16)]
        assert raw == stream["raw"]
        s = C.c_uint32(seed)
        can = [L.ora_canonical(C.byref(s)) for _ in range(16)]
        assert can == stream["canonical"]  # bit exact
        s = C.c_uint32(seed)
        a, b = C.c_double(), C.c_double()
        normals = []
        for _ in range(len(stream["normals"]) // 2):
            L.ora_normal_pair(C.byref(s), C.byref(a), C.byref(b))
            normals += [a.value, b.value]
        np.testing.assert_allclose(normals, stream["normals"], rtol=0, atol=0)
        assert L.ora_minstd_next(C.byref(s)) == stream["next_raw_after"]


def test_survey_rng_anchor(golden_dir):
    # SURVEY.md G5 [probe]: first normals of the default engine
    n = _load(golden_dir, "rng_kat.json")["streams"][0]["normals"]
    np.testing.assert_allclose(n[:6], [-0.121965784, -1.08681804, 0.684289944, -1.07518915,
                                       0.0332694764, 0.744835598], rtol=1e-8)


def test_imu_noise_argument_order(ora, golden_dir):
    """g++ evaluates Vec3f(float(n(g)), float(n(g)), float(n(g))) right to left
    (Quadcopter_T.cpp:167-169,176-178): the oracle's IMU must show the same
    component assignment as the compiled call-site shape."""
    stream = _load(golden_dir, "rng_kat.json")["streams"][0]
    p = ora.params_from_type(5)
    b = ora.Batch(1, [p])
    b.pos[2] = 10.0
    b.step(1e-3, 1, ticks=[1])  # at rest, zero motor speed: gyro = noise only
    sg = np.float32(0.1)
    want = (np.float32(stream["ctor_order_gyro_xyz"]) * sg).astype(np.float32)
    np.testing.assert_array_equal(b.gyro[:, 0], want)
    # free fall: proper acceleration is zero, accelerometer = noise only
    sa = np.float32(0.2)
    want_a = (np.float32(stream["ctor_order_acc_xyz"]) * sa).astype(np.float32)
    np.testing.assert_allclose(b.acc[:, 0], want_a, rtol=0, atol=1e-7)
    n = stream["normals"]
    assert stream["ctor_order_gyro_xyz"][2] == pytest.approx(n[0], rel=1e-7)
    assert stream["ctor_order_gyro_xyz"][0] == pytest.approx(n[2], rel=1e-7)
    assert stream["ctor_order_acc_xyz"][2] == pytest.approx(n[3], rel=1e-7)


def test_oracle_regression_vectors(golden_dir):
    """Freezes the restatement (oracle-generated, NOT a reference pin)."""
    from tests.scenarios import random_ensemble
    g = np.load(os.path.join(golden_dir, "oracle_regression.npz"))
    ens = random_ensemble(n=256, seed=int(g["seed"]))
    b = ens.to_oracle_batch()
    b.step(1e-3, 1, ticks=[1])
    for k, a in (("pos", b.pos), ("vel", b.vel), ("att", b.att), ("ang_vel", b.ang_vel),
                 ("motor", b.motor_speed), ("gyro", b.gyro), ("acc", b.acc), ("rng", b.rng)):
        np.testing.assert_array_equal(a, g["s1_" + k], err_msg=k)
    b = ens.to_oracle_batch()
    ticks = np.zeros(100, np.uint8)
    ticks[1::2] = 1
    b.step(1e-3, 100, ticks=ticks)
    for k, a in (("pos", b.pos), ("vel", b.vel), ("att", b.att), ("ang_vel", b.ang_vel),
                 ("motor", b.motor_speed), ("gyro", b.gyro), ("acc", b.acc), ("rng", b.rng)):
        np.testing.assert_array_equal(a, g["s100_" + k], err_msg=k)
