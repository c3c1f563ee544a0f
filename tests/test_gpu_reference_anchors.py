"""The ENGINE in the reference's config-1 loop (tests/test_reference_anchors.py has the story and the oracle's
version): physics, IMU synthesis, noise and the onboard rates logic on the GPU, the reference's offboard chain
(tests/offboard_reference.py) on the host, one launch per 1 ms step.  Needs an MI355X."""
import numpy as np
import pytest

from tests.scenarios import MEASUREMENTS, afa
from tests.test_reference_anchors import ANCHOR_10S, ANCHOR_10S_THRUST0, ANCHOR_1S, fly_reference_loop

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", [afa.AFE_F64, afa.AFE_F32])
def test_engine_in_the_reference_loop_lands_on_the_reference_numbers(precision):
    p = afa.params_from_type(5)
    with afa.Ensemble(1, precision=precision) as e:
        e.set_type_table([p])
        e.set_logic_period(1 / 500)
        e.set_imu_noise(True, 0.1, 0.2, afa.AFE_SEED_REFERENCE)
        e.set_rates_logic([afa.rates_logic_params_from_type(5)])

        def step(it):
            if it > 0:                                   # Run() #0 sees dt = 0 and returns
                e.step(1000, 1)

        def pose():
            st = e.get_state()
            return st["pos"][:, 0], st["att"][:, 0]

        out = fly_reference_loop(step, pose, e.set_rates_commands, 10.0)
        thrust0 = p.prop_thrust_from_speed_sqr * float(e.get_state()["motor_speed"][0, 0]) ** 2
    err1, err10 = np.abs(out[1.0] - ANCHOR_1S), np.abs(out[10.0] - ANCHOR_10S)
    tag = "f64" if precision == afa.AFE_F64 else "f32"
    MEASUREMENTS["config1_vs_reference_anchors_" + tag] = {
        "pos_1s": [float(x) for x in out[1.0]], "abs_err_1s": [float(x) for x in err1],
        "pos_10s": [float(x) for x in out[10.0]], "abs_err_10s": [float(x) for x in err10], "thrust0_10s": thrust0}
    if precision == afa.AFE_F64:
        # like the oracle: every digit the survey printed, at 1 s and after 10 000 steps / 1 000 radio commands
        assert ["%.9g" % x for x in out[1.0]] == ["%.9g" % x for x in ANCHOR_1S], err1
        assert ["%.9g" % x for x in out[10.0]] == ["%.9g" % x for x in ANCHOR_10S], err10
        assert "%.6g" % thrust0 == "%.6g" % ANCHOR_10S_THRUST0
    else:
        # fp32 state: 1e-5 relative of the 2.4 m climbed, and the hover point to the millimetre the gyro noise
        # (identical stream, quantised commands) leaves open
        assert err1.max() < 5e-5, err1
        assert err10.max() < 2e-3 and err10[2] < 2e-5, err10
