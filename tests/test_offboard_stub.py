"""The test-side offboard loop (tests/offboard_stub.py) and the config-0 closed
loop on the oracle alone.  CPU only."""
import numpy as np

from tests.offboard_stub import OffboardHover, radio_quantise


def test_radio_quantisation():
    """RadioTypes.hpp:73-116: 16-bit, +-35, saturating, NaN -> minimum"""
    v = np.array([0.0, 9.81, -9.81, 34.999, 35.0, 40.0, -35.0, -40.0, np.nan], np.float32)
    q = radio_quantise(v, 35)
    lsb = 35.0 / 32768
    assert abs(q[0]) < 1e-9 and np.all(np.abs(q[1:4] - v[1:4]) <= 1.5 * lsb)   # negatives truncate toward zero
    assert q[4] == q[5] == np.float32(35 * 32767 / 32768)
    assert q[6] == q[7] == q[8] == -35.0
    # int() truncates toward zero: the +0.5 rounds positives to nearest, negatives up
    assert radio_quantise(np.float32(-lsb * 0.75), 35) == 0.0
    assert radio_quantise(np.float32(lsb * 0.75), 35) == np.float32(lsb)


def test_controller_signs():
    c = OffboardHover(3)
    pos = np.array([[0, 1.0, 0], [0, 0, 0], [3.5, 3.5, 0.0]], np.float32)      # at goal / +x of goal / below goal
    vel = np.zeros((3, 3), np.float32)
    att = np.array([[1, 1, 1], [0, 0, 0], [0, 0, 0], [0, 0, 0]], np.float32)
    thrust, w = c.controller(pos, vel, att)
    assert abs(thrust[0] - 9.81) < 1e-5 and np.abs(w[:, 0]).max() < 1e-6
    assert w[1, 1] < 0        # +x position error: pitch nose down... negative rotation about y tilts thrust to -x
    assert thrust[2] > 15.0   # 3.5 m below the goal: climbs


def test_config0_hover_on_the_oracle(ora):
    """BASELINE config 0: one MINIQUAD, at rest on the ground, hover set-point
    (0, 0, 3.5), dt = 1 ms, onboard logic 500 Hz, offboard 100 Hz, 30 ms radio
    delay; 6 s here (the 10 s run is in the GPU suite)."""
    from tests.closed_loop import fly_oracle
    b, log = fly_oracle(ora, 1, 6.0)
    assert abs(b.pos[2, 0] - 3.5) < 0.05
    assert np.abs(b.pos[:2, 0]).max() < 0.05 and np.abs(b.vel[:, 0]).max() < 0.1
    assert b.att[0, 0] > 0.999
    z = log[:, 2, 0]
    assert z[0] == 0.0 and z.max() < 3.9            # leaves the ground, little overshoot
    assert np.all(log[:3, 2, 0] == 0.0)             # sits on the ground until the first delayed command
