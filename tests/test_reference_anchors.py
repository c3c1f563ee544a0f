"""Config 1 against numbers of the reference itself.

SURVEY.md Appendix B (`drv`) records where the UNMODIFIED reference -- Simulation::Quadcopter with
Onboard::QuadcopterLogic, Offboard::MocapStateEstimator, Offboard::QuadcopterController, the 16-bit radio and the
30 ms CommunicationsDelay, in the loop order of Simulator/Rappids_Simulator/main.cpp (pre-takeoff branch, no
AirSim), g++ -- is after 1 s and after 10 s of config 1 at dt = 1 ms:

    t = 1 s   pos = (-0.00135107814, -0.00294128835, 2.40432118)
    t = 10 s  pos = (-0.00182557336,  0.00179743433, 3.49997392),  motor-0 thrust 0.345444 N

The surveyor compiled it with a minimal Eigen stand-in (the image has no Eigen), so by the rules of this build the
anchors are informational, not a pin.  They are still the only numbers in reach that the reference produced, and the
flight they come from exercises everything the oracle restates on the hot path: Quadcopter_T::Run, Motor::Run, the
clock gate, the IMU synthesis with libstdc++'s noise stream in g++'s draw order, and the onboard rates logic.  Here the
ORACLE is flown in that loop, with the offboard chain restated in tests/offboard_reference.py:

  * after 1 s (1 000 steps, 500 logic ticks, 100 radio commands) AND after 10 s it stands on the reference's position
    to all nine digits the survey printed, and on its hover thrust to all six.

(What it took: the offboard attitude time constants are DERIVED in float in QuadcopterConstants.hpp:225-228 --
(0.04f * 5) * 2 is 0.399999976, one ulp below the 0.4f a restatement would naturally write; with 0.4f every rate
command was an ulp off and one 16-bit radio code differed somewhere in the tenth second: 3e-6 m at 10 s.)
"""
import numpy as np

from tests.offboard_reference import Clock, ReferenceOffboard

ANCHOR_1S = (-0.00135107814, -0.00294128835, 2.40432118)
ANCHOR_10S = (-0.00182557336, 0.00179743433, 3.49997392)
ANCHOR_10S_THRUST0 = 0.345444


def fly_reference_loop(step, get_pose, set_cmd, seconds, dt_us=1000, marks=(1.0, 10.0)):
    """main.cpp:330,391-392,451-476,737-739 around any vehicle: step(it) runs quad->Run() number `it`"""
    clock = Clock()
    off = ReferenceOffboard(clock)
    n_runs = int(round(seconds * 1e6 / dt_us))
    out = {}
    for it in range(n_runs):
        step(it)
        clock.us += dt_us
        pos, att = get_pose()
        msg = off.iterate(pos, att)
        if msg is not None:
            set_cmd(np.atleast_1d(msg[0]), np.asarray(msg[1]).reshape(3, 1))
        for m in marks:
            if it + 1 == int(round(m * 1e6 / dt_us)):
                out[m] = np.array(pos, float)
    return out


def test_oracle_in_the_reference_loop_lands_on_the_reference_numbers(ora):
    period = 1 / 500
    b = ora.Batch(1, [ora.params_from_type(5)])          # id 1 -> QC_TYPE_CF_MINIQUAD, at rest on the ground
    b.rng[:] = 1                                        # default-constructed std::default_random_engine
    cl = ora.ClosedLoopBatch(b, [ora.logic_params_from_type(5, period)], period)
    dts, ticks = ora.clock_ticks(1e-3, period, 10000)

    def step(it):
        if dts[it] > 0:                                  # Run() #0 sees dt = 0 and returns (Quadcopter_T.cpp:88-90)
            cl.step(dts[it], [ticks[it]])

    out = fly_reference_loop(step, lambda: (b.pos[:, 0].copy(), b.att[:, 0].copy()), cl.set_rates_cmd, 10.0)
    assert ["%.9g" % x for x in out[1.0]] == ["%.9g" % x for x in ANCHOR_1S]
    assert ["%.9g" % x for x in out[10.0]] == ["%.9g" % x for x in ANCHOR_10S]
    kf = ora.params_from_type(5).k_thrust
    assert "%.6g" % (kf * b.motor_speed[0, 0] ** 2) == "%.6g" % ANCHOR_10S_THRUST0
