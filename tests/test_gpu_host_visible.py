"""afe_create_host_visible: the state arena in pinned host memory, for small ensembles with the host in the loop of
every step (Simulator/Rappids_Simulator/main.cpp:330-392: `vehicle->Run()` then `GetPosition()` ... every millisecond;
Quadcopter_T.cpp:159-189 with the onboard logic on the host).  Same bits as an engine with a device arena through every
stepping mode, with getters and setters between the steps -- and those leave a resident grid where it is."""
import importlib
import time

import numpy as np
import pytest

from tests.scenarios import random_ensemble

afa = importlib.import_module("agri-fly_amd")
pytestmark = pytest.mark.gpu


def make(n, precision, host_visible, mode, logic=False, seed=11, policy=None):
    ens = random_ensemble(n, seed=seed, with_wrench=True, type_ids=(5,))
    d = ens.data
    e = afa.Ensemble(n, precision=precision, host_visible=host_visible)
    e.set_type_table([afa.params_from_type(d.type_ids[0])])
    e.set_logic_period(1 / 500)
    e.set_imu_noise(True, 0.1, 0.2, afa.AFE_SEED_DECORRELATED if policy is None else policy)
    e.set_state(d.pos, d.vel, d.att, d.ang_vel, d.motor_speed)
    e.set_motor_cmds(d.motor_cmd)
    e.set_external_force(d.ext_force)
    if logic:
        e.set_rates_logic([afa.rates_logic_params_from_type(d.type_ids[0])])
        rng = np.random.default_rng(3)
        e.set_rates_commands(np.full(n, 9.5, np.float32), (0.2 * rng.standard_normal((3, n))).astype(np.float32))
    e.set_step_mode(mode)
    return e, d


def everything(e):
    st = e.get_state()
    gyro, acc = e.get_imu()
    return dict(st, gyro=gyro, acc=acc, rng=e.get_rng_state(), cmd=e.get_motor_cmds(), force=e.get_external_force(),
                time=np.array([e.time_us, e.logic_ticks]))


def assert_same(a, b, what=""):
    xa, xb = everything(a), everything(b)
    for k in xa:
        assert np.array_equal(xa[k], xb[k], equal_nan=True), (what, k)


@pytest.mark.parametrize("precision", [afa.AFE_F32, afa.AFE_F64])
@pytest.mark.parametrize("mode", [afa.AFE_STEP_LAUNCH, afa.AFE_STEP_PERSISTENT, afa.AFE_STEP_RESIDENT])
@pytest.mark.parametrize("n,logic", [(1, False), (777, False), (5000, True)])
def test_host_visible_engine_is_bitwise_the_device_engine(precision, mode, n, logic):
    a, d = make(n, precision, False, afa.AFE_STEP_LAUNCH, logic)
    b, _ = make(n, precision, True, mode, logic)
    rng = np.random.default_rng(n)
    for rnd in range(40):
        k = int(rng.integers(1, 6))
        a.step(1000, k); b.step(1000, k)
        what = rnd % 5
        if what == 0:
            assert_same(a, b, rnd)
        elif what == 1 and not logic:
            cmd = np.clip(d.motor_cmd * (1 + 0.01 * rnd), 0, None).astype(np.float32)
            a.set_motor_cmds(cmd); b.set_motor_cmds(cmd)
        elif what == 2:
            f = d.ext_force * (1 - 0.02 * rnd)
            a.set_external_force(f); b.set_external_force(f)
        elif what == 3:
            # one vehicle's state through the range form, the way the facade's setters do it
            i = int(rng.integers(0, n))
            p = np.array([[1.0 + rnd], [2.0], [3.0]])
            for e in (a, b):
                e.set_state(pos=p, first=i, count=1)
        elif what == 4:
            ga, gb = a.get_imu(), b.get_imu()
            assert np.array_equal(ga[0], gb[0]) and np.array_equal(ga[1], gb[1])
    assert_same(a, b, "end")
    a.close(); b.close()


def test_getters_and_setters_leave_the_resident_grid_where_it_is():
    n = 64
    e, d = make(n, afa.AFE_F32, True, afa.AFE_STEP_PERSISTENT)
    ref, _ = make(n, afa.AFE_F32, False, afa.AFE_STEP_LAUNCH)
    e.step(1000, 1); ref.step(1000, 1)
    kept = 0
    for s in range(400):
        e.step(1000, 1); ref.step(1000, 1)
        st = e.get_state()                        # waits for the step, reads host memory
        g, a = e.get_imu()
        cmd = (d.motor_cmd * (1 + 1e-4 * s)).astype(np.float32)
        e.set_motor_cmds(cmd); ref.set_motor_cmds(cmd)
        kept += bool(e.persistent_running)        # (a grid the interpreter kept waiting > 200 us has left by itself: allowed)
        assert e.steps_completed == s + 2
    assert kept > 200, "getters or setters are parking the resident grid (%d of 400 steps found it resident)" % kept
    assert_same(e, ref)
    e.close(); ref.close()


def test_checkpoints_cross_between_the_two_kinds_of_arena():
    n = 3000
    a, _ = make(n, afa.AFE_F32, False, afa.AFE_STEP_LAUNCH, logic=True)
    b, _ = make(n, afa.AFE_F32, True, afa.AFE_STEP_PERSISTENT, logic=True)
    a.step(1000, 13); b.step(1000, 13)
    blob_a, blob_b = a.save_checkpoint(), b.save_checkpoint()
    a.step(1000, 20); b.step(1000, 20)
    expect = everything(a)
    a.load_checkpoint(blob_b); b.load_checkpoint(blob_a)
    a.step(1000, 20); b.step(1000, 20)
    for e in (a, b):
        got = everything(e)
        for k in expect:
            assert np.array_equal(expect[k], got[k], equal_nan=True), k
    a.close(); b.close()


def test_a_quiet_host_and_a_long_batch():
    """the grid parks itself while the host sleeps; the next getter finds the steps done all the same; more steps than
    the ring holds in one call"""
    n = 200
    a, _ = make(n, afa.AFE_F64, False, afa.AFE_STEP_LAUNCH)
    b, _ = make(n, afa.AFE_F64, True, afa.AFE_STEP_PERSISTENT)
    for k in (3, 1, 5000, 2):
        a.step(1000, k); b.step(1000, k)
        time.sleep(0.005)
        assert_same(a, b, k)
    a.close(); b.close()
