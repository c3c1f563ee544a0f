"""On-device onboard rates logic against the oracle's
restated logic in the loop.  Needs an MI355X."""
import numpy as np
import pytest

from tests.scenarios import afa, random_ensemble, rel_err

pytestmark = pytest.mark.gpu


def _closed_loop_pair(n, seed, precision, type_ids=(5, 1, 2, 4)):
    from oracle import oracle_py as ora
    ens = random_ensemble(n, seed=seed, type_ids=type_ids, with_wrench=True, ground_fraction=0.0)
    d = ens.data
    d.pos[2] += 20.0
    d.ang_vel *= 0.3
    # start with motors near hover and modest tilt so the rates loop is in its linear range
    for k, t in enumerate(d.type_ids):
        p = afa.params_from_type(t)
        sel = d.types == k
        d.motor_speed[:, sel] = p.hover_speed * (1 + 0.05 * (d.motor_speed[:, sel] / p.motor_max_speed - 0.5))
    d.motor_cmd[:] = 0
    b = ens.to_oracle_batch()
    b.rng[:] = 1 + np.arange(n)
    lp = [ora.logic_params_from_type(t, 1 / 500) for t in d.type_ids]
    cl = ora.ClosedLoopBatch(b, lp, 1 / 500)
    e = ens.to_engine(precision)
    e.set_imu_noise(True, 0.1, 0.2, afa.AFE_SEED_DECORRELATED)
    e.set_rates_logic([afa.rates_logic_params_from_type(t) for t in d.type_ids])
    rng = np.random.default_rng(seed)
    thrust = (9.81 + rng.normal(0, 0.5, n)).astype(np.float32)
    wdes = rng.normal(0, 0.5, (3, n)).astype(np.float32)
    return ens, b, cl, e, thrust, wdes


@pytest.mark.parametrize("precision,tol", [(afa.AFE_F64, 1e-9), (afa.AFE_F32, 1e-5)])
def test_closed_loop_matches_oracle(precision, tol):
    n, steps = 512, 60
    ens, b, cl, e, thrust, wdes = _closed_loop_pair(n, 31, precision)
    ticks = afa.plan_ticks(1 / 500, 0, 1000, steps)[0]
    with e:
        # phase 1: IDLE (no radio command yet): logic runs, commands stay zero
        e.step(1000, 10)
        cl.step(1e-3, ticks[:10])
        np.testing.assert_array_equal(e.get_motor_cmds(), 0.0)
        # phase 2: rates command arrives
        e.set_rates_commands(thrust, wdes)
        cl.set_rates_cmd(thrust, wdes)
        e.step(1000, steps - 10)        # one fused launch: ticks and command updates inside
        cl.step(1e-3, ticks[10:])
        st = e.get_state()
        cmds = e.get_motor_cmds()
        rng = e.get_rng_state()
    np.testing.assert_array_equal(rng, b.rng)
    from tests.scenarios import record_parity
    for k, ref in dict(pos=b.pos, vel=b.vel, att=b.att, ang_vel=b.ang_vel, motor_speed=b.motor_speed).items():
        err = record_parity("closed loop with the on-device rates logic, 60 steps", precision, k, st[k], ref)
        assert err <= tol, (k, err)
    # motor commands are float in the reference; 1e-5 of ~3e3 rad/s
    assert rel_err(cmds, b.motor_cmd, 1.0) <= max(tol, 1e-6)


def test_logic_is_bit_exact_given_identical_imu():
    """fp64 physics => the float IMU sample is (almost always) the reference's
    bit for bit, and then so are the float motor commands"""
    n = 256
    ens, b, cl, e, thrust, wdes = _closed_loop_pair(n, 32, afa.AFE_F64, type_ids=(5,))
    ticks = afa.plan_ticks(1 / 500, 0, 1000, 9)[0]
    with e:
        e.set_rates_commands(thrust, wdes)
        cl.set_rates_cmd(thrust, wdes)
        e.step(1000, 9)
        cl.step(1e-3, ticks)
        gyro, _ = e.get_imu()
        cmds = e.get_motor_cmds()
    same_imu = np.all(gyro == b.gyro, axis=0)
    assert same_imu.mean() > 0.9
    np.testing.assert_array_equal(cmds[:, same_imu], b.motor_cmd[:, same_imu])


def test_fused_and_single_launches_agree_with_logic_on():
    n = 1000
    ens, b, cl, e1, thrust, wdes = _closed_loop_pair(n, 33, afa.AFE_F32)
    _, _, _, e2, _, _ = _closed_loop_pair(n, 33, afa.AFE_F32)
    with e1, e2:
        for e in (e1, e2):
            e.set_rates_commands(thrust, wdes)
        e1.step(1000, 41)
        for _ in range(41):
            e2.step(1000, 1)
        s1, s2 = e1.get_state(dtype=np.float32), e2.get_state(dtype=np.float32)
        for k in s1:
            np.testing.assert_array_equal(s1[k], s2[k], err_msg=k)
        np.testing.assert_array_equal(e1.get_motor_cmds(), e2.get_motor_cmds())


def test_closed_loop_hover_1m_vehicles_stays_upright():
    """config 2/4 at scale with the loop closed on the GPU: 1 s of flight"""
    n = 1 << 20
    p = afa.params_from_type(5)
    data = afa.scenarios.gust_ensemble(n, p, seed=4, sigma_max=0.05)
    with afa.Ensemble(n) as e:
        e.set_type_table([p])
        e.set_state(data.pos, data.vel, data.att, data.ang_vel, data.motor_speed)
        e.set_external_force(data.ext_force)
        e.set_imu_noise(True, 0.1, 0.2, afa.AFE_SEED_DECORRELATED)
        e.set_rates_logic([afa.rates_logic_params_from_type(5)])
        e.set_rates_commands(np.full(n, 9.81, np.float32), np.zeros((3, n), np.float32))
        e.step(1000, 1000)
        st = e.get_state(dtype=np.float32)
        assert e.logic_ticks == 499
    assert np.isfinite(st["pos"]).all()
    assert np.abs(np.linalg.norm(st["att"], axis=0) - 1).max() < 1e-6
    assert (st["att"][0] > 0.99).all()          # rates loop holds attitude against gyro noise
    assert np.abs(st["ang_vel"]).max() < 2.0


def test_disable_logic_restores_host_commands():
    ens = random_ensemble(300, seed=34)
    with ens.to_engine(afa.AFE_F32) as e:
        with pytest.raises(afa.AfeError):
            e.set_rates_commands(np.zeros(300, np.float32), np.zeros((3, 300), np.float32))
        e.set_rates_logic([afa.rates_logic_params_from_type(t) for t in ens.data.type_ids])
        np.testing.assert_array_equal(e.get_motor_cmds(), 0.0)
        e.set_rates_logic(None)
        e.set_motor_cmds(ens.data.motor_cmd)
        e.step(1000, 4)
        np.testing.assert_array_equal(e.get_motor_cmds(), ens.data.motor_cmd)
        with pytest.raises(afa.AfeError):
            e.set_rates_logic([afa.rates_logic_params_from_type(5)])   # 4 vehicle types need 4 records


def test_radio_packets_drive_the_device_logic():
    """SetCommandRadioMsg for the on-device logic: 23-byte packets in, same
    behaviour as decoded float commands; idle returns the motors to zero"""
    n = 500
    ens, b, cl, e1, thrust, wdes = _closed_loop_pair(n, 35, afa.AFE_F32, type_ids=(5,))
    _, _, _, e2, _, _ = _closed_loop_pair(n, 35, afa.AFE_F32, type_ids=(5,))
    raw = np.stack([afa.radio_create_rates_command(0, thrust[i], wdes[:, i]) for i in range(n)])
    with e1, e2:
        e1.set_commands_from_radio(raw)
        dec = np.array([list(afa.radio_decode(raw[i]).floats)[:4] for i in range(n)], np.float32)
        e2.set_rates_commands(dec[:, 0], dec[:, 1:4].T)
        for e in (e1, e2):
            e.step(1000, 20)
        np.testing.assert_array_equal(e1.get_motor_cmds(), e2.get_motor_cmds())
        s1, s2 = e1.get_state(dtype=np.float32), e2.get_state(dtype=np.float32)
        np.testing.assert_array_equal(s1["ang_vel"], s2["ang_vel"])
        idle = np.zeros((n, 23), np.uint8)
        idle[:, 0] = 6
        e1.set_commands_from_radio(idle)
        e1.step(1000, 4)
        np.testing.assert_array_equal(e1.get_motor_cmds(), 0.0)
        bad = idle.copy()
        bad[3, 0] = 3     # positionCommand needs the host-side logic
        with pytest.raises(afa.AfeError):
            e1.set_commands_from_radio(bad)


def test_a_hundred_seconds_of_closed_loop_flight_stay_sane():
    """1e5 steps (100 s at dt = 1 ms) of 2^18 vehicles with the rates loop closed on the device, per-vehicle noise
    streams and gusts, in fused launches: nothing goes non-finite, the fp32 quaternions stay unit to a rounding,
    body rates stay at the noise floor and nobody tips over (the rates loop alone does not hold position -- the
    gusts push the ensemble downwind -- but it must hold the attitude)."""
    n = 1 << 18
    p = afa.params_from_type(5)
    data = afa.scenarios.gust_ensemble(n, p, seed=11)
    with afa.Ensemble(n) as e:
        e.set_type_table([p])
        e.set_logic_period(1 / 500)
        e.set_imu_noise(True, 0.1, 0.2, afa.AFE_SEED_DECORRELATED)
        e.set_state(data.pos, data.vel, data.att, data.ang_vel, data.motor_speed)
        e.set_external_force(data.ext_force * 0.2)
        e.set_rates_logic([afa.rates_logic_params_from_type(5)])
        e.set_rates_commands(np.full(n, 9.81, np.float32), np.zeros((3, n), np.float32))
        for _ in range(100):
            e.step(1000, 1000)
        st = e.get_state()
        assert e.time_us == 100_000_000 and e.logic_ticks == 49_999
    assert all(np.isfinite(st[k]).all() for k in st)
    q = st["att"]
    assert np.abs(np.linalg.norm(q, axis=0) - 1).max() < 5e-7
    assert (2 * np.arccos(np.clip(np.abs(q[0]), 0, 1))).max() < 0.5
    assert np.linalg.norm(st["ang_vel"], axis=0).max() < 0.3
