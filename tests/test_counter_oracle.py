"""The checker's counter-based generator (oracle/agrifly_oracle_counter.c): Philox4x32-10 against the published
known-answer vectors of the Random123 library (Salmon et al., SC'11 -- the third-party algorithm behind the engine's
AFE_SEED_COUNTER policy and its gust process; the reference itself has neither, see the header), the Box-Muller stage
against a numpy restatement, and the statistics of what comes out."""
import numpy as np

from oracle import oracle_py as ora

# Random123 v1.14, examples/kat_vectors: "philox4x32 10 <ctr x4> <key x2> <expected x4>"
KAT = [
    ([0x00000000] * 4, [0x00000000] * 2, [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0], [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
]


def test_philox4x32_10_known_answers():
    for ctr, key, want in KAT:
        assert ora.philox4x32_10(ctr, key) == want


def test_block_addressing_and_box_muller():
    seed, index, tick = 0x123456789abcdef0, 0x0000_1234_5678_9abc, 0x1_0000_0007
    key = [seed & 0xffffffff, seed >> 32]
    z = ora.imu_normals(seed, index, tick)
    got = []
    for block in (0, 1):
        ctr = [index & 0xffffffff, ((index >> 32) & 0xffff) | (1 << 16) | (block << 24), tick & 0xffffffff, tick >> 32]
        w = ora.philox4x32_10(ctr, key)
        for k in (0, 2):
            u_r = ((w[k] >> 9) + 0.5) / 2.0 ** 23
            u_a = (w[k + 1] >> 8) / 2.0 ** 24
            r = np.sqrt(-2.0 * np.log(u_r))
            got += [r * np.cos(2 * np.pi * u_a), r * np.sin(2 * np.pi * u_a)]
    assert np.allclose(z, got[:6], rtol=0, atol=1e-15)
    # gusts: stream 2, block 0, scaled by the vehicle's sigma
    f = ora.gust_force(seed, 300, 1001, 42, 0.5)
    ctr = [300, 2 << 16, 42, 0]
    w = ora.philox4x32_10(ctr, key)
    u_r, u_a = ((w[0] >> 9) + 0.5) / 2.0 ** 23, (w[1] >> 8) / 2.0 ** 24
    assert abs(f[0] - 0.5 * 300 / 1000 * np.sqrt(-2 * np.log(u_r)) * np.cos(2 * np.pi * u_a)) < 1e-15


def test_the_normals_are_normal_and_independent_across_vehicles_ticks_and_axes():
    z = np.array([ora.imu_normals(7, i, t) for i in range(400) for t in range(50)])      # 20 000 x 6
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01
    assert np.abs(z).max() < 5.7                                                          # 23-bit radius uniform: |z| <= 5.65
    c = np.corrcoef(z.T)
    assert np.abs(c - np.eye(6)).max() < 0.03
    per_vehicle = z.reshape(400, 50, 6)
    assert abs(np.corrcoef(per_vehicle[:-1, :, 0].ravel(), per_vehicle[1:, :, 0].ravel())[0, 1]) < 0.03   # neighbours in index
    assert abs(np.corrcoef(per_vehicle[:, :-1, 0].ravel(), per_vehicle[:, 1:, 0].ravel())[0, 1]) < 0.03   # neighbours in time
    kurt = ((z - z.mean()) ** 4).mean() / z.var() ** 2
    assert abs(kurt - 3.0) < 0.1


def test_step_batch_counter_only_changes_the_noise_and_the_force():
    """with sigma = 0 and no gusts the counter driver is ora_step_batch bit for bit; with noise the libstdc++ word
    does not move and the samples differ from the noise-free ones by sigma * z exactly as defined"""
    n = 16
    rng = np.random.default_rng(1)
    p = ora.params_from_type(5)

    def batch():
        b = ora.Batch(n, [p])
        r = np.random.default_rng(2)
        b.pos[:] = r.normal(0, 1, (3, n)); b.pos[2] += 5
        b.vel[:] = r.normal(0, 1, (3, n))
        q = r.normal(0, 1, (4, n)); b.att[:] = q / np.linalg.norm(q, axis=0)
        b.ang_vel[:] = r.normal(0, 1, (3, n))
        b.motor_cmd[:] = 2800 + r.normal(0, 30, (4, n))
        b.ext_force[:] = r.normal(0, 0.1, (3, n))
        b.rng[:] = 1 + np.arange(n)
        return b

    ticks = np.array([0, 1, 0, 1, 0, 1], np.uint8)
    a, b = batch(), batch()
    a.step(1e-3, 6, ticks=ticks)
    ora.step_counter(b, 1000, 6, ticks, counter_noise=False)
    for f in ("pos", "vel", "att", "ang_vel", "gyro", "acc", "rng"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    c = batch()
    ora.step_counter(c, 1000, 6, ticks, counter_noise=True, seed=11, first_global=100, tick_base=40)
    assert np.array_equal(c.rng, 1 + np.arange(n))
    assert np.array_equal(c.pos, a.pos)
    d = batch()
    d.table[0].sigma_gyro = 0.0
    d.table[0].sigma_acc = 0.0
    d.step(1e-3, 6, ticks=ticks)
    for i in range(n):
        z = ora.imu_normals(11, 100 + i, 42)          # the third tick of the call: 40, 41, 42
        want_g = d.gyro[:, i] + np.float32(0.1) * z[:3].astype(np.float32)
        assert np.array_equal(c.gyro[:, i], want_g.astype(np.float32))
