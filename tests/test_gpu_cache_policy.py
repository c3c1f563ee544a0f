"""afe_set_cache_policy: `nt` hints on the one-step launches' buffer instructions are memory hints only -- every policy
gives the bits of the default one (state, IMU, noise words), with and without the on-device logic, fp32 and fp64, one
stream and two, on grids that are and are not a multiple of eight workgroups (policy 3 remaps workgroups per XCD)."""
import importlib

import numpy as np
import pytest

from tests.scenarios import random_ensemble

afa = importlib.import_module("agri-fly_amd")
pytestmark = pytest.mark.gpu


def _fly(ens, precision, policy, logic, parts, seed_policy):
    with ens.to_engine(precision) as e:
        e.set_step_mode(afa.AFE_STEP_LAUNCH)
        e.set_imu_noise(True, 0.1, 0.2, seed_policy)
        e.set_split_stepping(parts)
        e.set_cache_policy(policy)
        if logic:
            e.set_rates_logic([afa.rates_logic_params_from_type(5)])
            e.set_rates_commands(np.full(ens.data.n, 9.81, np.float32), np.zeros((3, ens.data.n), np.float32))
        for _ in range(7):
            e.step(1000, 1)
        e.step(1000, 4)            # a fused launch in between (default policy there)
        for _ in range(4):
            e.step(1000, 1)
        st = e.get_state()
        g, a = e.get_imu()
        return dict(st, gyro=g, acc=a, rng=e.get_rng_state(), cmd=e.get_motor_cmds())


@pytest.mark.parametrize("precision", [afa.AFE_F32, afa.AFE_F64])
@pytest.mark.parametrize("n", [512 * 9, 64 * 13 + 5])        # 72 workgroups (a multiple of 8) / 14 with a ragged last wave
@pytest.mark.parametrize("logic", [False, True])
def test_every_policy_gives_the_default_bits(precision, n, logic):
    ens = random_ensemble(n, seed=91, type_ids=(5,))
    ens.data.ext_torque = None           # a force, no torque: the configuration the policies are instantiated for
    for seed_policy in (afa.AFE_SEED_DECORRELATED, afa.AFE_SEED_COUNTER):
        ref = _fly(ens, precision, 0, logic, 1, seed_policy)
        for policy in (1, 2, 3, -1):
            for parts in (1, 2):
                got = _fly(ens, precision, policy, logic, parts, seed_policy)
                for k in ref:
                    assert np.array_equal(ref[k], got[k], equal_nan=True), (policy, parts, k)


def test_policy_argument_is_validated():
    ens = random_ensemble(64, seed=1, type_ids=(5,))
    with ens.to_engine(afa.AFE_F32) as e:
        for bad in (-2, 4):
            with pytest.raises(afa.AfeError):
                e.set_cache_policy(bad)


@pytest.mark.parametrize("log2n,expect", [(22, 1), (23, 3)])
def test_automatic_policy_at_the_sizes_it_is_for_gives_the_default_bits(log2n, expect):
    """2^22 vehicles (the state fits the Infinity Cache: inputs and outputs nt) and 2^23 (nothing fits: everything nt, one
    range per XCD) on the bench's own workload, two streams: twenty steps under the automatic policy against policy 0,
    every bit of the state and the IMU samples"""
    import bench
    n = 1 << log2n
    out = []
    for policy in (0, -1):
        e = bench.build_shard(afa, n, 0, n, 0)
        e.set_cache_policy(policy)
        assert e.cache_policy_in_use == (0 if policy == 0 else expect)
        for _ in range(20):
            e.step(1000, 1)
        st = e.get_state(dtype=np.float32)
        g, a = e.get_imu()
        out.append(dict(st, gyro=g, acc=a))
        e.close()
    for k in out[0]:
        assert np.array_equal(out[0][k], out[1][k], equal_nan=True), k
    assert out[0]["pos"].shape[1] == n and np.isfinite(out[0]["vel"]).all()
