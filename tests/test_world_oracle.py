"""CPU job: the shared-world checker (oracle/agrifly_oracle_world.c) against its pins, and the
host-only half of the product's UWB network (the noise stream) against the same fixture.

uwb_kat.json comes from oracle/_ref/uwb_probe: libstdc++'s std::mt19937 /
uniform_real_distribution / normal_distribution -- the classes the reference's UWBNetwork.cpp:4-6
instantiates -- run through the statement sequence of UWBNetwork::Run's completion branch (:66-71)."""
import json
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def kat(golden_dir):
    return json.load(open(os.path.join(golden_dir, "uwb_kat.json")))["cases"]


def test_oracle_mt19937_canonical_and_normal_match_libstdcxx(ora, kat):
    c = kat[0]
    u = ora.UwbNetwork()
    assert [u.raw() for _ in range(8)] == c["raw"]
    u = ora.UwbNetwork()
    assert [u.canonical() for _ in range(8)] == c["canonical"]          # bit-exact doubles
    u = ora.UwbNetwork()
    got = np.array([u.normal() for _ in range(9)])
    np.testing.assert_allclose(got, c["normals"], rtol=4e-16, atol=0)   # log / sqrt of the same libm


def test_oracle_transactions_match_the_reference_call_sequence(ora, kat):
    for c in kat:
        u = ora.UwbNetwork(c["noise_std"], c["outlier_prob"], c["outlier_std"])
        for k, (outlier, rng) in enumerate(c["transactions"]):
            true_range = 1.0 + k / 8.0
            r, o = u.range([true_range, 0.0, 0.0], [0.0, 0.0, 0.0])
            assert o == outlier, "transaction %d: outlier decision differs" % k
            assert r == np.float32(rng), "transaction %d: %r vs %r" % (k, r, rng)


def test_product_noise_stream_matches_the_same_fixture(afa, kat):
    """afe_uwb_draw is host-only (libstdc++ through the engine library): no GPU needed"""
    for c in kat:
        net = afa.UwbNetwork(c["noise_std"], c["outlier_prob"], c["outlier_std"])
        noise, out = net.draw(len(c["transactions"]))
        for k, (outlier, rng) in enumerate(c["transactions"]):
            assert out[k] == outlier
            want = np.float32(rng)
            got = np.float32(noise[k]) if outlier else np.float32((1.0 + k / 8.0) + noise[k])
            assert got == want, "transaction %d" % k
        net.close()


def test_product_and_oracle_streams_agree_across_calls(afa, ora):
    """the cached second normal and the generator persist from one batch to the next"""
    net = afa.UwbNetwork(0.1, 0.2, 5.0)
    u = ora.UwbNetwork(0.1, 0.2, 5.0)
    for n in (1, 1, 3, 7, 2):
        noise, out = net.draw(n)
        for k in range(n):
            w, o = u.draw()
            assert o == out[k] and w == noise[k]
    net.close()


def test_nearest_neighbour_definition(ora):
    rng = np.random.default_rng(5)
    xyz = rng.uniform(-3, 3, (3, 300)).astype(np.float32)
    xyz[:, 17] = xyz[:, 4]            # a coincident pair: distance 0, lowest index wins
    xyz[:, 250] = xyz[:, 4]
    xyz[0, 99] = np.nan               # a diverged vehicle neither finds nor is found
    d, i = ora.nearest_neighbour(xyz)
    for k in (0, 4, 17, 250, 123):
        diff = xyz - xyz[:, k:k + 1]
        d2 = (diff[0] * diff[0] + diff[1] * diff[1]) + diff[2] * diff[2]
        d2[k] = np.inf
        d2[np.isnan(d2)] = np.inf
        j = int(np.argmin(d2))
        assert i[k] == j and d[k] == d2[j]
    assert i[4] == 17 and i[17] == 4 and i[250] == 4 and d[4] == 0.0
    assert i[99] == -1 and d[99] == np.float32(3.4e38)
    assert 99 not in i
    d2, i2 = ora.nearest_neighbour(xyz, first=100, count=50)      # a shard's slice
    assert np.array_equal(d2, d[100:150]) and np.array_equal(i2, i[100:150])
