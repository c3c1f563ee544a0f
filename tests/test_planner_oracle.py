"""RAPPIDS planner: the oracle's pinned parts against the reference's own
stand-alone sources (RootFinder.hpp, SingleAxisTrajectory.{hpp,cpp}: tests/golden/
planner_math_kat.json), candidate sampling against libstdc++, and behavioural checks of the
unpinned search on synthetic depth images.  CPU only."""
import ctypes as C
import json
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def kat(golden_dir):
    return json.load(open(os.path.join(golden_dir, "planner_math_kat.json")))


def test_root_finder_matches_reference_header(ora, kat):
    L = ora.planner_lib()
    x = (C.c_double * 4)()
    assert len(kat["cubic"]) >= 32 and len(kat["quartic"]) >= 32
    for c in kat["cubic"]:
        n = L.ora_solve_cubic(*c["in"], x)
        assert n == c["count"]
        assert list(x)[:3] == c["x"]                       # bit exact (same libm)
    for c in kat["quartic"]:
        for k in range(4):
            x[k] = 0.0
        n = L.ora_solve_quartic(*c["in"], x)
        assert n == c["count"]
        assert list(x)[:n] == c["x"][:n]


def test_single_axis_trajectory_matches_reference_source(ora, kat):
    L = ora.planner_lib()
    for c in kat["axis"]:
        ax = ora.OraAxis()
        ax.p0, ax.v0, ax.a0, ax.pf, ax.vf, ax.af = 0.0, c["v0"], c["a0"], c["pf"], 0.0, 0.0
        L.ora_axis_generate(C.byref(ax), c["tf"])
        assert (ax.a, ax.b, ax.g, ax.cost) == (c["alpha"], c["beta"], c["gamma"], c["cost"])
        lo, hi = C.c_double(), C.c_double()
        L.ora_axis_minmax_acc(C.byref(ax), C.byref(lo), C.byref(hi), c["t1"], c["t2"])
        assert (lo.value, hi.value) == (c["amin"], c["amax"])
        assert L.ora_axis_max_jerk_sq(C.byref(ax), c["t1"], c["t2"]) == c["jmaxsq"]
        assert L.ora_axis_pos(C.byref(ax), c["tf"]) == c["pos_tf"]
        assert L.ora_axis_vel(C.byref(ax), 0.5 * c["tf"]) == c["vel_half"]


def test_candidate_sampling_matches_libstdcxx(ora, afa, kat):
    """std::mt19937 + uniform_real_distribution in the planner's call shape (g++ order)"""
    for key, seed in (("seed0", 0), ("seed1", 20261002)):
        want = np.array(kat["mt19937"][key])
        np.testing.assert_array_equal(ora.planner_samples(seed, 320, 240, len(want)), want)
        # the engine's generator IS libstdc++ (host side): must agree too
        np.testing.assert_array_equal(afa.planner_samples(seed, 320, 240, len(want)), want)


def _scene(afa, ora, seed, **kw):
    img = afa.scenarios.synthetic_depth_image(seed=seed, **kw)
    # Rappids_Simulator geometry: MINIQUAD arm 0.058 -> radii 0.116 / 0.174, 0.5 m (main.cpp:166-169)
    cfg = ora.planner_config(320, 240, 10.0 / 256.0, 160.0, 0.116, 0.174, 0.5)
    return img, cfg


def test_planner_finds_collision_free_trajectory_and_flags_are_monotone(ora, afa):
    img, cfg = _scene(afa, ora, 3)
    samples = ora.planner_samples(0, 320, 240, 400)
    res, flags = ora.planner_run(cfg, img, [0.5, 0, 1.0], [0, 0, 0], [0, 9.81, 0], samples)
    assert res.found == 1 and res.best_index >= 0 and res.n_generated == 400
    # the result bits nest: collision-free implies admissible implies feasible implies low-cost
    for f in flags:
        assert f in (0, 1, 3, 7, 15)
    assert (flags == 15).sum() == res.n_collision_free and flags[res.best_index] == 15
    assert (flags & 1).sum() == res.n_cost_checks and (flags & 2).astype(bool).sum() == res.n_collision_checks
    # costs of successive winners decrease; the winner is the last one
    assert np.flatnonzero(flags == 15)[-1] == res.best_index
    # an independent dense sampled check agrees that the winner is free
    co = np.array([[res.coeffs[q][a] for a in range(3)] for q in range(6)])
    assert ora.planner_lib().ora_planner_sampled_collision(C.byref(cfg), img.ctypes.data, co.ctypes.data, res.tf, 400) == 0
    # end state: at rest at the sampled point
    s = samples[res.best_index]
    end = sum(co[q] * res.tf ** (5 - q) for q in range(6))
    np.testing.assert_allclose(end, [s[2] * (s[0] - 160) / 160, s[2] * (s[1] - 120) / 160, s[2]], rtol=1e-12, atol=1e-12)


def test_planner_is_conservative(ora, afa):
    """everything the pyramid test calls collision-free passes the dense sampled check
    (the reference's own MeasureConservativeness idea, DepthImagePlanner.cpp:972-1002)"""
    L = ora.planner_lib()
    n_free = 0
    for seed in range(6):
        img, cfg = _scene(afa, ora, 100 + seed, n_trunks=8)
        cfg.cost_type = 1
        cfg.cost_vec[0], cfg.cost_vec[1], cfg.cost_vec[2] = 0.0, 0.0, 120.0
        samples = ora.planner_samples(seed, 320, 240, 300)
        # disable the cost pruning by planning each candidate on its own
        for k in range(0, 300, 7):
            res, flags = ora.planner_run(cfg, img, [0.2, -0.1, 0.8], [0, 0, 0], [0, 9.81, 0], samples[k:k + 1])
            if flags[0] == 15:
                n_free += 1
                co = np.array([[res.coeffs[q][a] for a in range(3)] for q in range(6)])
                assert L.ora_planner_sampled_collision(C.byref(cfg), img.ctypes.data, co.ctypes.data, res.tf, 300) == 0
    assert n_free > 20


def test_blocked_scene_and_pyramid_limit(ora, afa):
    img, cfg = _scene(afa, ora, 5)
    wall = np.full_like(img, 30)                    # a wall 1.2 m ahead: nothing to find
    samples = ora.planner_samples(0, 320, 240, 200)
    res, flags = ora.planner_run(cfg, wall, [0, 0, 1.0], [0, 0, 0], [0, 9.81, 0], samples)
    assert res.found == 0 and res.best_index == -1 and (flags & 8).sum() == 0
    cfg.max_pyramids = 1                            # SetMaxNumberOfPyramids
    res1, _ = ora.planner_run(cfg, img, [0.5, 0, 1.0], [0, 0, 0], [0, 9.81, 0], samples)
    assert res1.n_pyramids <= 1
