"""The rows of SURVEY.md section 8 working together: an ensemble flies the RAPPIDS loop of
Simulator/Rappids_Simulator/main.cpp through a procedural orchard -- physics + IMU + onboard rates
logic (a, f1), uplink quantisation and delay (f2), depth camera (f4) and planner (f3) on the GPU,
depth images never leaving HBM; tests/orchard_flight.py is the offboard side.  This is a system
check (does it fly, does it stay clear of the trees, is it reproducible), not a parity test: every
component has its own bit-level test against its checker.  Needs an MI355X."""
import numpy as np
import pytest

from tests.orchard_flight import fly_orchard
from tests.scenarios import afa

pytestmark = pytest.mark.gpu


def test_ensemble_flies_through_the_orchard_without_touching_a_tree():
    log = fly_orchard(afa, n=48, seconds=8.0, seed=0)
    pos = log["pos"]
    assert np.isfinite(pos).all() and np.isfinite(log["vel"]).all()
    advance = pos[-1, 0] - log["pos0"][0]
    print("\norchard flight: %d vehicles, x advance %.1f..%.1f m in 8 s, min trunk clearance %.3f m, "
          "min canopy level %.2f, plans found %.0f %%, planner %.1f ms / render %.2f ms per frame"
          % (pos.shape[2], advance.min(), advance.max(), log["trunk"].min(), log["canopy"].min(),
             100 * log["found"].mean(), log["plan_ms"].mean(), log["render_ms"].mean()))
    assert advance.min() > 10.0                       # every vehicle made its way east
    assert pos[:, 2].min() > 0.1                      # nobody on the ground
    assert log["trunk"].min() > 0.116                 # physicalVehicleRadius = 2 * armLength (main.cpp:167)
    assert log["canopy"].min() > 1.0                  # outside every canopy ellipsoid
    assert log["found"].mean() > 0.9 and log["planned"].all()
    assert np.sqrt((log["vel"] ** 2).sum(1)).max() < 5.0   # DepthImagePlanner's velocity limit (DIP.cpp:48)


def test_flight_is_reproducible():
    a = fly_orchard(afa, n=16, seconds=2.5, seed=3, rows=4, cols=6)
    b = fly_orchard(afa, n=16, seconds=2.5, seed=3, rows=4, cols=6)
    np.testing.assert_array_equal(a["pos"], b["pos"])
    np.testing.assert_array_equal(a["found"], b["found"])
