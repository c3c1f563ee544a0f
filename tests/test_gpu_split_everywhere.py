"""afe_set_split_stepping defers the moment the engine's stream is ordered after the steps to the next engine call
that touches state -- so every entry point has to do that ordering.  The proof is the rest of the suite: with
AFE_FORCE_SPLIT=1 every engine of a process steps its two halves on two streams (ensembles of 1 024 vehicles and
more), and the tests must not notice.  The whole GPU suite passes that way (run it: AFE_FORCE_SPLIT=1 pytest -m gpu);
this test re-runs the files that lean hardest on ordering -- the closed loops, the C++ facade, the shared-world
exchange with its logical shards and peer copies, checkpoints, the camera and planner reading engine state -- in a
child process with the switch on.  Needs an MI355X."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_suite_does_not_notice_split_stepping():
    if os.environ.get("AFE_FORCE_SPLIT"):
        pytest.skip("already inside a forced-split run")
    files = ["tests/test_gpu_logic.py", "tests/test_gpu_facade.py", "tests/test_gpu_sharedworld.py", "tests/test_gpu_config01.py",
             "tests/test_gpu_orchard_flight.py", "tests/test_gpu_headless.py"]
    env = dict(os.environ, AFE_FORCE_SPLIT="1")
    out = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"] + files, cwd=ROOT, env=env,
                         capture_output=True, text=True)
    tail = "\n".join(out.stdout.strip().split("\n")[-15:])
    assert out.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail


@pytest.mark.parametrize("mode", ["1", "3"])
def test_the_suite_does_not_notice_the_resident_grid(mode):
    """the same argument for afe_set_step_mode: with AFE_FORCE_STEP_MODE every engine of a process starts in
    AFE_STEP_PERSISTENT (1) or AFE_STEP_RESIDENT (3) -- homogeneous ensembles on the engine's own stream are then stepped
    by the resident grid, which every other entry point has to park -- and the files that lean hardest on ordering,
    plus the parity files, must not notice (the whole suite passes this way: AFE_FORCE_STEP_MODE=1 pytest -m gpu)."""
    if os.environ.get("AFE_FORCE_STEP_MODE") or os.environ.get("AFE_FORCE_SPLIT"):
        pytest.skip("already inside a forced run")
    files = ["tests/test_gpu_logic.py", "tests/test_gpu_facade.py", "tests/test_gpu_sharedworld.py", "tests/test_gpu_config01.py",
             "tests/test_gpu_orchard_flight.py", "tests/test_gpu_headless.py", "tests/test_gpu_parity.py"]
    env = dict(os.environ, AFE_FORCE_STEP_MODE=mode)
    out = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"] + files, cwd=ROOT, env=env,
                         capture_output=True, text=True)
    tail = "\n".join(out.stdout.strip().split("\n")[-15:])
    assert out.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail
