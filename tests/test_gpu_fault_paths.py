"""Failure paths of the resident grid's own AQL queue (round-5 review item 6): the code that only runs when the runtime
refuses something -- hsa_queue_create fails, the kernel descriptor is not found, the code object declares another
kernel-argument size, a parked grid does not come back, a sync request is never answered.  Each is forced in a library
built with -DAFE_DEV_HOOKS (AFE_FAULT=<name>, agri-fly_amd/csrc/afe_host.h; the release build has no such switch) in a
child process.  The contract: fall back to the HIP stream with the launched kernels' bits, or return AFE_ERR_* -- never
hang, never crash -- and say on stderr which way it went.  Needs an MI355X and agri-fly_amd/lib/dev/ (built by
__graft_entry__.build())."""
import os
import subprocess
import sys
import time

import pytest

from tests.scenarios import dev_hooks_env

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STEP_BOTH_WAYS = r'''
import importlib, sys, numpy as np
sys.path.insert(0, %r)
from tests.test_gpu_persistent import make, assert_same
afa = importlib.import_module("agri-fly_amd")
assert afa.library().afe_has_dev_hooks()
a, _ = make(20000, afa.AFE_F32, False)          # launched kernels
b, _ = make(20000, afa.AFE_F32, True)           # resident grid: on the engine's own queue unless something refuses
for k in (1, 20, 1, 7):
    a.step(1000, k); b.step(1000, k)
    b.sync()
assert_same(a, b)
for _ in range(40):                              # and again, one step per call with a getter in between
    a.step(1000, 1); b.step(1000, 1)
    b.get_state(first=0, count=4)
assert_same(a, b)
a.close(); b.close()
print("ok")
''' % ROOT


def _child(code, fault, extra=None, timeout=300):
    env = dev_hooks_env()
    if env is None:
        pytest.skip("no library with -DAFE_DEV_HOOKS (make -C agri-fly_amd/csrc EXTRA=-DAFE_DEV_HOOKS OUT=../lib/dev/libagrifly_engine.so OBJ=../lib/dev/obj)")
    env = dict(env, AFE_PERSIST_AQL="1", **(extra or {}))
    env.pop("AFE_FAULT", None)
    if fault:
        env["AFE_FAULT"] = fault
    t0 = time.perf_counter()
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=timeout)
    return out, time.perf_counter() - t0


def test_without_a_fault_the_grid_runs_on_its_own_queue():
    out, _ = _child(STEP_BOTH_WAYS, None, {"AFE_PERSIST_DEBUG": "1"})
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]
    assert "AQL queue" in out.stderr
    for says in ("no AQL queue for the resident grid", "resident grid stays on the HIP stream", "AQL dispatch of the resident grid refused"):
        assert says not in out.stderr


@pytest.mark.parametrize("fault,says", [("queue_create", "no AQL queue for the resident grid"),
                                        ("kernel_symbol", "resident grid stays on the HIP stream"),
                                        ("kernarg_size", "AQL dispatch of the resident grid refused")])
def test_a_refusal_of_the_own_queue_falls_back_to_the_hip_stream_with_the_same_bits(fault, says):
    out, _ = _child(STEP_BOTH_WAYS, fault)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]
    assert says in out.stderr, out.stderr[-2000:]


def test_a_grid_that_does_not_come_back_from_its_park_is_an_error_not_a_hang():
    """persist_collect's wait reports "still running": the call returns AFE_ERR_HIP, the engine refuses further steps
    with the same message at once, afe_destroy waits for the grid (which has really left) and frees; the next engine of
    the process steps with the launched kernels' bits."""
    code = r'''
import importlib, sys, time, numpy as np
sys.path.insert(0, %r)
from tests.test_gpu_persistent import make, assert_same
afa = importlib.import_module("agri-fly_amd")
b, _ = make(20000, afa.AFE_F32, True)
b.step(1000, 5)
try:
    b.get_state()                  # a getter ends the grid: the park whose wait "times out"
    raise SystemExit("the park's timeout was not reported")
except afa.AfeError as ex:
    assert "still running" in str(ex), str(ex)
t0 = time.perf_counter()
for call in (lambda: b.step(1000, 1), b.sync):
    try:
        call()
        raise SystemExit("a failed engine went on stepping")
    except afa.AfeError:
        pass
assert time.perf_counter() - t0 < 1.0           # sticky, immediate
b.close()
import os
os.environ.pop("AFE_FAULT")
a, _ = make(20000, afa.AFE_F32, False)
c, _ = make(20000, afa.AFE_F32, True)
a.step(1000, 30); c.step(1000, 30)
assert_same(a, c)
print("ok")
''' % ROOT
    out, secs = _child(code, "park_timeout")
    assert out.returncode == 0 and "ok" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])
    assert secs < 120


def test_a_sync_request_nobody_answers_fails_once_and_stays_failed():
    """Round-5 advisor: afe_sync's wait ran into its 20 s and failed WITHOUT marking the engine, the posted request stayed
    posted, and the next afe_sync spun another 20 s on it.  Patience cut to 1 s here (AFE_SYNC_PATIENCE_S): the first
    afe_sync fails after ~1 s naming the request, the second at once."""
    code = r'''
import importlib, sys, time, numpy as np
sys.path.insert(0, %r)
from tests.test_gpu_persistent import make
afa = importlib.import_module("agri-fly_amd")
b, _ = make(20000, afa.AFE_F32, True)
b.step(1000, 5)
t0 = time.perf_counter()
try:
    b.sync()
    raise SystemExit("afe_sync came back although nothing may be heard")
except afa.AfeError as ex:
    first = time.perf_counter() - t0
    assert "no progress" in str(ex) and "waiting for step 5" in str(ex), str(ex)
t0 = time.perf_counter()
try:
    b.sync()
    raise SystemExit("second afe_sync came back")
except afa.AfeError:
    second = time.perf_counter() - t0
assert 0.9 < first < 10.0 and second < 0.5, (first, second)
b.close()
print("ok")
''' % ROOT
    out, secs = _child(code, "sync_answer", {"AFE_SYNC_PATIENCE_S": "1"})
    assert out.returncode == 0 and "ok" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])
    assert secs < 120
