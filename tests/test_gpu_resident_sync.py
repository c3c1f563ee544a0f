"""Round 4: the resident grid lives on the engine's own AQL queue (agri-fly_amd/csrc/afe_aql.cpp).  What that buys and
what it must not break:
  * afe_sync waits for the authorised steps (the workers answer a sync marker) and LEAVES the grid resident; the next
    afe_step is served without a launch; every reader of the state still sees the state after the last step;
  * a device-wide synchronisation issued elsewhere (torch.cuda.synchronize = hipDeviceSynchronize) neither waits for the
    grid nor ends it;
  * once afe_get_device_view has handed the slabs out, afe_sync ends the grid as before (a reader the engine does not know
    about must find the state in memory);
  * AFE_PERSIST_AQL=0 (the grid on the HIP stream, as in round 3) gives the same bits.
Loop shape replaced: Simulator/Rappids_Simulator/main.cpp:330,391-392 (Run(); clock += dt; ... no barrier between steps)."""
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from tests.scenarios import random_ensemble
from tests.test_gpu_persistent import assert_same, everything, make

afa = importlib.import_module("agri-fly_amd")
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("precision", [afa.AFE_F32, afa.AFE_F64])
@pytest.mark.parametrize("resident", [False, True])
def test_sync_leaves_the_grid_resident_and_blocks_of_steps_are_the_launched_bits(precision, resident):
    n = 40000
    a, _ = make(n, precision, True, resident=resident)
    b, _ = make(n, precision, False)
    with a, b:
        grids = 0
        for block in range(12):
            for _ in range(7):
                a.step(1000, 1)
                b.step(1000, 1)
            a.sync()
            assert a.steps_completed == 7 * (block + 1)
            grids += int(a.persistent_running)
        assert grids >= 10, "afe_sync ended the grid %d times out of 12" % (12 - grids)     # (a slow box may let one idle out)
        assert_same(a, b, "blocks of 7 steps with afe_sync in between")
        assert not a.persistent_running                       # the getters ended it
        # a second afe_sync with nothing new is immediate and true
        a.step(1000, 3); b.step(1000, 3)
        a.sync(); a.sync()
        assert_same(a, b, "double sync")


@pytest.mark.parametrize("n, logic, resident, own_queue", [
    (200000, False, False, -1),    # 3 125 workers, three or four on a SIMD: issue priority by steps in hand, own queue by size
    (200000, True, True, -1),      # ... the resident-state kernel with the logic: per-type constants from LDS
    (400000, False, False, 1),     # 6 250 chunks on 6 143 workers: some have two, most one (they poll every 8 us), priority on
    (1 << 20, False, False, 1),    # 2.7 chunks per worker: priority off, grids retired after 512 steps
])
def test_irregular_blocks_on_grids_of_every_regime_are_the_launched_bits(n, logic, resident, own_queue):
    """The second half of round 4 changed how a resident grid schedules itself, never what it computes: the pump reads the
    host's ring through the scalar path (eight entries at a time, 64 when the host is far ahead), workers take issue
    priority by the steps they have in hand, workers with chunks to spare poll rarely, the logic's constants come from
    LDS.  Blocks of every length from one step to more than a ring read, posted step by step and in bursts."""
    a, _ = make(n, afa.AFE_F32, True, logic=logic, resident=resident, seed=11)
    b, _ = make(n, afa.AFE_F32, False, logic=logic, seed=11)
    with a, b:
        a.set_resident_queue(own_queue)
        total = 0
        for block, k in enumerate((1, 2, 3, 5, 8, 13, 21, 70, 9, 1, 150, 4)):
            if block % 3 == 2:
                a.step(1000, k)                       # one call: the host is k entries ahead at once
            else:
                for _ in range(k):
                    a.step(1000, 1)
            b.step(1000, k)
            total += k
            a.sync()
            assert a.steps_completed == total
        assert_same(a, b, "%d vehicles, logic %s, resident state %s, queue %d" % (n, logic, resident, own_queue))


def test_a_device_wide_synchronise_neither_waits_for_the_grid_nor_ends_it():
    import torch
    n = 1 << 18
    a, _ = make(n, afa.AFE_F32, True)
    b, _ = make(n, afa.AFE_F32, False)
    with a, b:
        a.step(1000, 50); b.step(1000, 50)
        a.sync()
        assert a.persistent_running
        ts = []
        for _ in range(20):
            a.step(1000, 1)                                   # keeps the grid fed: it is resident during every synchronise below
            t0 = time.perf_counter()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        assert a.persistent_running, "hipDeviceSynchronize ended the grid"
        assert np.median(ts) < 100e-6, "hipDeviceSynchronize waited for the resident grid: %.0f us" % (np.median(ts) * 1e6)   # its idle patience is 200 us
        b.step(1000, 20)
        a.sync()
        assert_same(a, b, "steps around device-wide synchronisations")


def test_an_exported_device_view_makes_sync_end_the_grid():
    import torch
    n = 20000
    a, _ = make(n, afa.AFE_F32, True)
    b, _ = make(n, afa.AFE_F32, False)
    with a, b:
        a.step(1000, 5); b.step(1000, 5)
        a.sync()
        assert a.persistent_running
        v = a.device_view()                                   # from here on somebody else may read the slabs

        class _Wrap:
            def __init__(self, ptr, shape):
                self.__cuda_array_interface__ = dict(shape=shape, typestr="<f4", data=(ptr, False), version=2, strides=None)
        vel = torch.as_tensor(_Wrap(v.vel, (3, v.stride)), device="cuda")[:, :n]
        for _ in range(3):
            for _ in range(4):
                a.step(1000, 1); b.step(1000, 1)
            a.sync()
            assert not a.persistent_running                   # ended: the slabs are in memory for the outside reader
            np.testing.assert_array_equal(vel.cpu().numpy().astype(np.float64), b.get_state()["vel"])
        assert_same(a, b, "with an exported view")


def test_grid_time_counts_the_steps_the_grids_served():
    n = 1 << 17
    a, _ = make(n, afa.AFE_F32, True)
    with a:
        a.grid_time()
        for _ in range(5):
            a.step(1000, 40)
            a.sync()
        seconds, steps = a.grid_time()
        assert steps == 200
        assert 200 * 0.5e-6 < seconds < 200 * 50e-6, seconds      # device timestamps: microseconds per step, not garbage
        assert a.grid_time() == (0.0, 0)


def test_the_hip_stream_fallback_gives_the_same_bits():
    """AFE_PERSIST_AQL=0: the grid is launched on the engine's HIP stream and parked by every afe_sync (round 3's behaviour,
    and what a host gets where the runtime's queue interface cannot be reached)"""
    code = r'''
import importlib, json, sys
import numpy as np
sys.path.insert(0, %r)
afa = importlib.import_module("agri-fly_amd")
from tests.test_gpu_persistent import make, everything
a, _ = make(30000, afa.AFE_F32, True)
stayed = 0
for block in range(6):
    a.step(1000, 1); a.step(1000, 1); a.step(1000, 3)
    a.sync()
    stayed += int(a.persistent_running)
x = everything(a)
import hashlib
print(json.dumps({"stayed": stayed, "digest": hashlib.sha256(b"".join(np.ascontiguousarray(x[k]).tobytes() for k in sorted(x))).hexdigest()}))
''' % ROOT
    out = {}
    for aql in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, AFE_PERSIST_AQL=aql), capture_output=True, text=True, timeout=300, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        out[aql] = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["0"]["stayed"] == 0 and out["1"]["stayed"] >= 5
    assert out["0"]["digest"] == out["1"]["digest"]


def test_reserved_compute_units_let_other_kernels_run_beside_the_grid():
    """afe_set_reserved_compute_units: the grid's queue is masked off a row of compute units (one per shader engine and XCD);
    another stream's kernels then run beside a grid that is kept fed instead of waiting for it to idle out (200 us), and
    the bits are the launches' either way"""
    import torch
    n = 1 << 20
    a, _ = make(n, afa.AFE_F32, True)
    b, _ = make(n, afa.AFE_F32, False)
    with a, b:
        a.set_resident_queue(1)                               # (automatic would put a grid of this size on the HIP stream)
        x = torch.ones(1 << 20, device="cuda")
        med, steps = {}, 0
        for reserve in (0, 1):
            a.set_reserved_compute_units(reserve)
            a.step(1000, 30); steps += 30
            a.sync()
            assert a.persistent_running
            torch.cuda.synchronize()
            ts = []
            for _ in range(20):
                a.step(1000, 1); steps += 1                   # the grid has work while the other kernels want to start
                t0 = time.perf_counter()
                y = float((x * 2).sum().item())               # torch's stream: two small kernels and a read-back
                ts.append(time.perf_counter() - t0)
                assert y == 2.0 * (1 << 20)
            med[reserve] = float(np.median(ts))
        assert med[0] > 200e-6, "without a reservation the kernels should have waited for the grid's idle patience: %.0f us" % (med[0] * 1e6)
        assert med[1] < 0.6 * med[0], "reserved compute units did not let the kernels in: %.0f against %.0f us" % (med[1] * 1e6, med[0] * 1e6)
        a.set_reserved_compute_units(0)                       # parks; the next grid has the whole device again
        a.step(1000, 10); steps += 10
        a.sync()
        b.step(1000, steps)                                   # (the launched engine afterwards: its kernels want the whole device too)
        assert_same(a, b, "with and without reserved compute units")
