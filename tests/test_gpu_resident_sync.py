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


def test_the_own_queue_finds_its_agent_by_identity_not_by_being_alone():
    """Round-4 review, multi-GPU-only code: which HSA agent is HIP device d?  On a one-GPU box the old code could fall back
    to "the only GPU agent" and nobody would have noticed a PCI match that never worked -- on the 8-GPU node there is no
    such fallback and the grid would silently stay on the HIP stream.  AFE_PERSIST_DEBUG says which rule chose the agent:
    it must be the PCI address (or the unique id), the grid of a north-star shard (131 072 vehicles) must live on the own
    queue, survive afe_sync, and step the launched bits."""
    code = r'''
import importlib, json, sys, hashlib
import numpy as np
sys.path.insert(0, %r)
afa = importlib.import_module("agri-fly_amd")
from tests.test_gpu_persistent import make, everything
res = {}
for name, persistent in (("grid", True), ("launches", False)):
    e, _ = make(131072, afa.AFE_F32, persistent, seed=5)
    stayed = 0
    for block in range(5):
        for _ in range(9):
            e.step(1000, 1)
        e.sync()
        stayed += int(e.persistent_running)
    x = everything(e)
    res[name] = {"stayed": stayed, "digest": hashlib.sha256(b"".join(np.ascontiguousarray(x[k]).tobytes() for k in sorted(x))).hexdigest()}
    e.close()
print(json.dumps(res))
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, AFE_PERSIST_DEBUG="1"), capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    chosen = [l for l in r.stderr.splitlines() if "is the HSA agent chosen by" in l]
    assert chosen, r.stderr[-2000:]
    assert all(("chosen by PCI address" in l) or ("chosen by unique id" in l) for l in chosen), chosen
    assert out["grid"]["stayed"] >= 4, out                     # on the own queue: afe_sync left it resident
    assert out["grid"]["digest"] == out["launches"]["digest"]


@pytest.mark.parametrize("n, host_visible, own_queue", [
    (16384, True, -1),       # 256 workers on a host-visible arena: more than the 64 host marks -> the sync request's path; the host reads the slabs in place
    (1 << 20, False, 1),     # forced own queue: 6 143 workers, a third of them with chunks to spare (they poll every ~8 us)
    (200000, False, -1),
])
def test_sync_after_sync_with_pauses_around_the_grids_patience_never_returns_early(n, host_visible, own_queue):
    """Round-4 advisor: the workers' arrival counters of a sync request are never reset within a launch, so a request the
    host LEFT while only part of the workers had answered (through the pump's older completion word) misaligned them for
    the life of the grid, and a later afe_sync could return while workers were still stepping.  quiesce now leaves a
    posted request only through the workers' own answer.  Random pauses between afe_sync and the next afe_step -- below,
    at and above the grid's 200 us of idle patience, so that grids also park themselves in between -- random block
    lengths; after EVERY afe_sync the completed count is exact, and on the host-visible arena (where the host reads the
    slabs in place, without ending the grid) the state is the launched engine's, bit for bit."""
    rng = np.random.default_rng(n)
    d = random_ensemble(n, seed=21, with_wrench=True, type_ids=(5,)).data

    def engine(persistent):
        e = afa.Ensemble(n, precision=afa.AFE_F32, host_visible=host_visible)
        e.set_type_table([afa.params_from_type(d.type_ids[0])])
        e.set_logic_period(1 / 500)
        e.set_imu_noise(True, 0.1, 0.2, afa.AFE_SEED_DECORRELATED)
        e.set_state(d.pos, d.vel, d.att, d.ang_vel, d.motor_speed)
        e.set_motor_cmds(d.motor_cmd)
        e.set_external_force(d.ext_force)
        e.set_split_stepping(1)
        e.set_step_mode(afa.AFE_STEP_PERSISTENT if persistent else afa.AFE_STEP_LAUNCH)
        return e

    a, b = engine(True), engine(False)
    with a, b:
        if own_queue >= 0:
            a.set_resident_queue(own_queue)
        total = 0
        for block in range(60 if n <= 200000 else 30):
            k = int(rng.integers(1, 12))
            for _ in range(k):
                a.step(1000, 1)
            b.step(1000, k)
            total += k
            a.sync()
            assert a.steps_completed == total
            if host_visible:
                sa, sb = a.get_state(), b.get_state()
                for key in sa:
                    assert np.array_equal(sa[key], sb[key], equal_nan=True), (block, key)
            pause = float(rng.choice([0.0, 20e-6, 150e-6, 220e-6, 400e-6]))
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < pause:
                pass
        assert_same(a, b, "%d vehicles, %d steps in random blocks with pauses" % (n, total))
