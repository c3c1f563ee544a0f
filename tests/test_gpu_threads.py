"""Several host threads in one process, each with its own engine (its own stream), rendering ONE scene and planning at the
same time -- the shape of a host that drives the shards of a node from a thread each.  What they share inside the library:
the scene's tile-entry tables (taken under a mutex, freed by an event behind the render kernel that read them), its
pixel-ray tables, and the planner's device scratch (plan calls take turns).  Every image and every plan must be the one the
same call gives alone, with the engines stepping (launched kernels or the resident grid) between their frames.  In a child
process (ctypes releases the interpreter lock during the calls; a crash or a hang is this test's failure);
tools/host_sanitizers.sh runs it on the sanitized host build as well."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import importlib, sys, threading
import numpy as np
import torch  # noqa: F401
sys.path.insert(0, %r)
afa = importlib.import_module("agri-fly_amd")
T, ROUNDS, N = 4, 24, 96
tris = afa.scenarios.orchard_mesh(rows=5, cols=5, seed=11)
scene = afa.Scene(tris)
lo, hi = tris.reshape(-1, 3).min(0), tris.reshape(-1, 3).max(0)
cams = [afa.camera_default(160, 120), afa.camera_default(64, 48)]       # two camera geometries: two pixel-ray tables
mount = afa.camera_default_mount()
params = afa.params_from_type(5)
samples = {c.width: afa.planner_samples(0, c.width, c.height, 48) for c in cams}
cfgs = {c.width: afa.planner_default_config(c.width, c.height, c.depth_scale, c.focal_length, 2 * params.arm_length, 3 * params.arm_length, 0.5) for c in cams}

def make(k):
    rng = np.random.default_rng(100 + k)
    e = afa.Ensemble(N, precision=afa.AFE_F32 if k %% 2 else afa.AFE_F64)
    e.set_type_table([params])
    e.set_imu_noise(True, 0.1, 0.2, afa.AFE_SEED_DECORRELATED)
    e.set_step_mode(afa.AFE_STEP_PERSISTENT if k %% 3 else afa.AFE_STEP_LAUNCH)
    pos = np.stack([rng.uniform(lo[0] + 1, hi[0] - 1, N), rng.uniform(lo[1] + 1, hi[1] - 1, N), rng.uniform(0.5, 2.5, N)])
    att = afa.scenarios.random_attitudes(rng, N, max_tilt_deg=15.0)
    e.set_state(pos, np.zeros((3, N)), att, np.zeros((3, N)), np.full((4, N), afa.scenarios.hover_speed(params)))
    e.set_motor_cmds(np.full((4, N), afa.scenarios.hover_speed(params) * 1.02, np.float32))
    vel = rng.normal(0, 1.0, (3, N)); acc = rng.normal(0, 0.5, (3, N)); grav = np.tile(np.array([[0.0], [9.81], [0.0]]), (1, N))
    return e, vel, acc, grav

def frame(k, e, vel, acc, grav, cam, buf):
    e.step(1000, 3)                                 # (the vehicles climb and drift: every round sees other images)
    scene.render_engine(e, cam, mount, out=buf)
    img = buf.download(np.uint16, (N, cam.height, cam.width))
    out, _, _ = afa.rappids_plan(cfgs[cam.width], buf, vel, acc, grav, samples[cam.width])
    return img, afa.plans_as_array(out).tobytes()

twins = [make(k) for k in range(T)]                 # the same rounds, one engine after the other, one call at a time
engines = [make(k) for k in range(T)]
bufs = [afa.DeviceBuffer(N * 120 * 160 * 2) for _ in range(T)]
alone = [[frame(k, *twins[k], cams[(r + k) %% 2], bufs[k]) for r in range(ROUNDS)] for k in range(T)]
assert any(np.any(a[0][0] < 255) for a in alone) and len({a[0][1] for a in alone}) > 1          # (something to see, plans differ)
assert not np.array_equal(alone[0][0][0], alone[0][2][0])                                       # (and the views do move)
wrong, errors = [], []
start = threading.Barrier(T)

def worker(k):
    try:
        start.wait()
        for r in range(ROUNDS):
            img, plans = frame(k, *engines[k], cams[(r + k) %% 2], bufs[k])
            if not np.array_equal(img, alone[k][r][0]):
                wrong.append("thread %%d round %%d: %%d pixels differ" %% (k, r, int((img != alone[k][r][0]).sum())))
            if plans != alone[k][r][1]:
                wrong.append("thread %%d round %%d: plans differ" %% (k, r))
    except Exception as ex:        # noqa: BLE001
        errors.append("thread %%d: %%r" %% (k, ex))

threads = [threading.Thread(target=worker, args=(k,)) for k in range(T)]
for t in threads: t.start()
for t in threads: t.join(timeout=300)
assert not any(t.is_alive() for t in threads), "a thread hangs"
assert not errors, errors
assert not wrong, wrong[:10]
for e, *_ in engines + twins: e.close()
for b in bufs: b.close()
scene.close()
print("ok", T, "threads x", ROUNDS, "frames")
''' % ROOT


def test_threads_sharing_a_scene_and_the_planner_get_the_results_of_a_call_alone():
    out = subprocess.run([sys.executable, "-c", CODE], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "ok" in out.stdout, (out.returncode, out.stdout[-800:], out.stderr[-3000:])
