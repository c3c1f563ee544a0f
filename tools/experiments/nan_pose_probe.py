"""What a diverged vehicle costs the depth camera and the planner: 256 views / plans, then the same with ONE pose (state) made
NaN / inf.  python tools/experiments/nan_pose_probe.py"""
import importlib, os, sys, time
import numpy as np
import torch  # noqa: F401
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
afa = importlib.import_module("agri-fly_amd")
tris = afa.scenarios.orchard_mesh(rows=32, cols=32, seed=1)
scene = afa.Scene(tris)
cam = afa.camera_default(320, 240)
mount = afa.camera_default_mount()
rng = np.random.default_rng(4)
n = 256
lo, hi = tris.reshape(-1, 3).min(0), tris.reshape(-1, 3).max(0)
pos = np.stack([rng.uniform(lo[0] + 2, hi[0] - 2, n), rng.uniform(lo[1] + 2, hi[1] - 2, n), rng.uniform(0.5, 3.0, n)])
att = afa.scenarios.random_attitudes(rng, n, max_tilt_deg=20.0)
def timed(p, a, what):
    best = 1e30
    for _ in range(3):
        img, ms = scene.render(cam, p, a, mount)
        best = min(best, ms)
    print("%-34s %8.2f ms   view 7: %d pixels hit" % (what, best, int((img[7] < 255).sum())), flush=True)
    return img
ref = timed(pos, att, "256 views")
for what, fn in (("position x = NaN", lambda p, a: p.__setitem__((0, 7), np.nan)), ("position z = +inf", lambda p, a: p.__setitem__((2, 7), np.inf)),
                 ("attitude w = NaN", lambda p, a: a.__setitem__((0, 7), np.nan)), ("attitude = 1e200 (1, 1, 1, 1)", lambda p, a: a.__setitem__((slice(None), 7), 1e200))):
    p, a = pos.copy(), att.copy()
    fn(p, a)
    img = timed(p, a, "view 7: " + what)
    keep = np.arange(n) != 7
    assert np.array_equal(img[keep], ref[keep])
    assert (img[7] == 255).all(), "a view from nowhere shows something"

# ---- the planner: 256 plans on the first views, then one planner's state made NaN / inf
cfg = afa.planner_default_config(320, 240, cam.depth_scale, cam.focal_length, 0.116, 0.174, 0.5)
samples = afa.planner_samples(0, 320, 240, 192)
vel = rng.normal(0, 1.0, (3, n)); acc = rng.normal(0, 0.5, (3, n)); grav = np.tile(np.array([[0.0], [9.81], [0.0]]), (1, n))
def plan(v, a, g, what):
    best = 1e30
    for _ in range(3):
        out, _, ms = afa.rappids_plan(cfg, ref, v, a, g, samples)
        best = min(best, ms)
    arr = afa.plans_as_array(out)
    print("%-34s %8.2f ms   plan 7: found %d, %d collision checks, %d pyramids" % (what, best, arr["found"][7], arr["n_collision_checks"][7], arr["n_pyramids"][7]), flush=True)
    return arr
good = plan(vel, acc, grav, "256 plans")
for what, fn in (("velocity x = NaN", lambda v, a, g: v.__setitem__((0, 7), np.nan)), ("acceleration z = inf", lambda v, a, g: a.__setitem__((2, 7), np.inf)),
                 ("gravity = NaN", lambda v, a, g: g.__setitem__((slice(None), 7), np.nan)), ("velocity = 1e300", lambda v, a, g: v.__setitem__((slice(None), 7), 1e300))):
    v, a, g = vel.copy(), acc.copy(), grav.copy()
    fn(v, a, g)
    arr = plan(v, a, g, "plan 7: " + what)
    keep = np.arange(n) != 7
    assert arr[keep].tobytes() == good[keep].tobytes()
