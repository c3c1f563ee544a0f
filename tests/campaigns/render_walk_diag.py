import importlib, os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import torch
afa = importlib.import_module("agri-fly_amd")
from oracle import oracle_py as ora
scen = afa.scenarios
tris = scen.orchard_mesh(rows=16, cols=16, seed=11)
scene = afa.Scene(tris)
cam = afa.camera_default(320, 240)
mount = afa.camera_default_mount()
rng = np.random.default_rng(21)
n = 4096
pos = np.stack([rng.uniform(-10, 60, n), rng.uniform(-10, 60, n), rng.uniform(0.3, 9.0, n)])
q = rng.normal(size=(4, n)); q /= np.linalg.norm(q, axis=0)
k = n // 8
yaw = rng.integers(0, 4, k) * (np.pi / 2)
q[:, :k] = np.stack([np.cos(yaw / 2), 0 * yaw, 0 * yaw, np.sin(yaw / 2)])
pos[:, :k] = np.round(pos[:, :k])
ident = np.array([1.0, 0.0, 0.0, 0.0])
for name, m in (("mount", mount), ("ident", ident)):
    scene.set_walk(False); a, _ = scene.render(cam, pos, q, m)
    scene.set_walk(True); b, _ = scene.render(cam, pos, q, m)
    scene.set_walk(False)
    d = a != b
    views = np.nonzero(d.reshape(n, -1).any(1))[0]
    print(name, "differing pixels", int(d.sum()), "in views", views[:20], "of which axis-aligned:", int((views < k).sum()))
    if len(views):
        v = int(views[0])
        ys, xs = np.nonzero(d[v])
        print(" view", v, "pos", pos[:, v], "q", q[:, v], "pixels", list(zip(xs[:8], ys[:8])), "ordered", a[v][d[v]][:8], "plain", b[v][d[v]][:8])
        oc = ora.render_camera(cam.width, cam.height, cam.focal_length, cam.depth_scale, cam.max_count)
        oc.cx, oc.cy = cam.cx, cam.cy
        want = ora.render_depth(oc, tris, pos[:, v], q[:, v], m)
        print(" vs checker: ordered wrong", int((a[v] != want).sum()), "plain wrong", int((b[v] != want).sum()))
