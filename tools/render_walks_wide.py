import importlib, os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import torch
afa = importlib.import_module("agri-fly_amd")
tris = afa.scenarios.orchard_mesh(rows=32, cols=32, seed=5)
scene = afa.Scene(tris)
cam = afa.camera_default(320, 240)
mount = afa.camera_default_mount()
total = bad = 0
for rep in range(6):
    rng = np.random.default_rng(100 + rep)
    n = 4096
    pos = np.stack([rng.uniform(-10, 100, n), rng.uniform(-10, 130, n), rng.uniform(0.2, 12.0, n)])
    q = rng.normal(size=(4, n)); q /= np.linalg.norm(q, axis=0)
    k = n // 8
    ax = rng.integers(0, 3, k); ang = rng.integers(0, 4, k) * (np.pi / 2)
    q[:, :k] = 0; q[0, :k] = np.cos(ang / 2); q[1 + ax, np.arange(k)] = np.sin(ang / 2)
    pos[:, :k] = np.round(pos[:, :k])
    scene.set_walk(False); a, ms_a = scene.render(cam, pos, q, mount)
    scene.set_walk(True); b, ms_b = scene.render(cam, pos, q, mount)
    d = int((a != b).sum()); total += a.size; bad += d
    print("batch %d: %d views, ordered %.1f ms plain %.1f ms, differing pixels %d" % (rep, n, ms_a, ms_b, d), flush=True)
print("rays", total, "differing", bad)
