"""keeps the GPU busy for <seconds> (65 536 orchard plans over and over): the other process of the reproducibility probes"""
import importlib, os, sys, time
import numpy as np
import torch  # noqa
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
afa = importlib.import_module("agri-fly_amd")
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
n, m, n_img = 16384, 256, 64
scene = afa.Scene(afa.scenarios.orchard_mesh(rows=16, cols=16, seed=1))
cam = afa.camera_default(320, 240)
r4 = np.random.default_rng(4)
pos = np.stack([r4.uniform(-5, 40, n_img), r4.uniform(-5, 60, n_img), r4.uniform(0.8, 2.5, n_img)])
yaw = r4.uniform(-np.pi, np.pi, n_img)
att = np.stack([np.cos(yaw / 2), 0 * yaw, 0 * yaw, np.sin(yaw / 2)])
images, _ = scene.render(cam, pos, att, afa.camera_default_mount())
images = np.asarray(images).reshape(n_img, 240, 320)
cfg = afa.planner_default_config(320, 240, 10.0 / 256.0, 160.0, 0.116, 0.174, 0.5)
idx = (np.arange(n) % n_img).astype(np.int32)
rng = np.random.default_rng(9)
vel0 = np.stack([rng.normal(0, 0.4, n), rng.normal(0, 0.2, n), rng.uniform(0, 2.0, n)])
acc0 = rng.normal(0, 0.3, (3, n))
grav = np.tile(np.array([[0.0], [9.81], [0.0]]), (1, n))
samples = afa.planner_samples(0, 320, 240, m)
open(os.path.join("/tmp", "gpu_load_ready"), "w").write("1")
t0 = time.time()
while time.time() - t0 < seconds:
    afa.rappids_plan(cfg, images, vel0, acc0, grav, samples, image_index=idx)
