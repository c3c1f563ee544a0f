import importlib, os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
afa = importlib.import_module("agri-fly_amd")
tris = afa.scenarios.orchard_mesh(rows=32, cols=32, seed=1)
scene = afa.Scene(tris)
cam = afa.camera_default(320, 240); mount = afa.camera_default_mount()
rng = np.random.default_rng(4); n = 1024
pos = np.stack([rng.uniform(-5, 90, n), rng.uniform(-5, 120, n), rng.uniform(0.8, 2.5, n)])
yaw = rng.uniform(-np.pi, np.pi, n); att = np.stack([np.cos(yaw / 2), 0 * yaw, 0 * yaw, np.sin(yaw / 2)])
for _ in range(3): img, ms = scene.render(cam, pos, att, mount)
st, ms2 = scene.render_stats(cam, pos, att, mount)
print("render 1024 views: %.2f ms (%.3g rays/s); counting build %.2f ms" % (ms, n*76800/ms*1e3, ms2))
print(st)
r = st["rays"]; w = st["waves"]
print("triangles a tile really shows (distinct closest hits per wave) %.2f" % (st["visible_triangles_per_wave"]/w))
print("per ray: box tests %.1f, fp64 tests %.1f | per wave: nodes %.1f, tri box %.1f, tri fp64 executed %.1f" % (st["tri_box_tests_per_ray"]/r, st["tri_fp64_tests_per_ray"]/r, st["nodes_per_wave"]/w, st["tri_box_tests_per_wave"]/w, st["tri_fp64_tests_per_wave"]/w))
