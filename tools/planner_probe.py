"""Times afe_rappids_plan on the config-3 shape (65536 planners x 256 candidates, 16 synthetic
320x240 depth images) for kernel A/B work:  AGRIFLY_ENGINE_LIB=<variant.so> python tools/planner_probe.py"""
import importlib
import os
import sys

import numpy as np
import torch  # noqa: F401  (first: see INTEGRATION.md section 5)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
afa = importlib.import_module("agri-fly_amd")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    m, n_img = 256, 16
    rng = np.random.default_rng(9)
    images = np.stack([afa.scenarios.synthetic_depth_image(seed=300 + k, n_trunks=3 + k % 6) for k in range(n_img)])
    if len(sys.argv) > 2 and sys.argv[2] == "orchard":
        # the bench's cluttered case: views rendered from random poses inside the procedural orchard
        n_img = 512
        scene = afa.Scene(afa.scenarios.orchard_mesh(rows=32, cols=32, seed=1))
        cam = afa.camera_default(320, 240)
        r4 = np.random.default_rng(4)
        pos = np.stack([r4.uniform(-5, 90, n_img), r4.uniform(-5, 120, n_img), r4.uniform(0.8, 2.5, n_img)])
        yaw = r4.uniform(-np.pi, np.pi, n_img)
        att = np.stack([np.cos(yaw / 2), 0 * yaw, 0 * yaw, np.sin(yaw / 2)])
        images, _ = scene.render(cam, pos, att, afa.camera_default_mount())
        images = np.asarray(images).reshape(n_img, 240, 320)
    cfg = afa.planner_default_config(320, 240, 10.0 / 256.0, 160.0, 0.116, 0.174, 0.5)
    cfg.cost_type = 1
    cfg.cost_vec[2] = 120.0
    image_index = (np.arange(n) % n_img).astype(np.int32)
    vel0 = np.stack([rng.normal(0, 0.4, n), rng.normal(0, 0.2, n), rng.uniform(0, 2.0, n)])
    acc0 = rng.normal(0, 0.3, (3, n))
    grav = np.tile(np.array([[0.0], [9.81], [0.0]]), (1, n))
    samples = afa.planner_samples(0, 320, 240, m)
    import time
    best, wall = 1e30, 1e30
    for _ in range(3):
        t0 = time.perf_counter()
        out, _, ms = afa.rappids_plan(cfg, images, vel0, acc0, grav, samples, image_index=image_index)
        wall = min(wall, (time.perf_counter() - t0) * 1e3)
        best = min(best, ms)
    found = np.mean([o.found for o in out])
    pyr = np.mean([o.n_pyramids for o in out])
    cc = np.mean([o.n_collision_checks for o in out])
    print("%s: %d planners x %d candidates: %.1f ms (%.3g plans/s; call wall %.1f ms); found %.2f, pyramids/plan %.2f, collision checks/plan %.1f"
          % (os.environ.get("AGRIFLY_ENGINE_LIB", "default"), n, m, best, n / (best * 1e-3), wall, found, pyr, cc))


if __name__ == "__main__":
    main()
