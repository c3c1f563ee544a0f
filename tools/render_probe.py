"""Times afe_render_depth on the config-5 shape (320x240 DepthVis views over a 32x32-tree procedural
orchard) for kernel A/B work and rocprofv3:  python tools/render_probe.py [n_views]"""
import importlib
import os
import sys

import numpy as np
import torch  # noqa: F401  (first: see INTEGRATION.md section 5)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
afa = importlib.import_module("agri-fly_amd")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    tris = afa.scenarios.orchard_mesh(rows=32, cols=32, seed=1)
    scene = afa.Scene(tris)
    info = scene.info()
    cam = afa.camera_default(320, 240)
    mount = afa.camera_default_mount()
    rng = np.random.default_rng(4)
    lo, hi = tris.reshape(-1, 3).min(0), tris.reshape(-1, 3).max(0)
    pos = np.stack([rng.uniform(lo[0] * 0.5, hi[0] * 0.8, n), rng.uniform(lo[1] * 0.5, hi[1] * 0.8, n),
                    rng.uniform(0.4, 3.0, n)])
    att = afa.scenarios.random_attitudes(rng, n, max_tilt_deg=25.0)
    best = 1e30
    for _ in range(3):
        imgs, ms = scene.render(cam, pos, att, mount)
        best = min(best, ms)
    print("%s: %d views x 320x240 over %d triangles (%d nodes, depth %d): %.2f ms, %.3g rays/s, %.0f%% of pixels hit"
          % (os.environ.get("AGRIFLY_ENGINE_LIB", "default"), n, info["n_tri"], info["n_nodes"], info["depth"], best,
             n * 76800 / (best * 1e-3), 100 * np.mean(imgs < 255)))


if __name__ == "__main__":
    main()
