"""Writes tests/golden/render_regression.npz: one small depth image of the procedural orchard
as the CPU checker (oracle/agrifly_oracle_render.c) renders it.  A regression fixture for the
checker and the mesh generator -- the reference contains no renderer to take vectors from
(its image comes from AirSim/Unity, Simulator/Rappids_Simulator/main.cpp:332-354).

    python tests/golden/make_render_golden.py
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from oracle import oracle_py
    oracle_py.build(force=True)
    scen = importlib.import_module("agri-fly_amd").scenarios
    tris = scen.orchard_mesh(rows=3, cols=4, seed=5)
    cam = oracle_py.render_camera(80, 60)
    y = r = -np.pi / 2
    mount = np.array([np.cos(y / 2) * np.cos(r / 2), np.cos(y / 2) * np.sin(r / 2), np.sin(y / 2) * np.sin(r / 2),
                      np.sin(y / 2) * np.cos(r / 2)])          # FromEulerYPR(-90 deg, 0, -90 deg)
    img = oracle_py.render_depth(cam, tris, [-3.0, 2.0, 1.5], [1, 0, 0, 0], mount)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "render_regression.npz"), triangles=tris, image=img)
    print("render_regression.npz: %d triangles, %d%% of pixels hit" % (len(tris), 100 * np.mean(img < 255)))


if __name__ == "__main__":
    main()
