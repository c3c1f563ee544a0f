"""CPU checks of the depth-camera checker itself (oracle/agrifly_oracle_render.c).

The reference has no renderer (AirSim/Unity produce its image), so there is nothing to pin the
geometry against except analytic scenes; what IS taken from the reference is the image contract
(main.cpp:120-125,352-360,484-486,520), and these tests hold the checker to it."""
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def scen(afa):
    return afa.scenarios


def default_mount():
    # Rotationd::FromEulerYPR(-90 deg, 0, -90 deg), main.cpp:123-125
    y, p, r = -np.pi / 2, 0.0, -np.pi / 2
    cy, sy, cp, sp, cr, sr = np.cos(y / 2), np.sin(y / 2), np.cos(p / 2), np.sin(p / 2), np.cos(r / 2), np.sin(r / 2)
    return np.array([cy * cp * cr + sy * sp * sr, cy * cp * sr - sy * sp * cr, cy * sp * cr + sy * cp * sr,
                     sy * cp * cr - cy * sp * sr])


def test_mount_axes(ora):
    """Camera z (optical axis) = body x, camera x (image right) = body -y, camera y (image down) = body -z."""
    import ctypes as C
    R = np.empty(9)
    q = default_mount()
    ora.render_lib().ora_quat_to_matrix(q.ctypes.data_as(C.POINTER(C.c_double)), R.ctypes.data_as(C.POINTER(C.c_double)))
    R = R.reshape(3, 3)
    assert np.allclose(R @ [0, 0, 1], [1, 0, 0], atol=1e-15)
    assert np.allclose(R @ [1, 0, 0], [0, -1, 0], atol=1e-15)
    assert np.allclose(R @ [0, 1, 0], [0, 0, -1], atol=1e-15)


def test_ground_plane_is_analytic(ora):
    cam = ora.render_camera(64, 48)
    ground = np.array([[-100, -100, 0, 100, -100, 0, 100, 100, 0], [-100, -100, 0, 100, 100, 0, -100, 100, 0]],
                      np.float32)
    h = 1.25
    img = ora.render_depth(cam, ground, [0, 0, h], [1, 0, 0, 0], default_mount())
    v = (np.arange(48) - cam.cy) / cam.focal_length
    with np.errstate(divide="ignore"):
        z = np.where(v > 0, h / np.where(v > 0, v, 1), np.inf)      # level camera: ground depth depends on the row only
    want = np.where(np.isfinite(z), np.minimum(np.floor(z / cam.depth_scale), 255), 255).astype(np.uint16)
    diff = img.astype(int) - want[:, None].astype(int)
    assert np.abs(diff).max() <= 1 and np.count_nonzero(diff) <= 0.005 * img.size
    assert np.all(img[:24] == 255)                                   # rows at and above the horizon: no hit


def test_sphere_depth(ora, scen):
    sv, sf = scen._icosphere(3)
    R, D = 0.5, 4.0
    tris = (sv * R + [D, 0, 1.0])[sf].reshape(-1, 9).astype(np.float32)
    cam = ora.render_camera(32, 24)
    z = ora.render_pixel_depth(cam, tris, [0, 0, 1.0], [1, 0, 0, 0], default_mount(), 16, 12)
    assert abs(z - (D - R)) < 5e-3                                   # facet error of the 1280-triangle sphere
    assert not np.isfinite(ora.render_pixel_depth(cam, tris, [0, 0, 1.0], [1, 0, 0, 0], default_mount(), 0, 0))
    # yaw the vehicle by 90 deg: the sphere on +x is no longer in view
    q = [np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)]
    assert np.all(ora.render_depth(cam, tris, [0, 0, 1.0], q, default_mount()) == 255)


def test_counts_saturate_and_floor(ora):
    cam = ora.render_camera(16, 12)
    wall = lambda x: np.array([[x, -50, -50, x, 50, -50, x, 50, 50], [x, -50, -50, x, 50, 50, x, -50, 50]], np.float32)
    for x, want in ((1.0, 25), (0.0390625 * 7.5, 7), (9.99, 255), (25.0, 255)):
        img = ora.render_depth(cam, wall(x), [0, 0, 0], [1, 0, 0, 0], default_mount())
        assert np.all(img == want), (x, np.unique(img))


def test_orchard_mesh_shape_and_regression(ora, scen, golden_dir):
    tris = scen.orchard_mesh(rows=3, cols=4, seed=5)
    assert tris.shape == (2 + 12 * (16 + 80), 9) and tris.dtype == np.float32
    assert np.array_equal(tris, scen.orchard_mesh(rows=3, cols=4, seed=5))
    cam = ora.render_camera(80, 60)
    img = ora.render_depth(cam, tris, [-3.0, 2.0, 1.5], [1, 0, 0, 0], default_mount())
    assert 0.02 < np.mean(img < 255) < 0.98                           # trees and ground in view, and some sky
    gold = np.load(os.path.join(golden_dir, "render_regression.npz"))
    assert np.array_equal(gold["triangles"], tris)
    assert np.array_equal(gold["image"], img)


def test_bvh_builder_invariants_on_the_host(afa, scen):
    """afe_scene_check_hierarchy builds the hierarchy the GPU traverses and verifies it without a GPU:
    every triangle in exactly one leaf, boxes containing what hangs below them, depth within the
    traversal stack, and the eight octant-mirrored pair-node arrays the kernel walks (each reaches every
    triangle once, boxes mirrored exactly, children exchanged across mirrored split axes) -- for the orchard, for coincident / degenerate triangles (median-split fallback)
    and for a single triangle."""
    tris = scen.orchard_mesh(rows=6, cols=8, seed=3)
    n_nodes, depth, max_leaf = afa.scene_check_hierarchy(tris)
    assert n_nodes > len(tris) // 4 and depth <= 32 and 1 <= max_leaf <= 4
    one = np.array([[3, -1, 0, 3, 1, 0, 3, 0, 2]], np.float32)
    assert afa.scene_check_hierarchy(one) == (1, 1, 1)
    many = np.concatenate([np.repeat(one, 300, 0), np.array([[2, -1, 1, 2, 1, 1, 2, 1, 1.0000001]], np.float32),
                           np.array([[4, 0, 0, 4, 0, 0, 4, 0, 0]], np.float32)])
    n_nodes, depth, max_leaf = afa.scene_check_hierarchy(many)
    assert depth <= 32 and max_leaf <= 4
    rng = np.random.default_rng(0)
    soup = rng.uniform(-50, 50, (5000, 9)).astype(np.float32)          # long, overlapping triangles
    n_nodes, depth, max_leaf = afa.scene_check_hierarchy(soup)
    assert depth <= 32
    with pytest.raises(afa.AfeError):
        afa.scene_check_hierarchy(np.full((1, 9), np.inf, np.float32))
