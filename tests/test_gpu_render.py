"""Depth camera on the GPU (BASELINE config 5 shape) against the CPU checker:
the engine walks a BVH in its own traversal order, the checker tests every triangle, and the two
must produce IDENTICAL uint16 images (same fp64 operations per ray/triangle pair, so even the
floor() ties agree).  Also: poses taken from the engine's device state, and render -> plan kept
on the device.  Needs an MI355X."""
import numpy as np
import pytest

from tests.scenarios import MEASUREMENTS, afa

pytestmark = pytest.mark.gpu

scen = afa.scenarios
AFE_F32, AFE_F64 = afa.AFE_F32, afa.AFE_F64


def _ocam(ora, cam):
    c = ora.render_camera(cam.width, cam.height, cam.focal_length, cam.depth_scale, cam.max_count)
    c.cx, c.cy = cam.cx, cam.cy
    return c


def _poses(rng, n, tris, max_tilt_deg=25.0):
    lo, hi = tris.reshape(-1, 3).min(0), tris.reshape(-1, 3).max(0)
    pos = np.stack([rng.uniform(lo[0] * 0.5, hi[0] * 0.8, n), rng.uniform(lo[1] * 0.5, hi[1] * 0.8, n),
                    rng.uniform(0.4, 3.0, n)])
    att = scen.random_attitudes(rng, n, max_tilt_deg=max_tilt_deg)
    return pos, att


def _check(ora, scene, cam, tris, pos, att, mount):
    got, ms = scene.render(cam, pos, att, mount)
    oc = _ocam(ora, cam)
    m = (1.0, 0.0, 0.0, 0.0) if mount is None else mount
    for i in range(pos.shape[1]):
        want = ora.render_depth(oc, tris, pos[:, i], att[:, i], m)
        np.testing.assert_array_equal(got[i], want, err_msg="view %d" % i)
    return got, ms


def test_orchard_images_identical_to_checker(ora):
    tris = scen.orchard_mesh(rows=6, cols=8, seed=3)
    scene = afa.Scene(tris)
    info = scene.info()
    assert info["n_tri"] == len(tris) and info["depth"] <= 32 and info["n_nodes"] > len(tris) // 4
    cam = afa.camera_default(96, 72)
    assert (cam.focal_length, cam.cx, cam.cy, cam.depth_scale, cam.max_count) == (48.0, 48.0, 36.0, 10.0 / 256.0, 255)
    mount = afa.camera_default_mount()
    pos, att = _poses(np.random.default_rng(11), 10, tris)
    got, _ = _check(ora, scene, cam, tris, pos, att, mount)
    frac_hit = np.mean(got < 255)
    assert 0.2 < frac_hit < 0.98                     # trunks, canopies, ground ... and open sky / beyond 10 m


def test_odd_sizes_focal_and_identity_mount(ora):
    tris = scen.orchard_mesh(rows=2, cols=3, seed=9, canopy_subdiv=2)
    scene = afa.Scene(tris)
    rng = np.random.default_rng(5)
    for (w, h, f) in ((37, 23, 20.5), (16, 4, 8.0), (1, 1, 1.0), (130, 5, 200.0)):
        cam = afa.camera_default(w, h)
        cam.focal_length = f
        cam.cx, cam.cy = 0.37 * w, 0.61 * h
        cam.depth_scale, cam.max_count = 0.05, 1000
        pos, att = _poses(rng, 3, tris, max_tilt_deg=80.0)
        _check(ora, scene, cam, tris, pos, att, None)


def test_degenerate_meshes(ora):
    cam = afa.camera_default(40, 30)
    mount = afa.camera_default_mount()
    pos = np.array([[0.0], [0.0], [1.0]])
    att = np.array([[1.0], [0.0], [0.0], [0.0]])
    one = np.array([[3, -1, 0, 3, 1, 0, 3, 0, 2]], np.float32)
    _check(ora, afa.Scene(one), cam, one, pos, att, mount)
    # 300 coincident copies + slivers + a zero-area triangle: centroids coincide, the builder must
    # fall back to median splits and stay inside the traversal stack
    many = np.concatenate([np.repeat(one, 300, 0), np.array([[2, -1, 1, 2, 1, 1, 2, 1, 1.0000001]], np.float32),
                           np.array([[4, 0, 0, 4, 0, 0, 4, 0, 0]], np.float32)])
    s = afa.Scene(many)
    assert s.info()["depth"] <= 32
    _check(ora, s, cam, many, pos, att, mount)
    # camera inside a closed box: every pixel hits
    lo, hi = np.array([-2, -2, 0.0]), np.array([2, 2, 2.5])
    c = np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])], float)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    box = np.array([np.concatenate([c[a], c[b], c[d]]) for a, b, d, e in quads] +
                   [np.concatenate([c[a], c[d], c[e]]) for a, b, d, e in quads], np.float32)
    # (off-centre, so that no pixel's ray runs exactly into a box corner: the two-sided edge rule
    # may let such a ray through, in the checker and the engine alike)
    got, _ = _check(ora, afa.Scene(box), cam, box, pos + [[0.1], [0.05], [0.0]], att, mount)
    assert got.max() < 255
    with pytest.raises(afa.AfeError):
        afa.Scene(np.full((1, 9), np.nan, np.float32))


@pytest.mark.parametrize("precision", [AFE_F64, AFE_F32])
def test_poses_from_engine_state(ora, precision):
    tris = scen.orchard_mesh(rows=4, cols=4, seed=21)
    scene = afa.Scene(tris)
    cam = afa.camera_default(64, 48)
    mount = afa.camera_default_mount()
    n = 300                                            # not a multiple of anything: exercises the slab stride
    pos, att = _poses(np.random.default_rng(2), n, tris)
    e = afa.Ensemble(n, precision=precision)
    e.set_type_table([afa.params_from_type(5)])
    e.set_state(pos, np.zeros((3, n)), att, np.zeros((3, n)), np.zeros((4, n)))
    first, count = 17, 9
    imgs, ms = scene.render_engine(e, cam, mount, first=first, count=count)
    st = e.get_state(first, count)                     # what the device holds (rounded to fp32 for AFE_F32)
    oc = _ocam(ora, cam)
    for i in range(count):
        want = ora.render_depth(oc, tris, st["pos"][:, i], st["att"][:, i], mount)
        np.testing.assert_array_equal(imgs[i], want)
    if precision == AFE_F64:
        host, _ = scene.render(cam, pos[:, first:first + count], att[:, first:first + count], mount)
        np.testing.assert_array_equal(imgs, host)
    # after stepping the vehicles have moved and so have their images
    e.set_motor_cmds(np.zeros((4, n), np.float32))
    e.step(1000, 200)
    moved, _ = scene.render_engine(e, cam, mount, first=first, count=count)
    st2 = e.get_state(first, count)
    assert np.all(st2["pos"][2] < st["pos"][2])        # free fall
    for i in range(count):
        np.testing.assert_array_equal(moved[i], ora.render_depth(oc, tris, st2["pos"][:, i], st2["att"][:, i], mount))


def test_render_then_plan_stays_on_device(ora):
    """step -> render -> plan with the images never leaving HBM equals the same chain through host
    buffers, and the checker's own chain (oracle renderer -> oracle planner)."""
    tris = scen.orchard_mesh(rows=5, cols=6, seed=33)
    scene = afa.Scene(tris)
    cam = afa.camera_default(320, 240)
    mount = afa.camera_default_mount()
    n, m = 6, 64
    rng = np.random.default_rng(8)
    pos = np.stack([rng.uniform(-4, -1, n), rng.uniform(0, 14, n), rng.uniform(1.0, 2.0, n)])
    yaw = rng.uniform(-0.3, 0.3, n)
    att = np.stack([np.cos(yaw / 2), 0 * yaw, 0 * yaw, np.sin(yaw / 2)])
    e = afa.Ensemble(n, precision=AFE_F64)
    e.set_type_table([afa.params_from_type(5)])
    e.set_state(pos, np.zeros((3, n)), att, np.zeros((3, n)), np.zeros((4, n)))
    buf = afa.DeviceBuffer(n * 240 * 320 * 2)
    scene.render_engine(e, cam, mount, out=buf)
    host_imgs, _ = scene.render(cam, pos, att, mount)
    np.testing.assert_array_equal(buf.download(np.uint16, (n, 240, 320)), host_imgs)

    ocfg = ora.planner_config(320, 240, cam.depth_scale, cam.focal_length, 0.116, 0.174, 0.5)
    ocfg.max_pyramids = 64
    cfg = afa.planner_default_config(320, 240, cam.depth_scale, cam.focal_length, 0.116, 0.174, 0.5)
    vel0 = np.stack([np.zeros(n), np.zeros(n), rng.uniform(0.0, 1.5, n)])
    acc0 = np.zeros((3, n))
    grav = np.tile(np.array([[0.0], [9.81], [0.0]]), (1, n))
    samples = ora.planner_samples(4, 320, 240, m)
    out_dev, flags_dev, _ = afa.rappids_plan(cfg, buf, vel0, acc0, grav, samples, want_flags=True)
    out_host, flags_host, _ = afa.rappids_plan(cfg, host_imgs, vel0, acc0, grav, samples, want_flags=True)
    np.testing.assert_array_equal(flags_dev, flags_host)
    oc = _ocam(ora, cam)
    n_found = 0
    for i in range(n):
        assert (out_dev[i].found, out_dev[i].best_index) == (out_host[i].found, out_host[i].best_index)
        want_img = ora.render_depth(oc, tris, pos[:, i], att[:, i], mount) if i < 2 else host_imgs[i]
        res, rflags = ora.planner_run(ocfg, want_img, vel0[:, i], acc0[:, i], grav[:, i], samples)
        assert (out_dev[i].found, out_dev[i].best_index) == (res.found, res.best_index)
        np.testing.assert_array_equal(flags_dev[i], rflags)
        n_found += res.found
    assert n_found >= 1
    buf.close()


def test_config5_shape_throughput():
    """320x240 DepthVis for a batch of vehicles over a 32x32-tree orchard (~1e5 triangles): sky-only
    views saturate, results do not depend on the batch they were rendered in, and the rate is reported."""
    tris = scen.orchard_mesh(rows=32, cols=32, seed=1)
    scene = afa.Scene(tris)
    info = scene.info()
    cam = afa.camera_default(320, 240)
    mount = afa.camera_default_mount()
    n = 256
    pos, att = _poses(np.random.default_rng(4), n, tris)
    imgs, ms = scene.render(cam, pos, att, mount)
    rays = n * 320 * 240
    print("\nconfig-5 shape: %d views x 320x240 over %d triangles (%d BVH nodes, depth %d): %.2f ms, %.3g rays/s"
          % (n, info["n_tri"], info["n_nodes"], info["depth"], ms, rays / (ms * 1e-3)))
    part, _ = scene.render(cam, pos[:, 100:103], att[:, 100:103], mount)
    np.testing.assert_array_equal(part, imgs[100:103])
    up = np.array([[np.cos(-np.pi / 4)], [0.0], [np.sin(-np.pi / 4)], [0.0]])   # pitched 90 deg nose-up
    sky, _ = scene.render(cam, np.array([[10.0], [10.0], [30.0]]), up, mount)
    assert np.all(sky == 255)
    assert 0.2 < np.mean(imgs < 255) < 0.99


def test_ordered_and_plain_walk_give_the_same_images_on_thousands_of_views():
    """Two formulations of the traversal -- the ordered walk of the octant-mirrored trees (near face = lo, nearer
    child stored first, slack folded into the ray factors) and the plain sign-agnostic walk of the unmirrored
    tree (per-lane min / max, explicit slack) -- must agree on every pixel: 4 096 views from anywhere in and
    above the orchard, every attitude (all eight direction octants, tiles straddling the coordinate planes,
    views straight down and straight up), 3.1e8 rays.  Each is separately compared with the brute-force checker
    on a few views elsewhere in this file; this is the wide net for a box test that culls what it must not."""
    tris = scen.orchard_mesh(rows=16, cols=16, seed=11)
    scene = afa.Scene(tris)
    cam = afa.camera_default(320, 240)
    mount = afa.camera_default_mount()
    rng = np.random.default_rng(21)
    n = 4096
    pos = np.stack([rng.uniform(-10, 60, n), rng.uniform(-10, 60, n), rng.uniform(0.3, 9.0, n)])
    q = rng.normal(size=(4, n))
    q /= np.linalg.norm(q, axis=0)
    # a share of exactly axis-aligned attitudes: direction components that are exactly zero (inf / NaN slabs)
    k = n // 8
    yaw = rng.integers(0, 4, k) * (np.pi / 2)
    q[:, :k] = np.stack([np.cos(yaw / 2), 0 * yaw, 0 * yaw, np.sin(yaw / 2)])
    pos[:, :k] = np.round(pos[:, :k])          # and origins on whole metres (o * inv = 0 * inf)
    ident = np.array([1.0, 0.0, 0.0, 0.0])
    for m in (mount, ident):
        scene.set_walk(False)
        a, ms_a = scene.render(cam, pos, q, m)
        scene.set_walk(True)
        b, ms_b = scene.render(cam, pos, q, m)
        scene.set_walk(False)
        assert np.array_equal(a, b)
        assert a.min() < 255 and (a == 255).any()
    MEASUREMENTS["render_walks_4096_views"] = {"ordered_ms": ms_a, "plain_ms": ms_b, "rays": n * 76800}
    # and the checker on the class of views that once broke BOTH walks: an axis-aligned camera on whole-metre
    # coordinates has pixels whose ray direction has a component of exactly zero -- 1/d infinite, slab
    # distances inf - inf = NaN, and a min / max chain that took -inf from the other face shut boxes the ray
    # was inside of (found by this test; |1/d| is capped in the kernel now)
    from oracle import oracle_py as ora
    oc = _ocam(ora, cam)
    for v in (0, 1, 6, 20, 21):
        for m in (mount, ident):
            want = ora.render_depth(oc, tris, pos[:, v], q[:, v], m)
            for plain in (False, True):
                scene.set_walk(plain)
                got, _ = scene.render(cam, pos[:, v:v + 1], q[:, v:v + 1], m)
                np.testing.assert_array_equal(got[0], want)
    scene.set_walk(False)
    scene.close()


def test_a_camera_at_a_non_finite_position_sees_nothing_and_holds_nobody_up(ora):
    """A vehicle that has diverged (NaN / inf position) still gets its image rendered with everybody's.  By the checker's own
    arithmetic such a camera hits nothing -- and it must not cost more than a view that does: the NaN passes every box test
    (max / min drop a NaN operand), and before round 6's guard each of its tiles walked the whole tree, 1 150 views' worth of
    time for one view (tools/experiments/nan_pose_probe.py)."""
    tris = scen.orchard_mesh(rows=12, cols=12, seed=2)
    scene = afa.Scene(tris)
    cam = afa.camera_default(160, 120)
    mount = afa.camera_default_mount()
    rng = np.random.default_rng(9)
    n = 128
    lo, hi = tris.reshape(-1, 3).min(0), tris.reshape(-1, 3).max(0)
    pos = np.stack([rng.uniform(lo[0] + 2, hi[0] - 2, n), rng.uniform(lo[1] + 2, hi[1] - 2, n), rng.uniform(0.5, 3.0, n)])
    att = scen.random_attitudes(rng, n, max_tilt_deg=20.0)
    ref, _ = scene.render(cam, pos, att, mount)
    base = min(scene.render(cam, pos, att, mount)[1] for _ in range(5))
    ocam = ora.render_camera(160, 120)
    for value, where in ((np.nan, 0), (np.inf, 2), (-np.inf, 1)):
        p = pos.copy()
        p[where, 5] = value
        p[where, 77] = value
        for walk in (0, 1):
            scene.set_walk(walk)
            img, _ = scene.render(cam, p, att, mount)
            ms = min(scene.render(cam, p, att, mount)[1] for _ in range(5))
            keep = ~np.isin(np.arange(n), (5, 77))
            assert np.array_equal(img[keep], ref[keep])
            assert (img[5] == 255).all() and (img[77] == 255).all()
            if walk == 0:
                assert ms < 2.0 * base + 0.2, (value, ms, base)
        scene.set_walk(0)
        assert np.array_equal(ora.render_depth(ocam, tris[:2000], p[:, 5], att[:, 5], mount), np.full((120, 160), 255, np.uint16))
    scene.close()
