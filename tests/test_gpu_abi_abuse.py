"""The C ABI with LIVE handles and bad data arguments (the reference's plugin API validates at its boundary and so does this
one): NULL where a buffer is required, ranges that leave the ensemble, negative counts, cameras without pixels, planners
without images.  Every such call must come back with an AFE_ERR_* -- no crash, no hang, nothing written -- and the engine
must afterwards step exactly like one that was never abused.  Empty requests (count 0) are answered, not refused.  In a child
process (a crash is this test's failure); tools/host_sanitizers.sh runs it on the sanitized host build as well."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import ctypes as C, importlib, sys
import numpy as np
import torch  # noqa: F401
sys.path.insert(0, %r)
from tests.test_gpu_persistent import make, assert_same
afa = importlib.import_module("agri-fly_amd")
L = afa.library()
N = 1000
calls = [0]

wrong = []

def refused(rc, what):
    calls[0] += 1
    if rc == 0:
        wrong.append(what + ": accepted")

def answered(rc, what):
    calls[0] += 1
    if rc != 0:
        wrong.append(what + ": refused (%%d)" %% rc)

for persistent in (False, True):
    a, _ = make(N, afa.AFE_F32, persistent)          # abused
    b, _ = make(N, afa.AFE_F32, persistent)          # left alone
    a.step(1000, 7); b.step(1000, 7)
    h = a._h
    buf = np.zeros((16, N), np.float64)
    p = buf.ctypes.data
    # ---- state, commands, wrench, IMU, RNG words: (handle, first, count, pointers...)
    ranged = {"afe_set_state": 5, "afe_get_state": 5, "afe_set_state_f32": 5, "afe_get_state_f32": 5, "afe_set_rng_state": 1,
              "afe_get_rng_state": 1, "afe_set_motor_cmds": 1, "afe_get_motor_cmds": 1, "afe_set_external_force": 1,
              "afe_set_external_torque": 1, "afe_get_external_force": 1, "afe_get_imu": 2, "afe_set_vehicle_types": 1,
              "afe_set_commands_from_radio": 1}
    required = {"afe_set_rng_state", "afe_get_rng_state", "afe_set_motor_cmds", "afe_get_motor_cmds", "afe_get_external_force",
                "afe_set_vehicle_types", "afe_set_commands_from_radio"}
    skippable = {"afe_set_state", "afe_get_state", "afe_set_state_f32", "afe_get_state_f32", "afe_get_imu"}
    # (afe_set_external_force / _torque: NULL zeroes the range, by the header -- not tried on the ensemble under comparison)
    for name, n_ptr in ranged.items():
        fn = getattr(L, name)
        some = [p] * n_ptr
        if name in required:
            refused(fn(h, 0, 5, *([None] * n_ptr)), name + " with NULL buffers")
        elif name in skippable:                  # the header: "NULL field pointers are skipped"
            answered(fn(h, 0, 5, *([None] * n_ptr)), name + " with every field skipped")
        refused(fn(h, -1, 5, *some), name + " first = -1")
        refused(fn(h, N - 2, 5, *some), name + " past the end")
        refused(fn(h, 0, N + 1, *some), name + " count > n")
        refused(fn(h, 0, -3, *some), name + " count < 0")
        refused(fn(h, 2 ** 62, 2 ** 62, *some), name + " first + count overflows")
    refused(L.afe_set_rates_commands(h, 0, 5, None, None), "afe_set_rates_commands NULL")
    refused(L.afe_set_rates_commands(h, N - 1, 5, p, p), "afe_set_rates_commands past the end")
    # ---- tables, modes, numbers
    refused(L.afe_set_type_table(h, None, 1), "type table NULL")
    refused(L.afe_set_type_table(h, C.pointer(afa.params_from_type(5)), 0), "type table of 0 records")
    refused(L.afe_set_type_table(h, C.pointer(afa.params_from_type(5)), -1), "type table of -1 records")
    bad = afa.params_from_type(5); bad.mass = 0.0
    refused(L.afe_set_type_table(h, C.pointer(bad), 1), "mass 0")
    bad = afa.params_from_type(5); bad.mass = float("nan")
    refused(L.afe_set_type_table(h, C.pointer(bad), 1), "mass NaN")
    # (afe_set_rates_logic(NULL) is the header's "logic off", not an error)
    refused(L.afe_set_logic_period(h, -1.0), "negative logic period")
    refused(L.afe_set_logic_period(h, float("nan")), "NaN logic period")
    refused(L.afe_set_imu_noise(h, 1, -1.0, 0.2, afa.AFE_SEED_DECORRELATED), "negative noise")
    refused(L.afe_set_imu_noise(h, 1, 0.1, 0.2, 77), "unknown seed policy")
    refused(L.afe_set_step_mode(h, 99), "unknown step mode")
    refused(L.afe_set_cache_policy(h, 99), "unknown cache policy")
    refused(L.afe_set_addressing(h, 99), "unknown addressing")
    refused(L.afe_step(h, 1000, -1), "afe_step of -1 steps")
    refused(L.afe_steps_until_tick(h, 1000, None), "steps_until_tick NULL")
    refused(L.afe_time_us(h, None), "time NULL")
    refused(L.afe_logic_ticks(h, None), "ticks NULL")
    refused(L.afe_get_device_view(h, None), "device view NULL")
    refused(L.afe_checkpoint_size(h, None), "checkpoint size NULL")
    sz = C.c_uint64()
    answered(L.afe_checkpoint_size(h, C.byref(sz)), "checkpoint size")
    blob = np.zeros(sz.value, np.uint8)
    refused(L.afe_save_checkpoint(h, None, sz.value), "save to NULL")
    refused(L.afe_save_checkpoint(h, blob.ctypes.data, sz.value - 1), "save into a short buffer")
    refused(L.afe_load_checkpoint(h, blob.ctypes.data, sz.value), "load of zeros")
    refused(L.afe_load_checkpoint(h, None, sz.value), "load from NULL")
    # a real checkpoint with damaged header bytes or a cut-off tail: refused or (a flipped time stamp, say) taken -- never a
    # crash, never a read past the buffer (the sanitized build watches) -- and the undamaged one puts everything back
    answered(L.afe_save_checkpoint(h, blob.ctypes.data, sz.value), "save")
    good_blob = blob.copy()
    frng = np.random.default_rng(77)
    for trial in range(300):
        hurt = good_blob.copy()
        for at in frng.integers(0, 256, frng.integers(1, 5)):
            hurt[at] ^= np.uint8(1 << frng.integers(0, 8))
        L.afe_load_checkpoint(h, hurt.ctypes.data, sz.value)
        calls[0] += 1
    for cut in (0, 1, 16, 255, sz.value // 2, sz.value - 1):
        refused(L.afe_load_checkpoint(h, good_blob.ctypes.data, cut), "checkpoint cut to %%d bytes" %% cut)
    answered(L.afe_load_checkpoint(h, good_blob.ctypes.data, sz.value), "load of the undamaged checkpoint")
    # ---- shared-world queries
    idx = np.zeros(8, np.int64); d2 = np.zeros(8, np.float32)
    refused(L.afe_nearest_neighbour(h, None, 8, idx.ctypes.data, d2.ctypes.data), "nearest neighbour, NULL positions")
    refused(L.afe_nearest_neighbour_grid(h, None, 8, 1.0, idx.ctypes.data, d2.ctypes.data), "grid query, NULL positions")
    refused(L.afe_nearest_neighbour_bruteforce(h, None, 8, None, 8, None, None), "brute force, NULLs")
    refused(L.afe_pack_positions(h, None), "pack positions to NULL")
    # ---- empty requests are answered
    for name, n_ptr in ranged.items():
        if name in ("afe_set_vehicle_types", "afe_set_commands_from_radio"):
            continue
        answered(getattr(L, name)(h, 0, 0, *([p] * n_ptr)), name + " count 0")
    answered(L.afe_step(h, 1000, 0), "afe_step of 0 steps")
    # ---- and none of it left a trace
    a.step(1000, 30); b.step(1000, 30)
    assert_same(a, b, "after the abuse")
    a.close(); b.close()

# ---- depth camera
tris = afa.scenarios.orchard_mesh(rows=2, cols=2, seed=3)
refused(L.afe_scene_create(0, None, 5, C.byref(C.c_void_p())), "scene from NULL triangles")
refused(L.afe_scene_create(0, tris.ctypes.data, 0, C.byref(C.c_void_p())), "scene of 0 triangles")
refused(L.afe_scene_create(0, tris.ctypes.data, -4, C.byref(C.c_void_p())), "scene of -4 triangles")
refused(L.afe_scene_create(0, tris.ctypes.data, len(tris), None), "scene handle to NULL")
refused(L.afe_scene_create(99, tris.ctypes.data, len(tris), C.byref(C.c_void_p())), "scene on device 99")
scene = afa.Scene(tris)
s = scene._h
cam = afa.camera_default(64, 48)
pos = np.zeros((3, 4)); pos[2] = 1.0
att = np.zeros((4, 4)); att[0] = 1.0
out = np.zeros((4, 48, 64), np.uint16)
ms = C.c_float()
ref, _ = scene.render(cam, pos, att, afa.camera_default_mount())
refused(L.afe_render_depth(s, None, 4, pos.ctypes.data, att.ctypes.data, None, out.ctypes.data, None), "render without a camera")
refused(L.afe_render_depth(s, C.byref(cam), 4, None, att.ctypes.data, None, out.ctypes.data, None), "render without positions")
refused(L.afe_render_depth(s, C.byref(cam), 4, pos.ctypes.data, None, None, out.ctypes.data, None), "render without attitudes")
refused(L.afe_render_depth(s, C.byref(cam), 4, pos.ctypes.data, att.ctypes.data, None, None, None), "render to NULL")
refused(L.afe_render_depth(s, C.byref(cam), -1, pos.ctypes.data, att.ctypes.data, None, out.ctypes.data, None), "render of -1 views")
for field, value in (("width", 0), ("height", -3), ("focal_length", 0.0), ("depth_scale", -1.0), ("max_count", 0), ("max_count", 70000)):
    c2 = afa.camera_default(64, 48)
    setattr(c2, field, value)
    refused(L.afe_render_depth(s, C.byref(c2), 4, pos.ctypes.data, att.ctypes.data, None, out.ctypes.data, None), "camera with %%s = %%r" %% (field, value))
refused(L.afe_render_depth_stats(s, C.byref(cam), 4, pos.ctypes.data, att.ctypes.data, None, None, None), "stats to NULL")
e, _ = make(N, afa.AFE_F32, False)
refused(L.afe_render_depth_engine(e._h, s, C.byref(cam), N - 2, 5, None, out.ctypes.data, 0, None), "engine views past the end")
refused(L.afe_render_depth_engine(e._h, s, C.byref(cam), -1, 2, None, out.ctypes.data, 0, None), "engine views from -1")
refused(L.afe_render_depth_engine(e._h, None, C.byref(cam), 0, 2, None, out.ctypes.data, 0, None), "engine views without a scene")
refused(L.afe_render_depth_engine(None, s, C.byref(cam), 0, 2, None, out.ctypes.data, 0, None), "engine views without an engine")
answered(L.afe_render_depth(s, C.byref(cam), 0, pos.ctypes.data, att.ctypes.data, None, out.ctypes.data, None), "render of 0 views")
answered(L.afe_render_depth_engine(e._h, s, C.byref(cam), 0, 0, None, out.ctypes.data, 0, None), "render of 0 engine views")
refused(L.afe_render_depth_engine(e._h, s, C.byref(cam), 2 ** 62, 2 ** 62, None, out.ctypes.data, 0, None), "engine views whose range wraps")
again, _ = scene.render(cam, pos, att, afa.camera_default_mount())
assert np.array_equal(ref, again)
e.close()

# ---- planner
cfg = afa.planner_default_config(64, 48, 10.0 / 256.0, 32.0, 0.116, 0.174, 0.5)
samples = afa.planner_samples(0, 64, 48, 16)
img = np.full((2, 48, 64), 200, np.uint16)
z = np.zeros((3, 2))
good = afa.plans_as_array(afa.rappids_plan(cfg, img, z, z, z - [[0], [0], [9.81]], samples)[0])
PO = afa.PlanOutput if hasattr(afa, "PlanOutput") else None
outbuf = np.zeros(4096, np.uint8)
def plan(n=2, images=img.ctypes.data, n_images=2, vel=z.ctypes.data, acc=z.ctypes.data, grav=z.ctypes.data, smp=samples.ctypes.data,
         n_cand=16, out=outbuf.ctypes.data, config=cfg):
    return L.afe_rappids_plan(0, C.byref(config) if config is not None else None, n, images, n_images, None, vel, acc, grav, None, smp, 1, None,
                              n_cand, out, None, None)
refused(plan(config=None), "plan without a configuration")
refused(plan(images=None), "plan without images")
refused(plan(vel=None), "plan without velocities")
refused(plan(grav=None), "plan without gravity")
refused(plan(smp=None), "plan without samples")
refused(plan(out=None), "plan to NULL")
refused(plan(n=-1), "plan of -1 planners")
refused(plan(n_images=1), "two planners, one image, no index")
refused(plan(n_cand=0), "plan with 0 candidates")
refused(plan(n_cand=-5), "plan with -5 candidates")
answered(plan(n=0), "plan of 0 planners")
again = afa.plans_as_array(afa.rappids_plan(cfg, img, z, z, z - [[0], [0], [9.81]], samples)[0])
assert good.tobytes() == again.tobytes()

# ---- groups of shards on this one device, engines, events
g = C.c_void_p()
dev0 = (C.c_int * 2)(0, 0)
refused(L.afe_group_create(C.byref(g), 1000, afa.AFE_F32, None, 2), "group without devices")
refused(L.afe_group_create(C.byref(g), 1000, afa.AFE_F32, dev0, 0), "group of 0 shards")
refused(L.afe_group_create(C.byref(g), 1000, afa.AFE_F32, dev0, -2), "group of -2 shards")
refused(L.afe_group_create(C.byref(g), 0, afa.AFE_F32, dev0, 2), "group of 0 vehicles")
refused(L.afe_group_create(C.byref(g), -5, afa.AFE_F32, dev0, 2), "group of -5 vehicles")
refused(L.afe_group_create(C.byref(g), 1000, 7, dev0, 2), "group of an unknown precision")
refused(L.afe_group_create(None, 1000, afa.AFE_F32, dev0, 2), "group handle to NULL")
bad_dev = (C.c_int * 2)(0, 99)
refused(L.afe_group_create(C.byref(g), 1000, afa.AFE_F32, bad_dev, 2), "group with device 99")
answered(L.afe_group_create(C.byref(g), 1000, afa.AFE_F32, dev0, 2), "group of two shards on device 0")
sh = C.c_void_p(); f0 = C.c_int64(); c0 = C.c_int64()
refused(L.afe_group_shard(g, 2, C.byref(sh), C.byref(f0), C.byref(c0)), "shard 2 of 2")
refused(L.afe_group_shard(g, -1, C.byref(sh), C.byref(f0), C.byref(c0)), "shard -1")
answered(L.afe_group_shard(g, 1, C.byref(sh), C.byref(f0), C.byref(c0)), "shard 1 of 2")
assert f0.value + c0.value == 1000
refused(L.afe_group_step(g, 1000, -1), "group step of -1")
answered(L.afe_group_gather_positions(g, None), "group gather without the optional pointer table")
answered(L.afe_group_destroy(g), "group destroy")
h2 = C.c_void_p()
refused(L.afe_create(C.byref(h2), 0, afa.AFE_F32, 0, 0), "engine of 0 vehicles")
refused(L.afe_create(C.byref(h2), -1, afa.AFE_F32, 0, 0), "engine of -1 vehicles")
refused(L.afe_create(C.byref(h2), 1000, 9, 0, 0), "engine of an unknown precision")
refused(L.afe_create(C.byref(h2), 1000, afa.AFE_F32, 99, 0), "engine on device 99")
refused(L.afe_create(C.byref(h2), 1000, afa.AFE_F32, 0, -1), "engine with a negative global index")
refused(L.afe_create(None, 1000, afa.AFE_F32, 0, 0), "engine handle to NULL")
refused(L.afe_create(C.byref(h2), 2 ** 40, afa.AFE_F32, 0, 0), "engine of 2^40 vehicles")
ev = C.c_void_p()
answered(L.afe_event_create(C.byref(ev)), "event")
refused(L.afe_event_elapsed_ms(ev, ev, None), "elapsed to NULL")
refused(L.afe_event_record(None, ev), "record without an engine")
answered(L.afe_event_destroy(ev), "event destroy")

# ---- UWB, scratch
u = C.c_void_p()
answered(L.afe_uwb_create(C.byref(u)), "uwb create")
refused(L.afe_uwb_draw(u, 5, None, None), "uwb draw to NULL")
refused(L.afe_uwb_draw(u, -1, d2.ctypes.data, d2.ctypes.data), "uwb draw of -1")
# (afe_uwb_set_noise takes any three numbers, like UWBNetwork::SetNoiseProperties, UWBNetwork.hpp:28-33)
L.afe_uwb_destroy(u)
refused(L.afe_device_alloc(0, 8, None), "device alloc to NULL")
refused(L.afe_device_alloc(99, 8, C.byref(C.c_void_p())), "device alloc on device 99")
refused(L.afe_device_download(None, None, 8), "download NULL")
scene.close()
assert not wrong, "\n".join(wrong)
print("ok", calls[0], "calls")
''' % ROOT


def test_bad_arguments_are_refused_and_leave_no_trace():
    out = subprocess.run([sys.executable, "-c", CODE], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, (out.returncode, out.stdout[-800:], out.stderr[-3000:])
