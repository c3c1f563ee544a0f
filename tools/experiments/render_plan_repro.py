"""render (from engine state, into a device buffer) -> plan, over and over on fixed states while a second process keeps the
GPU busy: image hash and winning candidate must never change."""
import hashlib, importlib, os, subprocess, sys, time
import numpy as np
import torch  # noqa
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
afa = importlib.import_module("agri-fly_amd")
tris = afa.scenarios.orchard_mesh(rows=4, cols=8, seed=3)
tris = (tris.reshape(-1, 3, 3) + np.array([5.0, -2.0, 0.0])).reshape(-1, 9).astype(np.float32)
scene = afa.Scene(tris)
cam, mount = afa.camera_default(320, 240), afa.camera_default_mount()
p = afa.params_from_type(5)
cfg = afa.planner_default_config(320, 240, cam.depth_scale, cam.focal_length, 2 * p.arm_length, 3 * p.arm_length, 0.5)
cfg.cost_type = 1
samples = afa.planner_samples(0, 320, 240, 192)
rng = np.random.default_rng(1)
n = 32
pos = np.stack([rng.uniform(0, 25, n), rng.uniform(-1.5, 1.5, n), rng.uniform(0.3, 1.5, n)])
yaw = rng.uniform(-0.5, 0.5, n)
att = np.stack([np.cos(yaw / 2), 0 * yaw, 0 * yaw, np.sin(yaw / 2)])
host_visible = os.environ.get("REPRO_HOST_VISIBLE", "1") == "1"
e = afa.Ensemble(1, host_visible=host_visible)
e.set_type_table([p])
buf = afa.DeviceBuffer(240 * 320 * 2)
v = np.array([[0.0], [0.0], [1.0]]); z = np.zeros((3, 1)); g = np.array([[0.0], [9.81], [0.0]]); goal = np.array([[0.0], [0.0], [30.0]])
bg = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "experiments", "gpu_load.py"), os.environ.get("REPRO_LOAD_S", "120")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
time.sleep(20)
ref_img, ref_plan, bad_img, bad_plan = {}, {}, 0, 0
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for k in range(reps):
    i = k % n
    e.set_state(pos=pos[:, i:i + 1], att=att[:, i:i + 1], vel=np.zeros((3, 1)), ang_vel=np.zeros((3, 1)), motor_speed=np.zeros((4, 1)))
    scene.render_engine(e, cam, mount, out=buf)
    out, _, _ = afa.rappids_plan(cfg, buf, v, z, g, samples, cost_vec=goal)
    img = buf.download(np.uint16, (240, 320))
    h = hashlib.sha256(img.tobytes()).hexdigest()[:12]
    pl = (out[0].found, out[0].best_index, out[0].n_pyramids, out[0].n_collision_checks)
    if i in ref_img and ref_img[i] != h:
        bad_img += 1; print("view %d: image differs at repetition %d" % (i, k))
    if i in ref_plan and ref_plan[i] != pl:
        bad_plan += 1; print("view %d: plan differs at repetition %d: %s vs %s (image same: %s)" % (i, k, pl, ref_plan[i], ref_img.get(i) == h))
    ref_img.setdefault(i, h); ref_plan.setdefault(i, pl)
    if bg.poll() is not None:
        bg = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "experiments", "gpu_load.py"), os.environ.get("REPRO_LOAD_S", "120")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
print("%d render -> plan rounds: %d differing images, %d differing plans" % (reps, bad_img, bad_plan))
bg.wait()
