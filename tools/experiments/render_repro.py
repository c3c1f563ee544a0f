"""Does the depth camera give the same bits every time?  The same 64 poses rendered over and over while a second process
keeps the GPU busy; every image hashed."""
import hashlib, importlib, os, subprocess, sys
import numpy as np
import torch  # noqa
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
afa = importlib.import_module("agri-fly_amd")
tris = afa.scenarios.orchard_mesh(rows=4, cols=8, seed=3)
tris = (tris.reshape(-1, 3, 3) + np.array([5.0, -2.0, 0.0])).reshape(-1, 9).astype(np.float32)
scene = afa.Scene(tris)
cam, mount = afa.camera_default(320, 240), afa.camera_default_mount()
rng = np.random.default_rng(1)
n = 64
pos = np.stack([rng.uniform(0, 25, n), rng.uniform(-1.5, 1.5, n), rng.uniform(0.3, 1.5, n)])
yaw = rng.uniform(-0.5, 0.5, n)
att = np.stack([np.cos(yaw / 2), 0 * yaw, 0 * yaw, np.sin(yaw / 2)])
bg = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "experiments", "gpu_load.py"), os.environ.get("REPRO_LOAD_S", "120")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
import time
time.sleep(float(os.environ.get("REPRO_WAIT", "20")))      # the load has to be on the GPU before the renders begin
ref = None
bad = 0
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for k in range(reps):
    # one view per call, like the headless loop
    i = k % n
    img, _ = scene.render(cam, pos[:, i:i + 1], att[:, i:i + 1], mount)
    h = hashlib.sha256(np.asarray(img).tobytes()).hexdigest()[:12]
    if ref is None:
        ref = {}
    if i in ref and ref[i] != h:
        bad += 1
        print("view %d differs at repetition %d: %s vs %s" % (i, k, h, ref[i]))
    ref.setdefault(i, h)
    if bg.poll() is not None:
        bg = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "experiments", "gpu_load.py"), os.environ.get("REPRO_LOAD_S", "120")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
print("%d renders, %d differing" % (reps, bad))
bg.wait()
