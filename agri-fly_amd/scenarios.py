"""Seeded synthetic ensembles shaped like BASELINE.json's configs (numpy only).

Value distributions follow SURVEY.md 8(d): positions U(-5,5)^2 x U(0,5) m,
velocities N(0,2) m/s, attitude uniform with tilt <= 60 deg, body rates N(0,2)
rad/s, motor speeds and commands U(0, w_max).  PRNG: numpy PCG64.
"""
import numpy as np


class EnsembleData:
    """Planar host arrays for n vehicles; `types` index into `type_ids`."""

    def __init__(self, n):
        self.n = n
        self.type_ids = [5]                      # QuadcopterType values, table order
        self.types = np.zeros(n, np.uint8)       # index into type_ids
        self.pos = np.zeros((3, n))
        self.vel = np.zeros((3, n))
        self.att = np.zeros((4, n))
        self.att[0] = 1.0
        self.ang_vel = np.zeros((3, n))
        self.motor_speed = np.zeros((4, n))
        self.motor_cmd = np.zeros((4, n), np.float32)
        self.ext_force = None
        self.ext_torque = None

    def slice(self, first, count):
        out = EnsembleData(count)
        out.type_ids = list(self.type_ids)
        sl = slice(first, first + count)
        for name in ("pos", "vel", "att", "ang_vel", "motor_speed", "motor_cmd"):
            setattr(out, name, np.ascontiguousarray(getattr(self, name)[:, sl]))
        out.types = np.ascontiguousarray(self.types[sl])
        for name in ("ext_force", "ext_torque"):
            a = getattr(self, name)
            setattr(out, name, None if a is None else np.ascontiguousarray(a[:, sl]))
        return out


def _quat_mul(a, b):
    """a * b, product convention of Rotation.hpp:124-131 (this = a, r1 = b)."""
    return np.stack([
        b[0] * a[0] - b[1] * a[1] - b[2] * a[2] - b[3] * a[3],
        b[1] * a[0] + b[0] * a[1] + b[3] * a[2] - b[2] * a[3],
        b[2] * a[0] - b[3] * a[1] + b[0] * a[2] + b[1] * a[3],
        b[3] * a[0] + b[2] * a[1] - b[1] * a[2] + b[0] * a[3]])


def random_attitudes(rng, n, max_tilt_deg=60.0):
    yaw = rng.uniform(-np.pi, np.pi, n)
    tilt = rng.uniform(0.0, np.deg2rad(max_tilt_deg), n)
    az = rng.uniform(-np.pi, np.pi, n)
    qyaw = np.stack([np.cos(yaw / 2), 0 * yaw, 0 * yaw, np.sin(yaw / 2)])
    qtilt = np.stack([np.cos(tilt / 2), np.sin(tilt / 2) * np.cos(az),
                      np.sin(tilt / 2) * np.sin(az), 0 * tilt])
    q = _quat_mul(qyaw, qtilt)
    return q / np.linalg.norm(q, axis=0)


def hover_speed(params):
    return float(np.sqrt(params.mass * 9.81 / (4.0 * params.prop_thrust_from_speed_sqr)))


def random_ensemble(n, seed, type_ids=(5, 1, 2, 4), max_speeds=None, with_wrench=True,
                    ground_fraction=0.05):
    """G1-style random states over several vehicle types.
    max_speeds: {type_id: w_max}; defaults to 3000 rad/s when not given."""
    rng = np.random.Generator(np.random.PCG64(seed))
    e = EnsembleData(n)
    e.type_ids = list(type_ids)
    e.types = rng.integers(0, len(type_ids), n).astype(np.uint8)
    e.pos[0] = rng.uniform(-5, 5, n)
    e.pos[1] = rng.uniform(-5, 5, n)
    e.pos[2] = rng.uniform(0, 5, n)
    e.vel = rng.normal(0, 2, (3, n))
    e.att = random_attitudes(rng, n)
    e.ang_vel = rng.normal(0, 2, (3, n))
    wmax = np.array([(max_speeds or {}).get(t, 3000.0) for t in type_ids])[e.types]
    e.motor_speed = rng.uniform(0, 1, (4, n)) * wmax
    e.motor_cmd = (rng.uniform(0, 1, (4, n)) * wmax).astype(np.float32)
    # a few vehicles touching / entering the ground (Quadcopter_T.cpp:146-151)
    k = int(n * ground_fraction)
    if k:
        idx = rng.choice(n, k, replace=False)
        e.pos[2, idx] = rng.uniform(0, 1e-4, k)
        e.vel[2, idx] = -np.abs(e.vel[2, idx]) - 0.5
    # a few with |w| dt below the one-arc-second identity threshold (Rotation.hpp:39,86)
    k2 = max(1, n // 50)
    idx2 = rng.choice(n, k2, replace=False)
    e.ang_vel[:, idx2] *= 1e-6
    # a few negative commands (Motor.cpp:48-50)
    idx3 = rng.choice(n, max(1, n // 50), replace=False)
    e.motor_cmd[0, idx3] = -100.0
    if with_wrench:
        e.ext_force = rng.normal(0, 0.2, (3, n))
        e.ext_torque = rng.normal(0, 1e-3, (3, n))
    return e


def hover_ensemble(n, params, height=3.5):
    """Config 2: every vehicle hovering at `height` m, motors at hover speed,
    commands held at hover speed (open loop)."""
    e = EnsembleData(n)
    e.pos[2] = height
    w = hover_speed(params)
    e.motor_speed[:] = w
    e.motor_cmd[:] = np.float32(w)
    return e


def gust_ensemble(n, params, seed=4, sigma_max=0.5, height=3.5, first_global=0, n_global=None, spacing=None):
    """Config 4: hovering ensemble with a per-vehicle constant wind-gust force
    F ~ N(0, sigma_i^2) per axis, sigma_i swept 0..sigma_max over the GLOBAL
    ensemble (so a shard [first_global, first_global+n) of a bigger ensemble
    reproduces exactly the rows an unsharded run would have).
    spacing: None = every vehicle hovers over the origin (independent Monte-Carlo replicas);
    a length = the vehicles share one world, on a square lattice of that pitch, 1024 per row, indexed
    by GLOBAL vehicle index (what a shared-world neighbour query needs to be meaningful)."""
    n_global = n if n_global is None else n_global
    e = hover_ensemble(n, params, height)
    idx = np.arange(first_global, first_global + n)
    if spacing is not None:
        e.pos[0] = (idx % 1024) * float(spacing)
        e.pos[1] = (idx // 1024) * float(spacing)
    sigma = sigma_max * idx / max(1, n_global - 1)
    # counter-based: vehicle i's gust depends only on (seed, i)
    f = np.empty((3, n))
    for axis in range(3):
        ss = np.random.SeedSequence([seed, axis])
        # Philox is counter-based: jump to this shard's offset
        bg = np.random.Philox(key=ss.generate_state(2, np.uint64))
        bg = bg.advance(first_global)
        # one normal per vehicle drawn from one uniform pair (Box-Muller) so the
        # stream position is exactly one counter block per vehicle
        gen = np.random.Generator(bg)
        u = gen.random((n, 4))  # one 4x64-bit Philox block per vehicle
        f[axis] = np.sqrt(-2.0 * np.log1p(-u[:, 0])) * np.cos(2 * np.pi * u[:, 1])
    e.ext_force = f * sigma
    return e


def synthetic_depth_image(width=320, height=240, seed=0, n_trunks=6, far_m=10.0, depth_scale=10.0 / 256.0,
                          focal_length=None, trunk_radius_m=(0.10, 0.25), trunk_range_m=(2.0, 9.0),
                          ground_height_m=1.5):
    """Config-3 stand-in for the AirSim DepthVis image (the Helios orchard is not in the
    reference tree): vertical cylinders ("trunks") in front of a
    pinhole camera (x right, y down, z forward, focal = width/2, principal point = centre;
    main.cpp:360,484-488) plus a ground plane `ground_height_m` below the camera, quantised
    like the reference's pipeline: counts = floor(z / depth_scale) clipped to 255 (8-bit
    DepthVis widened to uint16, main.cpp:121-122,352-354).  Seeded (numpy PCG64)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    f = width / 2.0 if focal_length is None else focal_length
    cx, cy = width / 2.0, height / 2.0
    u = (np.arange(width) - cx) / f            # ray direction x/z per column
    v = (np.arange(height) - cy) / f           # y/z per row
    depth = np.full((height, width), far_m)
    # ground plane y = ground_height_m (y points down): z = h / (y/z) for rows looking down
    with np.errstate(divide="ignore"):
        zg = np.where(v > 1e-9, ground_height_m / np.maximum(v, 1e-9), np.inf)
    depth = np.minimum(depth, zg[:, None])
    for _ in range(n_trunks):
        r = rng.uniform(*trunk_radius_m)
        zc = rng.uniform(*trunk_range_m)
        xc = rng.uniform(-0.6, 0.6) * zc       # inside the horizontal field of view
        # ray (u, ., 1) z hits the cylinder (x - xc)^2 + (z - zc)^2 = r^2: per column
        a = u * u + 1.0
        b = -2.0 * (u * xc + zc)
        c = xc * xc + zc * zc - r * r
        disc = b * b - 4 * a * c
        z_hit = np.where(disc >= 0, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), np.inf)
        z_hit = np.where(z_hit > 0.2, z_hit, np.inf)
        depth = np.minimum(depth, z_hit[None, :])
    counts = np.floor(depth / depth_scale)
    return np.clip(counts, 0, 255).astype(np.uint16)


def _icosphere(subdiv):
    t = (1.0 + np.sqrt(5.0)) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    verts = [np.array(p, float) / np.linalg.norm(p) for p in v]
    faces = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2),
             (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11),
             (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(subdiv):
        cache = {}

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = verts[a] + verts[b]
                verts.append(m / np.linalg.norm(m))
                cache[key] = len(verts) - 1
            return cache[key]

        nf = []
        for a, b, c in faces:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        faces = nf
    return np.array(verts), np.array(faces)


def orchard_mesh(rows=8, cols=8, row_spacing=4.0, tree_spacing=3.0, seed=0, trunk_sides=8, canopy_subdiv=1,
                 margin=10.0, jitter=0.3, return_layout=False):
    """Stand-in for the Helios orchard scene the reference renders through AirSim/Unity (not in
    the reference tree): a ground plane at z = 0 and rows x cols
    trees (a prism trunk and an ellipsoidal canopy each) on a jittered lattice, rows along +x.
    World frame z up, metres.  Returns float32 [n_tri, 9] (v0 v1 v2); with return_layout also the
    analytic trees, one row each: trunk x, y, radius, height, canopy centre xyz, canopy semi-axes xyz
    (the mesh is inscribed in these shapes).  Seeded (numpy PCG64)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    tris = []
    layout = []
    x1, y1 = (cols - 1) * tree_spacing + margin, (rows - 1) * row_spacing + margin
    g = np.array([[-margin, -margin, 0], [x1, -margin, 0], [x1, y1, 0], [-margin, y1, 0]], float)
    tris += [np.concatenate([g[0], g[1], g[2]]), np.concatenate([g[0], g[2], g[3]])]
    sv, sf = _icosphere(canopy_subdiv)
    ang = 2 * np.pi * np.arange(trunk_sides) / trunk_sides
    for r in range(rows):
        for c in range(cols):
            cx = c * tree_spacing + rng.uniform(-jitter, jitter)
            cy = r * row_spacing + rng.uniform(-jitter, jitter)
            tr = rng.uniform(0.08, 0.18)
            th = rng.uniform(1.2, 2.0)
            ring = np.stack([cx + tr * np.cos(ang), cy + tr * np.sin(ang)], 1)
            for k in range(trunk_sides):
                a, b = ring[k], ring[(k + 1) % trunk_sides]
                tris.append(np.array([a[0], a[1], 0, b[0], b[1], 0, b[0], b[1], th]))
                tris.append(np.array([a[0], a[1], 0, b[0], b[1], th, a[0], a[1], th]))
            rad = np.array([rng.uniform(0.8, 1.3), rng.uniform(0.8, 1.3), rng.uniform(0.9, 1.5)])
            centre = np.array([cx, cy, th + 0.7 * rad[2]])
            pv = sv * rad + centre
            for f in sf:
                tris.append(pv[f].reshape(9))
            layout.append([cx, cy, tr, th, *centre, *rad])
    mesh = np.asarray(tris, dtype=np.float32)
    return (mesh, np.asarray(layout)) if return_layout else mesh
