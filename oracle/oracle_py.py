"""ctypes binding of oracle/libagrifly_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module, and only as the checker / the reported CPU baseline.  The product
path (agri-fly_amd/) never imports anything from oracle/.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libagrifly_oracle.so")


class OraParams(C.Structure):
    _fields_ = [
        ("mass", C.c_double),
        ("inertia", C.c_double * 9),
        ("inertia_inv", C.c_double * 9),
        ("motor_pos", (C.c_double * 3) * 4),
        ("motor_rot_axis", (C.c_double * 3) * 4),
        ("motor_thrust_axis", (C.c_double * 3) * 4),
        ("motor_min_speed", C.c_double),
        ("motor_max_speed", C.c_double),
        ("k_thrust", C.c_double),
        ("k_torque", C.c_double),
        ("motor_time_const", C.c_double),
        ("motor_inertia", C.c_double),
        ("lin_drag", C.c_double * 3),
        ("R_imu_inv", C.c_float * 9),
        ("sigma_acc", C.c_double),
        ("sigma_gyro", C.c_double),
    ]


class OraState(C.Structure):
    _fields_ = [
        ("pos", C.c_double * 3),
        ("vel", C.c_double * 3),
        ("att", C.c_double * 4),
        ("ang_vel", C.c_double * 3),
        ("motor_speed", C.c_double * 4),
        ("rng", C.c_uint32),
    ]


class OraClock(C.Structure):
    _fields_ = [
        ("now_us", C.c_uint64),
        ("integ_reset_us", C.c_uint64),
        ("logic_reset_us", C.c_uint64),
        ("logic_period", C.c_double),
    ]


def build(force=False):
    """Compile the checker (and the _ref probes when /root/reference exists)."""
    if force and os.path.exists(_LIB_PATH):
        os.remove(_LIB_PATH)
    # make decides staleness (every .c / .h of the checker is a prerequisite of the library)
    # make's chatter goes to stderr: bench.py's stdout carries exactly one JSON line
    subprocess.check_call(["make", "-s", "-C", _HERE, "libagrifly_oracle.so"], stdout=sys.stderr)
    subprocess.check_call(["make", "-s", "-C", _HERE, "ref"], stdout=sys.stderr)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()   # a no-op when up to date; an edited planner / render / logic .c never leaves a stale checker
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        fp = C.POINTER(C.c_float)
        L.ora_params_init.argtypes = [C.POINTER(OraParams), C.c_double, dp, C.c_double, dp,
                                      C.c_double, C.c_double, C.c_double, C.c_double,
                                      C.c_double, C.c_double, dp, C.c_float, C.c_float, C.c_float]
        L.ora_params_init.restype = None
        L.ora_params_from_type.argtypes = [C.POINTER(OraParams), C.c_int]
        L.ora_params_from_type.restype = C.c_int
        L.ora_type_from_id.argtypes = [C.c_uint]
        L.ora_type_from_id.restype = C.c_int
        L.ora_state_init.argtypes = [C.POINTER(OraState)]
        L.ora_state_init.restype = None
        L.ora_quad_step.argtypes = [C.POINTER(OraParams), C.POINTER(OraState), fp, dp, dp,
                                    C.c_double, C.c_int, fp, fp, dp]
        L.ora_quad_step.restype = None
        L.ora_motor_run.argtypes = [C.POINTER(OraParams), C.c_int, C.c_double, C.c_double,
                                    C.c_double, dp, dp, dp, dp]
        L.ora_motor_run.restype = C.c_double
        L.ora_minstd_next.argtypes = [C.POINTER(C.c_uint32)]
        L.ora_minstd_next.restype = C.c_uint32
        L.ora_canonical.argtypes = [C.POINTER(C.c_uint32)]
        L.ora_canonical.restype = C.c_double
        L.ora_normal_pair.argtypes = [C.POINTER(C.c_uint32), dp, dp]
        L.ora_normal_pair.restype = None
        for name, nin in (("ora_rot_matrix", 4), ("ora_rot_from_rotvec", 3), ("ora_rot_to_euler_ypr", 4)):
            getattr(L, name).argtypes = [dp, dp]
            getattr(L, name).restype = None
        for name in ("ora_rot_mul", "ora_rotate", "ora_rotate_inv"):
            getattr(L, name).argtypes = [dp, dp, dp]
            getattr(L, name).restype = None
        L.ora_rot_from_euler_ypr.argtypes = [C.c_double, C.c_double, C.c_double, dp]
        L.ora_rot_from_euler_ypr.restype = None
        L.ora_clock_init.argtypes = [C.POINTER(OraClock), C.c_double]
        L.ora_clock_init.restype = None
        L.ora_clock_run.argtypes = [C.POINTER(OraClock), C.POINTER(C.c_int)]
        L.ora_clock_run.restype = C.c_double
        L.ora_clock_advance.argtypes = [C.POINTER(OraClock), C.c_uint64]
        L.ora_clock_advance.restype = None
        L.ora_step_batch.argtypes = [C.c_int64, C.c_int, C.POINTER(OraParams), C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ora_step_batch.restype = None
        L.ora_set_batch_threads.argtypes = [C.c_int]
        L.ora_set_batch_threads.restype = None
        u32p, u64 = C.POINTER(C.c_uint32), C.c_uint64
        L.ora_philox4x32_10.argtypes = [u32p, u32p, u32p]
        L.ora_philox4x32_10.restype = None
        L.ora_counter_block.argtypes = [u64, u64, C.c_uint, C.c_uint, u64, u32p]
        L.ora_counter_block.restype = None
        L.ora_counter_normals4.argtypes = [u32p, dp]
        L.ora_counter_normals4.restype = None
        L.ora_imu_normals.argtypes = [u64, u64, u64, dp]
        L.ora_imu_normals.restype = None
        L.ora_gust_force.argtypes = [u64, u64, u64, u64, C.c_double, dp]
        L.ora_gust_force.restype = None
        L.ora_step_batch_counter.argtypes = [C.c_int64, C.c_int, C.POINTER(OraParams)] + [C.c_void_p] * 10 + [u64, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_int, u64, u64, u64, u64, u64, u64, u64, C.c_double]
        L.ora_step_batch_counter.restype = None
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def params_from_type(t):
    p = OraParams()
    if lib().ora_params_from_type(C.byref(p), int(t)) != 0:
        raise ValueError("invalid quadcopter type %r" % (t,))
    return p


def params_init(mass, inertia, arm_length, com_error, motor_min_speed, motor_max_speed,
                k_thrust, k_torque, motor_time_const, motor_inertia, lin_drag,
                imu_ypr=(0.0, 0.0, 0.0)):
    p = OraParams()
    I = np.ascontiguousarray(np.asarray(inertia, dtype=np.float64).reshape(9))
    ce = np.ascontiguousarray(np.asarray(com_error, dtype=np.float64))
    dr = np.ascontiguousarray(np.asarray(lin_drag, dtype=np.float64))
    lib().ora_params_init(C.byref(p), mass, _dp(I), arm_length, _dp(ce), motor_min_speed,
                          motor_max_speed, k_thrust, k_torque, motor_time_const,
                          motor_inertia, _dp(dr), *[float(x) for x in imu_ypr])
    return p


def params_table(plist):
    arr = (OraParams * len(plist))()
    for i, p in enumerate(plist):
        C.memmove(C.byref(arr[i]), C.byref(p), C.sizeof(OraParams))
    return arr


def params_to_dict(p):
    """Plain-python view of a record (what the engine's afe_vehicle_params wants)."""
    return dict(
        mass=p.mass, inertia=list(p.inertia), inertia_inv=list(p.inertia_inv),
        motor_pos=[list(r) for r in p.motor_pos],
        motor_rot_axis=[list(r) for r in p.motor_rot_axis],
        motor_thrust_axis=[list(r) for r in p.motor_thrust_axis],
        motor_min_speed=p.motor_min_speed, motor_max_speed=p.motor_max_speed,
        k_thrust=p.k_thrust, k_torque=p.k_torque, motor_time_const=p.motor_time_const,
        motor_inertia=p.motor_inertia, lin_drag=list(p.lin_drag),
        R_imu_inv=list(p.R_imu_inv), sigma_acc=p.sigma_acc, sigma_gyro=p.sigma_gyro)


class Batch:
    """Planar SoA ensemble stepped by the oracle (double precision)."""

    def __init__(self, n, table, types=None):
        self.n = int(n)
        self.table = table if not isinstance(table, (list, tuple)) else params_table(table)
        self.types = (np.zeros(n, np.uint8) if types is None
                      else np.ascontiguousarray(types, dtype=np.uint8))
        self.pos = np.zeros((3, n))
        self.vel = np.zeros((3, n))
        self.att = np.zeros((4, n))
        self.att[0] = 1.0
        self.ang_vel = np.zeros((3, n))
        self.motor_speed = np.zeros((4, n))
        self.rng = np.ones(n, np.uint32)
        self.motor_cmd = np.zeros((4, n), np.float32)
        self.ext_force = np.zeros((3, n))
        self.ext_torque = np.zeros((3, n))
        self.gyro = np.zeros((3, n), np.float32)
        self.acc = np.zeros((3, n), np.float32)

    def step(self, dt, n_steps=1, ticks=None):
        ticks = (np.zeros(n_steps, np.uint8) if ticks is None
                 else np.ascontiguousarray(ticks, dtype=np.uint8))
        assert ticks.shape == (n_steps,)
        for a in (self.pos, self.vel, self.att, self.ang_vel, self.motor_speed,
                  self.motor_cmd, self.ext_force, self.ext_torque, self.gyro, self.acc):
            assert a.flags.c_contiguous
        lib().ora_step_batch(self.n, n_steps, self.table, self.types.ctypes.data,
                             self.pos.ctypes.data, self.vel.ctypes.data, self.att.ctypes.data,
                             self.ang_vel.ctypes.data, self.motor_speed.ctypes.data,
                             self.rng.ctypes.data, self.motor_cmd.ctypes.data,
                             self.ext_force.ctypes.data, self.ext_torque.ctypes.data,
                             float(dt), ticks.ctypes.data, self.gyro.ctypes.data,
                             self.acc.ctypes.data)


def philox4x32_10(ctr, key):
    c, k, o = (C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), (C.c_uint32 * 4)()
    lib().ora_philox4x32_10(c, k, o)
    return [int(x) for x in o]


def imu_normals(seed, index, tick):
    """the six N(0,1) of the counter policy: gyro x y z, accelerometer x y z (agrifly_oracle_counter.h)"""
    z = np.zeros(6)
    lib().ora_imu_normals(int(seed), int(index), int(tick), _dp(z))
    return z


def gust_force(seed, index, n_global, epoch, sigma_max):
    f = np.zeros(3)
    lib().ora_gust_force(int(seed), int(index), int(n_global), int(epoch), float(sigma_max), _dp(f))
    return f


def gust_forces(seed, first, count, n_global, epoch, sigma_max):
    """planar [3, count] forces of vehicles first .. first + count - 1 during `epoch`"""
    return np.stack([gust_force(seed, first + i, n_global, epoch, sigma_max) for i in range(count)], axis=1)


def step_counter(b, dt_us, n_steps, ticks, counter_noise=True, seed=0, first_global=0, tick_base=0, gust_period_us=0, t0_us=0,
                 n_global=None, sigma_max=0.0, gust_seed=None):
    """Batch.step with the counter policy / the gust process (ora_step_batch_counter); b.ext_force receives the last gust"""
    ticks = np.ascontiguousarray(ticks, dtype=np.uint8)
    assert ticks.shape == (n_steps,)
    lib().ora_step_batch_counter(b.n, n_steps, b.table, b.types.ctypes.data, b.pos.ctypes.data, b.vel.ctypes.data, b.att.ctypes.data,
                                 b.ang_vel.ctypes.data, b.motor_speed.ctypes.data, b.rng.ctypes.data, b.motor_cmd.ctypes.data,
                                 b.ext_force.ctypes.data, b.ext_torque.ctypes.data, int(dt_us), ticks.ctypes.data, b.gyro.ctypes.data,
                                 b.acc.ctypes.data, int(bool(counter_noise)), int(seed), int(first_global), int(tick_base),
                                 int(seed if gust_seed is None else gust_seed), int(gust_period_us), int(t0_us), int(b.n if n_global is None else n_global), float(sigma_max))


def clock_ticks(loop_dt, period, n_runs):
    """dt and logic-tick pattern of n_runs Run() calls in the reference loop."""
    c = OraClock()
    lib().ora_clock_init(C.byref(c), period)
    adv = int(np.uint64(loop_dt * 1e6))  # main.cpp:392 uint64_t(dt*1e6)
    dts, ticks = [], []
    t = C.c_int(0)
    for _ in range(n_runs):
        dts.append(lib().ora_clock_run(C.byref(c), C.byref(t)))
        ticks.append(t.value)
        lib().ora_clock_advance(C.byref(c), adv)
    return dts, ticks


# ---------------------------------------------------------------------------
# onboard rates-control path (oracle/agrifly_oracle_logic.c), SURVEY 8f row f1

class OraLpf2(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("a1", "a2", "b0", "b1", "b2", "xm0", "xm1", "ym0", "ym1")]


class OraLogicParams(C.Structure):
    _fields_ = [("mass", C.c_float), ("inertia", C.c_float * 9), ("tc_xy", C.c_float), ("tc_z", C.c_float),
                ("d", C.c_float), ("kt", C.c_float), ("kf", C.c_float), ("max_thrust", C.c_float),
                ("min_thrust", C.c_float), ("max_cmd_total_thrust", C.c_float), ("R", C.c_float * 9),
                ("onboard_period", C.c_float), ("gyro_cutoff", C.c_float)]


class OraLogicState(C.Structure):
    _fields_ = [("gyro_lpf", OraLpf2 * 3), ("ang_vel_est", C.c_float * 3), ("imu_initialized", C.c_int),
                ("have_rates_cmd", C.c_int), ("thrust_norm", C.c_float), ("des_ang_vel", C.c_float * 3),
                ("motor_speed_cmd", C.c_float * 4), ("motor_force_cmd", C.c_float * 4)]


_logic_bound = False


def logic_lib():
    global _logic_bound
    L = lib()
    if not _logic_bound:
        L.ora_lpf2_init.argtypes = [C.POINTER(OraLpf2), C.c_float, C.c_float, C.c_float]
        L.ora_lpf2_init.restype = None
        L.ora_lpf2_apply.argtypes = [C.POINTER(OraLpf2), C.c_float]
        L.ora_lpf2_apply.restype = C.c_float
        L.ora_logic_params_from_type.argtypes = [C.POINTER(OraLogicParams), C.c_int, C.c_float]
        L.ora_logic_params_from_type.restype = C.c_int
        L.ora_logic_init.argtypes = [C.POINTER(OraLogicParams), C.POINTER(OraLogicState)]
        L.ora_logic_init.restype = None
        L.ora_logic_set_rates_cmd.argtypes = [C.POINTER(OraLogicState), C.c_float, C.POINTER(C.c_float)]
        L.ora_logic_set_rates_cmd.restype = None
        L.ora_logic_tick.argtypes = [C.POINTER(OraLogicParams), C.POINTER(OraLogicState), C.POINTER(C.c_float)]
        L.ora_logic_tick.restype = None
        _logic_bound = True
    return L


def logic_params_from_type(t, onboard_period):
    p = OraLogicParams()
    if logic_lib().ora_logic_params_from_type(C.byref(p), int(t), float(np.float32(onboard_period))) != 0:
        raise ValueError("invalid quadcopter type %r" % (t,))
    return p


class ClosedLoopBatch:
    """n vehicles stepped by the oracle with the restated onboard rates logic in
    the loop: the sequence of Quadcopter_T::Run (physics, gate, IMU, logic.Run,
    motor commands taking effect on the next step)."""

    def __init__(self, batch, logic_params_list, onboard_period):
        self.b = batch
        self.L = logic_lib()
        self.lp = logic_params_list
        self.states = []
        for i in range(batch.n):
            s = OraLogicState()
            self.L.ora_logic_init(C.byref(self.lp[int(batch.types[i])]), C.byref(s))
            self.states.append(s)

    def set_rates_cmd(self, thrust_norm, des_ang_vel):
        """thrust_norm[n], des_ang_vel[3, n]"""
        for i, s in enumerate(self.states):
            w = (C.c_float * 3)(*[float(des_ang_vel[k][i]) for k in range(3)])
            self.L.ora_logic_set_rates_cmd(C.byref(s), float(thrust_norm[i]), w)

    def step(self, dt, ticks):
        for tick in ticks:
            self.b.step(dt, 1, ticks=[int(tick)])
            if tick:
                for i, s in enumerate(self.states):
                    g = (C.c_float * 3)(*[float(self.b.gyro[k, i]) for k in range(3)])
                    self.L.ora_logic_tick(C.byref(self.lp[int(self.b.types[i])]), C.byref(s), g)
                    for m in range(4):
                        self.b.motor_cmd[m, i] = s.motor_speed_cmd[m]


# ---------------------------------------------------------------------------
# RAPPIDS depth-image planner (oracle/agrifly_oracle_planner.c), SURVEY 8f row f3

class OraAxis(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("p0", "v0", "a0", "pf", "vf", "af", "a", "b", "g", "cost")] + \
               [("peak_t", C.c_double * 2), ("peak_init", C.c_int)]


class OraPlannerConfig(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("depth_scale", C.c_double), ("focal_length", C.c_double),
                ("cx", C.c_double), ("cy", C.c_double), ("true_vehicle_radius", C.c_double),
                ("planning_vehicle_radius", C.c_double), ("min_checking_dist", C.c_double),
                ("min_thrust", C.c_double), ("max_thrust", C.c_double), ("max_ang_vel", C.c_double),
                ("max_velocity", C.c_double), ("min_section_time", C.c_double), ("max_pyramids", C.c_int),
                ("pixel_buffer", C.c_int), ("cost_type", C.c_int), ("cost_vec", C.c_double * 3)]


class OraPlanResult(C.Structure):
    _fields_ = [("found", C.c_int), ("best_index", C.c_int), ("best_cost", C.c_double),
                ("coeffs", (C.c_double * 3) * 6), ("tf", C.c_double), ("n_generated", C.c_int),
                ("n_cost_checks", C.c_int), ("n_collision_checks", C.c_int), ("n_velocity_checks", C.c_int),
                ("n_collision_free", C.c_int), ("n_pyramids", C.c_int)]


_planner_bound = False


def planner_lib():
    global _planner_bound
    L = lib()
    if not _planner_bound:
        dp = C.POINTER(C.c_double)
        L.ora_solve_cubic.argtypes = [C.c_double, C.c_double, C.c_double, dp]
        L.ora_solve_cubic.restype = C.c_uint
        L.ora_solve_quartic.argtypes = [C.c_double] * 4 + [dp]
        L.ora_solve_quartic.restype = C.c_uint
        L.ora_axis_generate.argtypes = [C.POINTER(OraAxis), C.c_double]
        L.ora_axis_generate.restype = None
        L.ora_axis_minmax_acc.argtypes = [C.POINTER(OraAxis), dp, dp, C.c_double, C.c_double]
        L.ora_axis_minmax_acc.restype = None
        L.ora_axis_max_jerk_sq.argtypes = [C.POINTER(OraAxis), C.c_double, C.c_double]
        L.ora_axis_max_jerk_sq.restype = C.c_double
        for n in ("ora_axis_pos", "ora_axis_vel", "ora_axis_acc"):
            getattr(L, n).argtypes = [C.POINTER(OraAxis), C.c_double]
            getattr(L, n).restype = C.c_double
        L.ora_planner_default_config.argtypes = [C.POINTER(OraPlannerConfig), C.c_int, C.c_int] + [C.c_double] * 5
        L.ora_planner_default_config.restype = None
        L.ora_planner_run.argtypes = [C.POINTER(OraPlannerConfig), C.c_void_p, dp, dp, dp, C.c_void_p, C.c_int,
                                      C.POINTER(OraPlanResult), C.c_void_p]
        L.ora_planner_run.restype = None
        L.ora_planner_nudge.argtypes = [C.c_long, C.c_int]
        L.ora_planner_nudge.restype = None
        L.ora_planner_nudge_calls.argtypes = []
        L.ora_planner_nudge_calls.restype = C.c_long
        L.ora_planner_samples.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.ora_planner_samples.restype = None
        L.ora_planner_sampled_collision.argtypes = [C.POINTER(OraPlannerConfig), C.c_void_p, C.c_void_p, C.c_double, C.c_int]
        L.ora_planner_sampled_collision.restype = C.c_int
        _planner_bound = True
    return L


def planner_config(width, height, depth_scale, focal_length, true_radius, planning_radius, min_checking_dist):
    c = OraPlannerConfig()
    planner_lib().ora_planner_default_config(C.byref(c), width, height, depth_scale, focal_length, true_radius,
                                             planning_radius, min_checking_dist)
    return c


def planner_samples(seed, width, height, n):
    s = np.empty((n, 4))
    planner_lib().ora_planner_samples(int(seed), width, height, n, s.ctypes.data)
    return s


def planner_nudge(call_index, ulps):
    """sensitivity hook: move the call_index-th acos / cos / pow result of the next planner_run on THIS thread
    by `ulps` (call_index < 0: off).  planner_nudge_calls() = how many such calls the last run made."""
    planner_lib().ora_planner_nudge(int(call_index), int(ulps))


def planner_nudge_calls():
    return int(planner_lib().ora_planner_nudge_calls())


def planner_run(cfg, depth, vel0, acc0, grav, samples):
    d = np.ascontiguousarray(depth, dtype=np.uint16)
    assert d.shape == (cfg.height, cfg.width)
    s = np.ascontiguousarray(samples, dtype=np.float64)
    flags = np.zeros(len(s), np.uint8)
    res = OraPlanResult()
    v, a, g = [np.ascontiguousarray(x, dtype=np.float64) for x in (vel0, acc0, grav)]
    planner_lib().ora_planner_run(C.byref(cfg), d.ctypes.data, _dp(v), _dp(a), _dp(g), s.ctypes.data, len(s),
                                  C.byref(res), flags.ctypes.data)
    return res, flags


# ---------------------------------------------------------------------------
# depth-camera checker (oracle/agrifly_oracle_render.c), SURVEY 8f row f4
# ---------------------------------------------------------------------------
class OraCamera(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("focal_length", C.c_double), ("cx", C.c_double),
                ("cy", C.c_double), ("depth_scale", C.c_double), ("max_count", C.c_int)]


_render_bound = False


def render_lib():
    global _render_bound
    L = lib()
    if not _render_bound:
        dp = C.POINTER(C.c_double)
        L.ora_quat_mul.argtypes = [dp, dp, dp]
        L.ora_quat_mul.restype = None
        L.ora_quat_to_matrix.argtypes = [dp, dp]
        L.ora_quat_to_matrix.restype = None
        L.ora_render_depth.argtypes = [C.POINTER(OraCamera), C.c_void_p, C.c_int64, dp, dp, dp, C.c_void_p]
        L.ora_render_depth.restype = None
        L.ora_render_pixel_depth.argtypes = [C.POINTER(OraCamera), C.c_void_p, C.c_int64, dp, dp, dp, C.c_int, C.c_int]
        L.ora_render_pixel_depth.restype = C.c_double
        _render_bound = True
    return L


def render_camera(width, height, focal_length=None, depth_scale=10.0 / 256.0, max_count=255):
    """main.cpp's camera: focal = width/2, principal point = centre (main.cpp:360,485-486)."""
    return OraCamera(width, height, width / 2.0 if focal_length is None else focal_length, width / 2.0, height / 2.0,
                     depth_scale, max_count)


def render_depth(cam, triangles, pos, att, mount=(1.0, 0.0, 0.0, 0.0)):
    t = np.ascontiguousarray(triangles, dtype=np.float32).reshape(-1, 9)
    out = np.empty((cam.height, cam.width), np.uint16)
    render_lib().ora_render_depth(C.byref(cam), t.ctypes.data, t.shape[0], _dp(np.ascontiguousarray(pos, dtype=float)),
                                  _dp(np.ascontiguousarray(att, dtype=float)), _dp(np.ascontiguousarray(mount, dtype=float)), out.ctypes.data)
    return out


def render_pixel_depth(cam, triangles, pos, att, mount, px, py):
    t = np.ascontiguousarray(triangles, dtype=np.float32).reshape(-1, 9)
    return render_lib().ora_render_pixel_depth(C.byref(cam), t.ctypes.data, t.shape[0], _dp(np.ascontiguousarray(pos, dtype=float)),
                                               _dp(np.ascontiguousarray(att, dtype=float)), _dp(np.ascontiguousarray(mount, dtype=float)), px, py)


# ---- shared-world consumers (agrifly_oracle_world.h) ---------------------------

class OraUwb(C.Structure):
    _fields_ = [("mt", C.c_uint32 * 624), ("idx", C.c_int), ("saved_available", C.c_int), ("saved", C.c_double),
                ("noise_std", C.c_double), ("outlier_prob", C.c_double), ("outlier_std", C.c_double)]


_world_bound = False


def world_lib():
    global _world_bound
    L = lib()
    if not _world_bound:
        dp = C.POINTER(C.c_double)
        L.ora_nearest_neighbour.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
        L.ora_nearest_neighbour.restype = None
        L.ora_uwb_init.argtypes = [C.POINTER(OraUwb), C.c_double, C.c_double, C.c_double]
        L.ora_uwb_init.restype = None
        L.ora_uwb_range.argtypes = [C.POINTER(OraUwb), dp, dp, C.POINTER(C.c_int)]
        L.ora_uwb_range.restype = C.c_float
        L.ora_uwb_draw.argtypes = [C.POINTER(OraUwb), C.POINTER(C.c_int)]
        L.ora_uwb_draw.restype = C.c_double
        L.ora_uwb_mt_next.argtypes = [C.POINTER(OraUwb)]
        L.ora_uwb_mt_next.restype = C.c_uint32
        L.ora_uwb_canonical.argtypes = [C.POINTER(OraUwb)]
        L.ora_uwb_canonical.restype = C.c_double
        L.ora_uwb_normal.argtypes = [C.POINTER(OraUwb)]
        L.ora_uwb_normal.restype = C.c_double
        _world_bound = True
    return L


def nearest_neighbour(all_xyz, first=0, count=None):
    """O(n^2) definition: (dist2 float32[count], index int32[count])"""
    a = np.ascontiguousarray(all_xyz, dtype=np.float32)
    n_all = a.shape[1]
    count = n_all - first if count is None else count
    d = np.empty(count, np.float32)
    i = np.empty(count, np.int32)
    world_lib().ora_nearest_neighbour(a.ctypes.data, n_all, first, count, d.ctypes.data, i.ctypes.data)
    return d, i


class UwbNetwork:
    """Simulation::UWBNetwork's completion branch, UWBNetwork.cpp:66-71."""

    def __init__(self, noise_std=0.0, outlier_prob=0.0, outlier_std=0.0):
        self.u = OraUwb()
        world_lib().ora_uwb_init(C.byref(self.u), noise_std, outlier_prob, outlier_std)

    def range(self, p_req, p_res):
        out = C.c_int(0)
        r = world_lib().ora_uwb_range(C.byref(self.u), _dp(np.ascontiguousarray(p_req, dtype=float)),
                                      _dp(np.ascontiguousarray(p_res, dtype=float)), C.byref(out))
        return np.float32(r), out.value

    def draw(self):
        out = C.c_int(0)
        return world_lib().ora_uwb_draw(C.byref(self.u), C.byref(out)), out.value

    def raw(self):
        return world_lib().ora_uwb_mt_next(C.byref(self.u))

    def canonical(self):
        return world_lib().ora_uwb_canonical(C.byref(self.u))

    def normal(self):
        return world_lib().ora_uwb_normal(C.byref(self.u))
