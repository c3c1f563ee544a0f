"""Test-side caller of the hot path: a numpy restatement of the offboard loop of
Simulator/Rappids_Simulator/main.cpp (pre-takeoff branch, :611-673,737-739)
so that config 0/1/2 of BASELINE.json can be flown CLOSED LOOP on both the
oracle and the HIP engine:

  truth state -> Offboard::QuadcopterController::Run
                   (Components/Components/Offboard/QuadcopterController.cpp:11-74,
                    QuadcopterPositionController.hpp:22-28,
                    QuadcopterAttitudeController.hpp:35-68)       float
              -> RadioMessageDecoded::CreateRatesCommand / decode
                   (Common/Common/DataTypes/RadioTypes.hpp:73-116,158-171,218-226)
                   16-bit fixed point, +-35
              -> CommunicationsDelay 30 ms (CommunicationsDelay.hpp:18-33)
              -> onboard rates logic (oracle restatement / device logic)

One deliberate simplification, stated here because this is NOT the reference's
loop: the reference feeds the controller from MocapStateEstimator::GetPrediction
(200 Hz mocap + 30 ms prediction); this stub feeds it the true state sampled at
the offboard tick.  It is test infrastructure (the caller side is out of scope,
SURVEY.md section 2 row 10); both sides of a parity test use the same stub.
"""
import numpy as np

F = np.float32

# glibc's float transcendentals for small batches: numpy's own float32 routines differ from sinf / cosf / asinf /
# acosf by an ulp now and then, and one ulp in a command flips a 16-bit radio code every few seconds of flight --
# enough to part from a run of the reference by 1e-5 m (tests/test_reference_anchors.py wants them identical, and
# over 10 s of config 1 they are)
import ctypes as _C
import ctypes.util as _Cu

_libm = _C.CDLL(_Cu.find_library("m") or "libm.so.6")
for _n in ("sinf", "cosf", "asinf", "acosf"):
    getattr(_libm, _n).restype = _C.c_float
    getattr(_libm, _n).argtypes = [_C.c_float]


def _tr(name, np_fn, x):
    x = np.asarray(x, F)
    if x.size > 64:
        return np_fn(x).astype(F)
    f = getattr(_libm, name)
    return np.array([f(float(v)) for v in x.reshape(-1)], F).reshape(x.shape)


def _quat_mul(a, b):
    """Rotation.hpp:124-131 (this = a, r1 = b), float32, arrays [4, n]"""
    return np.stack([
        b[0] * a[0] - b[1] * a[1] - b[2] * a[2] - b[3] * a[3],
        b[1] * a[0] + b[0] * a[1] + b[3] * a[2] - b[2] * a[3],
        b[2] * a[0] - b[3] * a[1] + b[0] * a[2] + b[1] * a[3],
        b[3] * a[0] + b[2] * a[1] - b[1] * a[2] + b[0] * a[3]]).astype(F)


def _quat_inv(q):
    return np.stack([q[0], -q[1], -q[2], -q[3]]).astype(F)


def _rot_matrix(q):
    """Rotation.hpp:196-220; returns [9, n]"""
    r0, r1, r2, r3 = q[0] * q[0], q[1] * q[1], q[2] * q[2], q[3] * q[3]
    two = F(2)
    return np.stack([
        r0 + r1 - r2 - r3, two * q[1] * q[2] - two * q[0] * q[3], two * q[1] * q[3] + two * q[0] * q[2],
        two * q[1] * q[2] + two * q[0] * q[3], r0 - r1 + r2 - r3, two * q[2] * q[3] - two * q[0] * q[1],
        two * q[1] * q[3] - two * q[0] * q[2], two * q[2] * q[3] + two * q[0] * q[1], r0 - r1 - r2 + r3]).astype(F)


def _rotate(q, v):
    R = _rot_matrix(q)
    return np.stack([R[0] * v[0] + R[1] * v[1] + R[2] * v[2],
                     R[3] * v[0] + R[4] * v[1] + R[5] * v[2],
                     R[6] * v[0] + R[7] * v[1] + R[8] * v[2]]).astype(F)


def _from_rotvec(r):
    """Rotation.hpp:84-97, float"""
    theta = np.sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]).astype(F)
    small = theta < F(4.84813681e-6)
    th = np.where(small, F(1), theta)
    s = _tr("sinf", np.sin, th * F(0.5))
    q = np.stack([_tr("cosf", np.cos, th * F(0.5)), s * (r[0] / th), s * (r[1] / th), s * (r[2] / th)]).astype(F)
    q[:, small] = np.array([[1], [0], [0], [0]], F)
    return q


def _to_rotvec(q):
    """Rotation.hpp:144-161"""
    sgn = np.where(q[0] > 0, F(1), F(-1))
    n = (q[1:4] * sgn).astype(F)
    norm = np.sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]).astype(F)
    angle = (_tr("asinf", np.arcsin, np.minimum(norm, F(1))) * F(2)).astype(F)
    small = angle < F(4.84813681e-6)
    out = (n * (angle / np.where(norm == 0, F(1), norm))).astype(F)
    out[:, small] = 0
    return out


def radio_quantise(val, limit):
    """encodeToRadioByte + decodeFromRadioBytes, RadioTypes.hpp:73-116 (16 bit)"""
    val = np.asarray(val, F)
    lim = F(limit)
    inside = (val > -lim) & (val < lim)
    scaled = np.where(inside, val * F(32768) / lim + F(0.5), F(0))
    code = np.where(inside, scaled.astype(np.int32) + 32768,
                    np.where(val > -lim, 65535, 0)).astype(np.int32)
    code = code % 65536   # two bytes on the wire
    return (lim * (code - 32768).astype(F) / F(32768)).astype(F)


class OffboardHover:
    """QuadcopterController with the MINIQUAD tuning (QuadcopterConstants.hpp:
    34-39,214-224; main.cpp:225-229), desired position (0, 0, 3.5) (main.cpp:240).
    The attitude time constants are derived in float there: 0.04f * 2 and (0.04f * 5) * 2 = 0.399999976 (not 0.4f)."""

    def __init__(self, n, des_pos=(0.0, 0.0, 3.5), nat_freq=2.0, damping=0.7, tc_xy=F(0.04) * F(2), tc_z=(F(0.04) * F(5)) * F(2),
                 period_offboard=1.0 / 100.0, delay=0.03):
        self.n = n
        self.des_pos = np.tile(np.asarray(des_pos, F).reshape(3, 1), (1, n))
        self.nat_freq, self.damping, self.tc_xy, self.tc_z = F(nat_freq), F(damping), F(tc_xy), F(tc_z)
        self.period = period_offboard
        self.delay_us = int(np.uint64(delay * 1e6))
        self.reset_us = 0
        self.queue = []

    def controller(self, pos, vel, att):
        """QuadcopterController::Run, QuadcopterController.cpp:11-74 (float)"""
        p, v, q = np.asarray(pos, F), np.asarray(vel, F), np.asarray(att, F)
        acc = ((self.des_pos - p) * self.nat_freq * self.nat_freq
               + (F(0) - v) * F(2) * self.nat_freq * self.damping + F(0)).astype(F)
        proper = (acc + np.array([[0], [0], [9.81]], F)).astype(F)
        norm = np.sqrt((proper * proper).sum(0)).astype(F)
        sat = norm > F(20)
        proper[:, sat] = (proper[:, sat] * (F(20) / norm[sat])).astype(F)
        proper[2] = np.maximum(proper[2], F(0.5 * 9.81))
        norm = np.sqrt((proper * proper).sum(0)).astype(F)
        tdir = (proper / norm).astype(F)
        e3 = np.zeros((3, self.n), F)
        e3[2] = 1
        body_z = _rotate(q, e3)
        thrust = np.maximum(norm * (body_z * tdir).sum(0).astype(F), F(-1)).astype(F)
        cosang = tdir[2]
        angle = np.where(cosang >= F(1 - 1e-12), F(0),
                         np.where(cosang <= F(-(1 - 1e-12)), F(np.pi), _tr("acosf", np.arccos, np.clip(cosang, -1, 1)))).astype(F)
        rot_ax = np.stack([-tdir[1], tdir[0], np.zeros(self.n, F)]).astype(F)   # e3 x tdir
        nrm = np.sqrt((rot_ax * rot_ax).sum(0)).astype(F)
        tiny = nrm < F(1e-6)
        cmd_att = _from_rotvec((rot_ax * (angle / np.where(tiny, F(1), nrm))).astype(F))
        cmd_att[:, tiny] = np.array([[1], [0], [0], [0]], F)
        # desired yaw 0: FromRotationVector((0,0,0)) = identity
        # GetDesiredAngularVelocity, QuadcopterAttitudeController.hpp:35-68
        err = _quat_mul(_quat_inv(cmd_att), q)
        des_rot = _to_rotvec(err)
        z_in_err = _rotate(_quat_inv(err), e3)
        red_ax = np.stack([z_in_err[1], -z_in_err[0], np.zeros(self.n, F)]).astype(F)   # (.) x e3
        cos_red = z_in_err[2]
        red_an = np.where(cos_red >= F(1), F(0), np.where(cos_red <= F(-1), F(np.pi),
                                                            _tr("acosf", np.arccos, np.clip(cos_red, -1, 1)))).astype(F)
        nn = np.sqrt((red_ax * red_ax).sum(0)).astype(F)
        red_ax = np.where(nn < F(1e-12), F(0), red_ax / np.where(nn < F(1e-12), F(1), nn)).astype(F)
        k3, k12 = F(1) / self.tc_z, F(1) / self.tc_xy
        ang_vel = (-k3 * des_rot - (k12 - k3) * red_an * red_ax).astype(F)
        return thrust, ang_vel

    def maybe_command(self, now_us, pos, vel, att):
        """the 100 Hz offboard gate (Timer strict >, main.cpp:471-476) + radio + delay queue;
        returns a (thrust, ang_vel) pair when a delayed message is due at now_us"""
        el = (now_us - self.reset_us) * 1e-6
        if el > self.period:
            self.reset_us += int(np.uint64(self.period * 1e6))
            thrust, w = self.controller(pos, vel, att)
            msg = (radio_quantise(thrust, 35), radio_quantise(w, 35))
            self.queue.append((now_us + self.delay_us, msg))
        if self.queue and now_us >= self.queue[0][0]:
            return self.queue.pop(0)[1]
        return None
