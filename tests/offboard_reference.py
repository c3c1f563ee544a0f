"""Test-side restatement of the reference's OFFBOARD chain of config 1, so that the oracle (and the engine) can be
flown in exactly the loop the reference flies -- not on the truth-state stub of tests/offboard_stub.py:

  truth pose at 200 Hz -> Offboard::MocapStateEstimator::UpdateWithMeasurement
                            (Components/Components/Offboard/MocapStateEstimator.cpp:121-258: per-axis two-state
                             Kalman filter, position / attitude decoupled, prediction driven by the commands in
                             flight through a PredictionPipe, PredictionPipe.hpp:25-68)
                       -> GetPrediction(0.03) every loop iteration (:62-119)
  at 100 Hz           -> QuadcopterController::Run on the predicted state (tests/offboard_stub.py, float)
                       -> RadioMessageDecoded::CreateRatesCommand, 16-bit fixed point
                       -> est.SetPredictedValues(cmd ang vel, att * e3 * thrust - g)  (MocapStateEstimator.hpp:70-76)
                       -> CommunicationsDelay 30 ms -> onboard rates logic

Loop order and cadences: Simulator/Rappids_Simulator/main.cpp:330,391-392,451-457,468-476,625-649,673,737-739
(AirSim and the planner branch absent: the pre-takeoff branch, desired position (0, 0, 3.5)).

Why it exists: SURVEY.md Appendix B records where the UNMODIFIED reference, compiled by the surveyor with a minimal
Eigen stand-in (so: informational, not a pin), is after 1 s and after 10 s of this very loop at dt = 1 ms.  Flying the
oracle in the same loop and landing on those numbers is the closest this repository can come to checking its
restatement of Quadcopter_T::Run / Motor::Run / the onboard rates logic against the reference itself
(tests/test_reference_anchors.py).  Scalar (one vehicle), double precision, operation order of the sources.
"""
import math

import numpy as np

from tests.offboard_stub import OffboardHover, radio_quantise

SMALL_TIME = 1e-6
MIN_ANGLE = 4.84813681e-6     # Rotation.hpp: one arc second


# ---- Rotationd / Vec3d, scalar double ------------------------------------------------------------------------------
def q_mul(a, b):
    """Rotation.hpp:124-131, this = a, r1 = b"""
    return (b[0] * a[0] - b[1] * a[1] - b[2] * a[2] - b[3] * a[3],
            b[1] * a[0] + b[0] * a[1] + b[3] * a[2] - b[2] * a[3],
            b[2] * a[0] - b[3] * a[1] + b[0] * a[2] + b[1] * a[3],
            b[3] * a[0] + b[2] * a[1] - b[1] * a[2] + b[0] * a[3])


def q_inv(q):
    return (q[0], -q[1], -q[2], -q[3])


def norm2(v):
    return math.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])     # Vec3.hpp:113-116


def q_from_rotvec(r):
    """Rotation.hpp:84-97"""
    theta = norm2(r)
    if theta < MIN_ANGLE:
        return (1.0, 0.0, 0.0, 0.0)
    u = (r[0] / theta, r[1] / theta, r[2] / theta)
    s = math.sin(theta * 0.5)
    return (math.cos(theta * 0.5), s * u[0], s * u[1], s * u[2])


def q_to_rotvec(q):
    """Rotation.hpp:144-161"""
    n = (q[1], q[2], q[3]) if q[0] > 0 else (-q[1], -q[2], -q[3])
    nn = norm2(n)
    angle = math.asin(nn) * 2 if nn <= 1.0 else float("nan")
    if angle < MIN_ANGLE:
        return (0.0, 0.0, 0.0)
    k = angle / nn
    return (n[0] * k, n[1] * k, n[2] * k)


def q_angle(q):
    """Rotation.hpp:138-142; acos of a |q0| a rounding above 1 (fp32 attitudes) is a NaN there, not an exception"""
    a = abs(q[0])
    return math.acos(a) * 2.0 if a <= 1.0 else float("nan")


def q_rotate(q, v):
    """Rotation.hpp:196-245: the 3x3 built from the quaternion, then mat-vec"""
    r0, r1, r2, r3 = q[0] * q[0], q[1] * q[1], q[2] * q[2], q[3] * q[3]
    R = (r0 + r1 - r2 - r3, 2 * q[1] * q[2] - 2 * q[0] * q[3], 2 * q[1] * q[3] + 2 * q[0] * q[2],
         2 * q[1] * q[2] + 2 * q[0] * q[3], r0 - r1 + r2 - r3, 2 * q[2] * q[3] - 2 * q[0] * q[1],
         2 * q[1] * q[3] - 2 * q[0] * q[2], 2 * q[2] * q[3] + 2 * q[0] * q[1], r0 - r1 - r2 + r3)
    return (R[0] * v[0] + R[1] * v[1] + R[2] * v[2], R[3] * v[0] + R[4] * v[1] + R[5] * v[2],
            R[6] * v[0] + R[7] * v[1] + R[8] * v[2])


def v_add(a, b):
    return (a[0] + b[0], a[1] + b[1], a[2] + b[2])


def v_sub(a, b):
    return (a[0] - b[0], a[1] - b[1], a[2] - b[2])


def v_scale(a, k):
    return (a[0] * k, a[1] * k, a[2] * k)


# ---- clocks (Timer.hpp / ManualTimer.hpp semantics on an integer microsecond master clock) ---------------------------
class Clock:
    def __init__(self):
        self.us = 0


class Timer:
    def __init__(self, clock):
        self.clock, self.reset_us = clock, clock.us

    def seconds(self):
        return (self.clock.us - self.reset_us) * 1e-6

    def reset(self):
        self.reset_us = self.clock.us

    def adjust(self, seconds):
        """Timer::AdjustTimeBySeconds: a negative adjustment moves the reset point forward"""
        if seconds < 0:
            self.reset_us += int(-seconds * 1e6)
        else:
            self.reset_us -= int(seconds * 1e6)


class PredictionPipe:
    """PredictionPipe.hpp:25-68"""

    def __init__(self, clock, delay):
        self.timer, self.delay, self.msgs = Timer(clock), delay, []

    def add(self, msg):
        self.msgs.append((self.timer.seconds() + self.delay, msg))

    def active(self, t):
        """(message, time remaining until the next one becomes active) or None"""
        if not self.msgs:
            return None
        t_last = 1e10
        for k in range(len(self.msgs) - 1, -1, -1):
            t_act, m = self.msgs[k]
            if (t + SMALL_TIME) >= t_act:
                return m, t_last - t_act
            t_last = t_act
        return None

    def clear_expired(self, now):
        for _ in range(len(self.msgs)):
            if len(self.msgs) < 2:
                return
            if self.msgs[1][0] <= now:
                self.msgs.pop(0)


class MocapStateEstimator:
    """MocapStateEstimator.cpp; statistics of :24-33"""

    def __init__(self, clock, delay):
        self.clock = clock
        self.timer = Timer(clock)                 # wall clock since construction
        self.est_us = 0                           # ManualTimer: the time the estimate is valid at
        self.pipe = PredictionPipe(clock, delay)
        self.tc_ang_vel = 0.04
        self.reject_dist = 6.0
        self.meas_pos, self.meas_att = 0.02, 5 * math.pi / 180
        self.proc_pos, self.proc_att = 1.0 * 9.81, 200.0
        self.n_rejected = self.n_rejected_run = 0
        self.reset()

    def reset(self):
        self.initialized = False
        self.pos, self.vel, self.ang_vel = (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0)
        self.att = (1.0, 0.0, 0.0, 0.0)
        self.reset_variance()
        self.est_us = self.clock.us - self.timer.reset_us      # _estimateTimer.ResetMicroseconds(_timer.GetMicroSeconds())

    def reset_variance(self):
        self.Pp = [[25.0, 0.0], [0.0, 25.0]]
        self.Pa = [[1.0, 0.0], [0.0, 400.0]]

    def est_seconds(self):
        return self.est_us * 1e-6

    def set_predicted(self, ang_vel, acc):
        self.pipe.add((acc, ang_vel, False))

    def _command(self, t):
        got = self.pipe.active(t)
        if got is None:
            return ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0), True), 1e10
        return got

    def prediction(self, dt):
        """:62-119 -- note the MEMBER _vel / _angVel in the position / attitude lines"""
        t_end = dt + self.timer.seconds()
        t = self.est_seconds()
        pos, vel, att, ang_vel = self.pos, self.vel, self.att, self.ang_vel
        while (t + SMALL_TIME) < t_end:
            (acc, cmd_w, ballistic), remaining = self._command(t)
            dt_int = t_end - t
            if dt_int > (remaining + SMALL_TIME):
                dt_int = remaining
            new_pos = tuple(pos[k] + self.vel[k] * dt_int + acc[k] * dt_int * dt_int / 2 for k in range(3))
            new_vel = tuple(vel[k] + acc[k] * dt_int for k in range(3))
            new_att = q_mul(att, q_from_rotvec(v_scale(self.ang_vel, dt_int)))
            c = math.exp(-dt_int / self.tc_ang_vel)
            if ballistic:
                c = 1
            new_w = tuple(c * ang_vel[k] + (1 - c) * cmd_w[k] for k in range(3))
            pos, vel, att, ang_vel = new_pos, new_vel, new_att, new_w
            t += dt_int
        return pos, vel, att, ang_vel

    def update(self, meas_pos, meas_att):
        """:121-258"""
        if not self.initialized:
            self.initialized = True
            self.pos, self.vel, self.att, self.ang_vel = tuple(meas_pos), (0.0, 0.0, 0.0), tuple(meas_att), (0.0, 0.0, 0.0)
            self.reset_variance()
            return
        t0, t_end = self.est_seconds(), self.timer.seconds()
        if t_end > t0:
            while True:
                t_now = self.est_seconds()
                if (t_now + SMALL_TIME) >= t_end:
                    break
                (acc, cmd_w, ballistic), remaining = self._command(self.est_seconds())
                dt_int = t_end - self.est_seconds()
                if dt_int > (remaining + SMALL_TIME):
                    dt_int = remaining
                pos, vel, att, w = self.pos, self.vel, self.att, self.ang_vel
                self.pos = tuple(pos[k] + vel[k] * dt_int for k in range(3))
                self.vel = tuple(vel[k] + acc[k] * dt_int for k in range(3))
                self.att = q_mul(att, q_from_rotvec(v_scale(w, dt_int)))
                c = math.exp(-dt_int / self.tc_ang_vel)
                if ballistic:
                    c = 1
                self.ang_vel = tuple(c * w[k] + (1 - c) * cmd_w[k] for k in range(3))
                self.est_us += int(0.5 + dt_int * 1e6)
                for P, proc in ((self.Pp, self.proc_pos), (self.Pa, self.proc_att)):
                    # A P A^T + Q with A = [[1, dt], [0, 1]], Q = diag(dt^4 proc / 4, dt^2 proc)
                    m00, m01 = P[0][0] + dt_int * P[1][0], P[0][1] + dt_int * P[1][1]
                    m10, m11 = P[1][0], P[1][1]
                    n00, n01, n10, n11 = m00 + m01 * dt_int, m01, m10 + m11 * dt_int, m11
                    P[0][0] = n00 + dt_int * dt_int * dt_int * dt_int * proc / 4
                    P[0][1], P[1][0] = n01, n10
                    P[1][1] = n11 + dt_int * dt_int * proc
        inn_p = self.Pp[0][0] + self.meas_pos * self.meas_pos
        inn_a = self.Pa[0][0] + self.meas_att * self.meas_att
        dist_p = norm2(v_sub(meas_pos, self.pos)) / math.sqrt(3 * inn_p)
        dist_a = q_angle(q_mul(q_inv(meas_att), self.att)) / math.sqrt(inn_a)
        reject = dist_p > self.reject_dist or dist_a > self.reject_dist
        if reject and self.n_rejected_run < 10:
            self.n_rejected += 1
            self.n_rejected_run += 1
        else:
            if self.n_rejected_run >= 10:
                self.reset()
                inn_p = self.Pp[0][0] + self.meas_pos * self.meas_pos
                inn_a = self.Pa[0][0] + self.meas_att * self.meas_att
            self.n_rejected_run = 0
            kp0, kp1 = self.Pp[0][0] * (1 / inn_p), self.Pp[1][0] * (1 / inn_p)
            ka0, ka1 = self.Pa[0][0] * (1 / inn_a), self.Pa[1][0] * (1 / inn_a)
            err = v_sub(meas_pos, self.pos)
            self.pos = v_add(self.pos, v_scale(err, kp0))
            self.vel = v_add(self.vel, v_scale(err, kp1))
            err_a = q_to_rotvec(q_mul(q_inv(self.att), meas_att))
            self.att = q_mul(self.att, q_from_rotvec(v_scale(err_a, ka0)))
            self.ang_vel = v_add(self.ang_vel, v_scale(err_a, ka1))
            for P, k0, k1 in ((self.Pp, kp0, kp1), (self.Pa, ka0, ka1)):
                # (I - K H) P with H = [1, 0]
                p00, p01, p10, p11 = P[0][0], P[0][1], P[1][0], P[1][1]
                P[0][0], P[0][1] = (1 - k0) * p00 + 0.0 * p10, (1 - k0) * p01 + 0.0 * p11
                P[1][0], P[1][1] = (0 - k1) * p00 + 1.0 * p10, (0 - k1) * p01 + 1.0 * p11
        for P in (self.Pp, self.Pa):
            a, b = (P[0][1] + P[1][0]) * 0.5, (P[1][0] + P[0][1]) * 0.5
            P[0][0], P[1][1] = (P[0][0] + P[0][0]) * 0.5, (P[1][1] + P[1][1]) * 0.5
            P[0][1], P[1][0] = a, b
        self.pipe.clear_expired(self.est_seconds())


class ReferenceOffboard:
    """the offboard side of one loop iteration, after quad->Run() and the clock advance"""

    def __init__(self, clock, des_pos=(0.0, 0.0, 3.5), delay=0.03):
        self.clock = clock
        self.est = MocapStateEstimator(clock, delay)
        self.ctrl = OffboardHover(1, des_pos=des_pos)
        self.t_mocap, self.t_main = Timer(clock), Timer(clock)
        self.delay_us = int(delay * 1e6)
        self.queue = []

    def iterate(self, true_pos, true_att):
        """returns a (thrust, ang_vel[3]) command when a delayed radio message is delivered in this iteration"""
        if self.t_mocap.seconds() > 1.0 / 200:
            self.t_mocap.adjust(-1.0 / 200)
            self.est.update(tuple(float(x) for x in true_pos), tuple(float(x) for x in true_att))
        pos, vel, att, _ = self.est.prediction(0.03)
        if self.t_main.seconds() > 1.0 / 100:
            self.t_main.adjust(-1.0 / 100)
            thrust, w = self.ctrl.controller(np.array(pos).reshape(3, 1), np.array(vel).reshape(3, 1), np.array(att).reshape(4, 1))
            thr, wd = float(thrust[0]), tuple(float(w[k, 0]) for k in range(3))
            msg = (radio_quantise(np.float32(thr), 35), radio_quantise(w, 35))
            z = q_rotate(att, (0.0, 0.0, 1.0))
            self.est.set_predicted(wd, v_sub(v_scale(z, thr), (0.0, 0.0, 9.81)))
            self.queue.append((self.clock.us + self.delay_us, msg))
        if self.queue and self.clock.us >= self.queue[0][0]:
            return self.queue.pop(0)[1]
        return None
