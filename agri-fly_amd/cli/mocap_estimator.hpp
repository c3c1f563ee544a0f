// mocap_estimator.hpp -- the caller-side state estimator of the Rappids_Simulator loop, restated for the
// headless program (which is built outside the agri-fly tree, where Offboard::MocapStateEstimator itself is
// not available; inside the tree the reference's class drops into the same three calls).
//
// What it restates (Components/Components/Offboard/MocapStateEstimator.{hpp,cpp}, PredictionPipe.hpp):
//   * one constant-velocity filter for the position and one for the attitude, each a two-state
//     (value, rate) Kalman filter with a scalar gain shared by the three axes (MocapStateEstimator.cpp:121-258;
//     initial variances :51-59, noise figures :24-33, 6-sigma measurement gate with a forced reset after ten
//     consecutive rejections :181-205);
//   * between measurements the state is propagated with the COMMANDS in flight: every offboard tick announces
//     (angular velocity, acceleration) and the pair becomes the active command one radio delay later
//     (PredictionPipe.hpp:25-55; the angular velocity relaxes towards the command with a 40 ms time constant,
//     the velocity integrates the commanded acceleration);
//   * Predict(dt) extrapolates the estimate to `dt` past the wall clock without changing it (:62-119) --
//     position and attitude with the filter's CURRENT rates, not the extrapolated ones (kept).
// Arithmetic and its order follow the sources (double; the 2x2 products are written out with the same
// operand order a dense product of [[1, dt], [0, 1]] gives), so that a flight of this program can be laid next to
// a flight of the reference: tests/test_gpu_headless.py holds the headless config 1 against the positions
// SURVEY.md Appendix B records for the unmodified reference.
#pragma once
#include <cmath>
#include <cstdint>
#include <deque>

#include "agrifly/standalone_types.hpp"

namespace agrifly_cli {

struct Estimate {
  Vec3d pos, vel, angVel;
  Rotationd att;
};

class MocapEstimator {
 public:
  MocapEstimator(BaseTimer *master, double commandDelay) : wall_(master), pipeClock_(master), delay_(commandDelay) { Reset(); }

  bool Initialised() const { return started_; }

  // announce what the vehicle will be told to do; active `commandDelay` seconds from now
  void Announce(const Vec3d &angVel, const Vec3d &acc) {
    Plan p;
    p.from = pipeClock_.GetSeconds<double>() + delay_;
    p.acc = acc;
    p.angVel = angVel;
    p.freeFall = false;
    plans_.push_back(p);
  }

  Estimate Predict(double ahead) const {
    const double until = ahead + wall_.GetSeconds<double>();
    Estimate e;
    e.pos = x_; e.vel = v_; e.att = q_; e.angVel = w_;
    double t = validAt_us_ * 1e-6;
    while ((t + kTick) < until) {
      double span = 0;
      const Plan p = Active(t, span);
      double h = until - t;
      if (h > (span + kTick)) h = span;
      const Vec3d x1 = e.pos + v_ * h + p.acc * h * h / 2;          // sic: the filter's own velocity
      const Vec3d v1 = e.vel + p.acc * h;
      const Rotationd q1 = e.att * Rotationd::FromRotationVector(w_ * h);   // sic: the filter's own rate
      double keep = std::exp(-h / rateTimeConstant_);
      if (p.freeFall) keep = 1;
      const Vec3d w1 = keep * e.angVel + (1 - keep) * p.angVel;
      e.pos = x1; e.vel = v1; e.att = q1; e.angVel = w1;
      t += h;
    }
    return e;
  }

  void Measure(const Vec3d &pos, const Rotationd &att) {
    if (!started_) {
      started_ = true;
      x_ = pos; v_ = Vec3d(0, 0, 0); q_ = att; w_ = Vec3d(0, 0, 0);
      FreshVariances();
      return;
    }
    const double now = wall_.GetSeconds<double>();
    if (now > validAt_us_ * 1e-6) {
      for (;;) {
        const double at = validAt_us_ * 1e-6;
        if ((at + kTick) >= now) break;
        double span = 0;
        const Plan p = Active(at, span);
        double h = now - at;
        if (h > (span + kTick)) h = span;
        const Vec3d x0(x_), v0(v_), w0(w_);
        const Rotationd q0(q_);
        x_ = x0 + v0 * h;
        v_ = v0 + p.acc * h;
        q_ = q0 * Rotationd::FromRotationVector(w0 * h);
        double keep = std::exp(-h / rateTimeConstant_);
        if (p.freeFall) keep = 1;
        w_ = keep * w0 + (1 - keep) * p.angVel;
        validAt_us_ += uint64_t(0.5 + h * 1e6);
        Grow(P_, h, sigmaAcc_);
        Grow(A_, h, sigmaAngAcc_);
      }
    }
    double sP = P_.vv + sigmaPos_ * sigmaPos_, sA = A_.vv + sigmaAtt_ * sigmaAtt_;
    const double dP = (pos - x_).GetNorm2() / std::sqrt(3 * sP);
    const double dA = Angle(att.Inverse() * q_) / std::sqrt(sA);
    const bool outlier = (dP > gate_) || (dA > gate_);
    if (outlier && rejectedInARow_ < 10) {
      rejected_++;
      rejectedInARow_++;
    } else {
      if (rejectedInARow_ >= 10) {
        Reset();
        sP = P_.vv + sigmaPos_ * sigmaPos_;
        sA = A_.vv + sigmaAtt_ * sigmaAtt_;
      }
      rejectedInARow_ = 0;
      const double gP0 = P_.vv * (1 / sP), gP1 = P_.rv * (1 / sP);
      const double gA0 = A_.vv * (1 / sA), gA1 = A_.rv * (1 / sA);
      const Vec3d ex = pos - x_;
      x_ = x_ + gP0 * ex;
      v_ = v_ + gP1 * ex;
      const Vec3d ea = (q_.Inverse() * att).ToRotationVector();
      q_ = q_ * Rotationd::FromRotationVector(gA0 * ea);
      w_ = w_ + gA1 * ea;
      Shrink(P_, gP0, gP1);
      Shrink(A_, gA0, gA1);
    }
    Symmetrise(P_);
    Symmetrise(A_);
    // forget the plans the estimate has moved past (the newest one always stays)
    const double at = validAt_us_ * 1e-6;
    for (size_t n = plans_.size(), i = 0; i < n; i++) {
      if (plans_.size() < 2) break;
      if (plans_[1].from <= at) plans_.pop_front();
    }
  }

  unsigned Rejected() const { return rejected_; }

 private:
  struct Plan { double from; Vec3d acc, angVel; bool freeFall; };
  struct Var { double vv, vr, rv, rr; };   // 2x2 covariance of (value, rate): [[vv, vr], [rv, rr]]

  static double Angle(const Rotationd &r) { return std::acos(std::fabs(r[0])) * 2.0; }   // Rotation.hpp:138-142

  // newest plan already active at t, and for how long it stays the newest (PredictionPipe.hpp:33-55)
  Plan Active(double t, double &span) const {
    double next = 1e10;
    for (size_t k = plans_.size(); k-- > 0;) {
      if ((t + kTick) >= plans_[k].from) { span = next - plans_[k].from; return plans_[k]; }
      next = plans_[k].from;
    }
    Plan none;
    none.from = 0; none.acc = Vec3d(0, 0, 0); none.angVel = Vec3d(0, 0, 0); none.freeFall = true;
    span = 1e10;
    return none;
  }

  // F V F' + Q, F = [[1, h], [0, 1]], Q = diag(h^4 s / 4, h^2 s)   (:157-172)
  static void Grow(Var &V, double h, double s) {
    const double a = 1 * V.vv + h * V.rv, b = 1 * V.vr + h * V.rr;
    const double c = 0 * V.vv + 1 * V.rv, d = 0 * V.vr + 1 * V.rr;
    Var n;
    n.vv = (a * 1 + b * h) + h * h * h * h * s / 4;
    n.vr = (a * 0 + b * 1) + 0;
    n.rv = (c * 1 + d * h) + 0;
    n.rr = (c * 0 + d * 1) + h * h * s;
    V = n;
  }
  // (I - g [1 0]) V   (:236-242)
  static void Shrink(Var &V, double g0, double g1) {
    const double a = 1 - g0 * 1, b = 0 - g0 * 0, c = 0 - g1 * 1, d = 1 - g1 * 0;
    Var n;
    n.vv = a * V.vv + b * V.rv; n.vr = a * V.vr + b * V.rr;
    n.rv = c * V.vv + d * V.rv; n.rr = c * V.vr + d * V.rr;
    V = n;
  }
  static void Symmetrise(Var &V) {
    Var n;
    n.vv = (V.vv + V.vv) * 0.5; n.vr = (V.vr + V.rv) * 0.5;
    n.rv = (V.rv + V.vr) * 0.5; n.rr = (V.rr + V.rr) * 0.5;
    V = n;
  }
  void FreshVariances() {
    P_.vv = 25.0; P_.rr = 25.0; P_.vr = P_.rv = 0.0;
    A_.vv = 1.0; A_.rr = 400; A_.vr = A_.rv = 0.0;
  }
  void Reset() {
    started_ = false;
    x_ = Vec3d(0, 0, 0); v_ = Vec3d(0, 0, 0); w_ = Vec3d(0, 0, 0);
    q_ = Rotationd::Identity();
    FreshVariances();
    validAt_us_ = wall_.GetMicroSeconds();
  }

  static constexpr double kTick = 1e-6;
  Timer wall_, pipeClock_;
  double delay_;
  std::deque<Plan> plans_;
  uint64_t validAt_us_ = 0;            // the estimate is the state at this time
  bool started_ = false;
  Vec3d x_, v_, w_;
  Rotationd q_;
  Var P_, A_;
  unsigned rejected_ = 0, rejectedInARow_ = 0;
  double rateTimeConstant_ = 0.04, gate_ = 6.0;
  double sigmaPos_ = 0.02, sigmaAtt_ = 5 * M_PI / 180, sigmaAcc_ = 1.0 * 9.81, sigmaAngAcc_ = 200;
};

}  // namespace agrifly_cli
