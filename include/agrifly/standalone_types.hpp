// standalone_types.hpp -- minimal value types for using the facade OUTSIDE the
// agri-fly tree (tests, new hosts).  Inside the agri-fly tree do not include
// this file: define AGRIFLY_USE_REFERENCE_TYPES and include the tree's own
// Common/Math/Vec3.hpp, Common/Math/Rotation.hpp, Common/Time/*.hpp first, and
// the facade binds to those instead.
//
// Only the members the facade and its callers touch are provided, with the
// same names and meaning as the reference's classes (Vec3: x,y,z, operator[];
// Rotation: scalar-first operator[], Identity, Inverse, FromEulerYPR,
// operator* on vectors and rotations, From/ToRotationVector, ToEulerYPR; BaseTimer::GetMicroSeconds;
// ManualTimer; Timer incl. AdjustTimeBySeconds).
#pragma once
#include <stdint.h>

#include <cmath>
#include <limits>

template <typename Real>
struct Vec3 {
  Real x, y, z;
  Vec3() : x(std::numeric_limits<Real>::quiet_NaN()), y(x), z(x) {}  // NaN like the reference
  Vec3(Real a, Real b, Real c) : x(a), y(b), z(c) {}
  template <typename Other>
  explicit Vec3(const Vec3<Other> &o) : x(Real(o.x)), y(Real(o.y)), z(Real(o.z)) {}
  Real operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
  Real &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
  Vec3 operator+(const Vec3 &r) const { return Vec3(x + r.x, y + r.y, z + r.z); }
  Vec3 operator-(const Vec3 &r) const { return Vec3(x - r.x, y - r.y, z - r.z); }
  Vec3 operator*(Real s) const { return Vec3(s * x, s * y, s * z); }
  Vec3 operator/(Real s) const { return Vec3(x / s, y / s, z / s); }
  Vec3 operator-() const { return Vec3(-x, -y, -z); }
  Vec3 &operator*=(Real s) { x *= s; y *= s; z *= s; return *this; }
  Real Dot(const Vec3 &r) const { return x * r.x + y * r.y + z * r.z; }
  Vec3 Cross(const Vec3 &r) const { return Vec3(y * r.z - z * r.y, z * r.x - x * r.z, x * r.y - y * r.x); }   // Vec3.hpp:106-109
  Real GetNorm2() const { return std::sqrt(Dot(*this)); }
};
template <typename Real>
inline Vec3<Real> operator*(Real s, const Vec3<Real> &v) { return v * s; }
typedef Vec3<float> Vec3f;
typedef Vec3<double> Vec3d;

template <typename Real>
class Rotation {
 public:
  Rotation() {}
  Rotation(Real a, Real b, Real c, Real d) { v_[0] = a; v_[1] = b; v_[2] = c; v_[3] = d; }
  static Rotation Identity() { return Rotation(1, 0, 0, 0); }
  Rotation Inverse() const { return Rotation(v_[0], -v_[1], -v_[2], -v_[3]); }
  template <typename Other>
  explicit Rotation(const Rotation<Other> &o) { for (unsigned i = 0; i < 4; i++) v_[i] = Real(o[i]); }
  // Rotation.hpp:84-97
  static Rotation FromRotationVector(const Vec3<Real> r) {
    const Real theta = r.GetNorm2();
    if (theta < Real(4.84813681e-6)) return Identity();
    const Vec3<Real> u = r / theta;
    const Real half = Real(0.5);
    return Rotation(std::cos(theta * half), std::sin(theta * half) * u.x, std::sin(theta * half) * u.y,
                    std::sin(theta * half) * u.z);
  }
  // Rotation.hpp:144-161
  Vec3<Real> ToRotationVector() const {
    const Vec3<Real> n = v_[0] > 0 ? Vec3<Real>(v_[1], v_[2], v_[3]) : Vec3<Real>(-v_[1], -v_[2], -v_[3]);
    const Real norm = n.GetNorm2();
    const Real angle = std::asin(norm) * 2;
    if (angle < Real(4.84813681e-6)) return Vec3<Real>(0, 0, 0);
    return n * (angle / norm);
  }
  // Rotation.hpp:163-176
  Vec3<Real> ToEulerYPR() const {
    const Real y = std::atan2(Real(2.0) * v_[1] * v_[2] + Real(2.0) * v_[0] * v_[3],
                              v_[1] * v_[1] + v_[0] * v_[0] - v_[3] * v_[3] - v_[2] * v_[2]);
    const Real p = -std::asin(Real(2.0) * v_[1] * v_[3] - Real(2.0) * v_[0] * v_[2]);
    const Real r = std::atan2(Real(2.0) * v_[2] * v_[3] + Real(2.0) * v_[0] * v_[1],
                              v_[3] * v_[3] - v_[2] * v_[2] - v_[1] * v_[1] + v_[0] * v_[0]);
    return Vec3<Real>(y, p, r);
  }
  // rotation product r2 * r1 (r1 first), Rotation.hpp:124-131
  Rotation operator*(const Rotation &r1) const {
    const Real c0 = r1[0] * v_[0] - r1[1] * v_[1] - r1[2] * v_[2] - r1[3] * v_[3];
    const Real c1 = r1[1] * v_[0] + r1[0] * v_[1] + r1[3] * v_[2] - r1[2] * v_[3];
    const Real c2 = r1[2] * v_[0] - r1[3] * v_[1] + r1[0] * v_[2] + r1[1] * v_[3];
    const Real c3 = r1[3] * v_[0] + r1[2] * v_[1] - r1[1] * v_[2] + r1[0] * v_[3];
    return Rotation(c0, c1, c2, c3);
  }
  static Rotation FromEulerYPR(Real y, Real p, Real r) {  // 3-2-1
    const Real cy = std::cos(y / 2), sy = std::sin(y / 2), cp = std::cos(p / 2), sp = std::sin(p / 2);
    const Real cr = std::cos(r / 2), sr = std::sin(r / 2);
    return Rotation(cy * cp * cr + sy * sp * sr, cy * cp * sr - sy * sp * cr,
                    cy * sp * cr + sy * cp * sr, sy * cp * cr - cy * sp * sr);
  }
  Vec3<Real> operator*(const Vec3<Real> &in) const {
    const Real a = v_[0], b = v_[1], c = v_[2], d = v_[3];
    const Real R[9] = {a * a + b * b - c * c - d * d, 2 * b * c - 2 * a * d, 2 * b * d + 2 * a * c,
                       2 * b * c + 2 * a * d, a * a - b * b + c * c - d * d, 2 * c * d - 2 * a * b,
                       2 * b * d - 2 * a * c, 2 * c * d + 2 * a * b, a * a - b * b - c * c + d * d};
    return Vec3<Real>(R[0] * in.x + R[1] * in.y + R[2] * in.z, R[3] * in.x + R[4] * in.y + R[5] * in.z,
                      R[6] * in.x + R[7] * in.y + R[8] * in.z);
  }
  Real &operator[](unsigned i) { return v_[i]; }
  const Real &operator[](unsigned i) const { return v_[i]; }

 private:
  Real v_[4];
};
typedef Rotation<float> Rotationf;
typedef Rotation<double> Rotationd;

class BaseTimer {
 public:
  virtual ~BaseTimer() {}
  virtual uint64_t GetMicroSeconds(void) const = 0;
};

class ManualTimer : public BaseTimer {
 public:
  ManualTimer() : now_(0) {}
  void ResetMicroseconds(uint64_t t) { now_ = t; }
  void AdvanceMicroSeconds(uint64_t dt) { now_ += dt; }
  virtual uint64_t GetMicroSeconds(void) const { return now_; }

 private:
  uint64_t now_;
};

class Timer {
 public:
  explicit Timer(BaseTimer *const master) : master_(master) { Reset(); }
  uint64_t GetMicroSeconds() const { return master_->GetMicroSeconds() - last_; }
  template <typename Real>
  Real GetSeconds() const { return (Real)(GetMicroSeconds() * Real(1e-6)); }
  void Reset() { last_ = master_->GetMicroSeconds(); }
  template <typename Real>
  void AdjustTimeBySeconds(Real additionalSeconds) {   // Timer.hpp:27-33
    if (additionalSeconds > 0) last_ -= uint64_t(additionalSeconds * Real(1e6));
    else last_ += uint64_t(additionalSeconds * Real(-1e6));
  }
  BaseTimer *GetMasterTimer() const { return master_; }

 private:
  BaseTimer *const master_;
  uint64_t last_;
};
