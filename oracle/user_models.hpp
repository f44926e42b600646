// ORACLE — test infrastructure only (see bal_model.hpp header).
//
// Factor functions OTHER than the BAL camera, for the parity tests of the engine instantiated on user traits
// (include/graphite/engine_model.hpp, tests/cpp/test_engine_model.hip), differentiated the way the reference differentiates
// a Differentiation::Auto factor: forward-mode dual numbers, one direction per column
// (/root/reference/include/graphite/dual.hpp:8-230 for the arithmetic rules, ops/linearize.hpp:43-79 for the column loop:
// seed parameter `col` with dual = 1, evaluate error<Dual>, the duals of the error are column `col` of the E x d block).
// The two models are the test's own (they are not in the reference); what they pin is that the per-factor
// precision / loss / constraint-data plumbing and the zero-padded (6, 3) -> 2 layout give the reference's ALGORITHM
// (bal_pipeline.hpp) the same numbers on the GPU as here.
//   MODEL_K3      : the BAL camera with a sixth-order radial term, d = 1 + k1 r^2 + k2 r^4 + k3 r^6, k3 = data[0]
//   MODEL_PINHOLE : pose [angle-axis r(3), t(3)] (entries 6..8 of the 9-wide pose block are padding), point X:
//                   P = R(r) X + t, residual = (fx Px / Pz + cx - u, fy Py / Pz + cy - v), data = [fx fy cx cy]
#pragma once
#include <cmath>

namespace gro {

enum { MODEL_BAL = 0, MODEL_K3 = 1, MODEL_PINHOLE = 2 };

template <typename T> struct DualNumber {
  T real, dual;
  DualNumber() : real(0), dual(0) {}
  DualNumber(T r) : real(r), dual(0) {}
  DualNumber(T r, T d) : real(r), dual(d) {}
  DualNumber operator+(const DualNumber &o) const { return {real + o.real, dual + o.dual}; }
  DualNumber operator-(const DualNumber &o) const { return {real - o.real, dual - o.dual}; }
  DualNumber operator-() const { return {-real, -dual}; }
  DualNumber operator*(const DualNumber &o) const { return {real * o.real, real * o.dual + dual * o.real}; }
  DualNumber operator/(const DualNumber &o) const { const T den = o.real * o.real; return {real / o.real, (dual * o.real - real * o.dual) / den}; }
  bool operator>(const DualNumber &o) const { return real > o.real; }
};
template <typename T> inline DualNumber<T> sqrt(const DualNumber<T> &x) { const T s = std::sqrt(x.real); return {s, s == 0 ? T(0) : x.dual / (2 * s)}; }
template <typename T> inline DualNumber<T> sin(const DualNumber<T> &x) { return {std::sin(x.real), x.dual * std::cos(x.real)}; }
template <typename T> inline DualNumber<T> cos(const DualNumber<T> &x) { return {std::cos(x.real), -x.dual * std::sin(x.real)}; }
inline float sqrt(float x) { return std::sqrt(x); }
inline double sqrt(double x) { return std::sqrt(x); }
inline float sin(float x) { return std::sin(x); }
inline double sin(double x) { return std::sin(x); }
inline float cos(float x) { return std::cos(x); }
inline double cos(double x) { return std::cos(x); }

// P = R(r) X + t, Rodrigues matrix, identity at theta == 0
template <typename D, typename T> inline void user_transform(const D *pose, const D *pt, D *P) {
  const D rx = pose[0], ry = pose[1], rz = pose[2];
  const D theta2 = rx * rx + ry * ry + rz * rz;
  if (theta2 > D(T(0))) {
    const D theta = sqrt(theta2);
    const D ax = rx / theta, ay = ry / theta, az = rz / theta;
    const D s = sin(theta), c = cos(theta), k = D(T(1)) - c;
    const D R[9] = {k * ax * ax + c,      k * ax * ay - s * az, k * ax * az + s * ay,
                    k * ax * ay + s * az, k * ay * ay + c,      k * ay * az - s * ax,
                    k * ax * az - s * ay, k * ay * az + s * ax, k * az * az + c};
    for (int i = 0; i < 3; ++i) P[i] = R[3 * i] * pt[0] + R[3 * i + 1] * pt[1] + R[3 * i + 2] * pt[2] + pose[3 + i];
  } else {
    for (int i = 0; i < 3; ++i) P[i] = pt[i] + pose[3 + i];
  }
}
template <typename D, typename T> inline void user_model_error(int kind, const D *cam, const D *pt, const T *obs, const T *data, D *err) {
  D P[3];
  user_transform<D, T>(cam, pt, P);
  if (kind == MODEL_PINHOLE) {
    err[0] = D(data[0]) * (P[0] / P[2]) + D(data[2]) - D(obs[0]);
    err[1] = D(data[1]) * (P[1] / P[2]) + D(data[3]) - D(obs[1]);
    return;
  }
  const D px = -P[0] / P[2], py = -P[1] / P[2];
  const D r2 = px * px + py * py;
  const D d = D(T(1)) + cam[7] * r2 + cam[8] * r2 * r2 + D(data ? data[0] : T(0)) * r2 * r2 * r2;
  err[0] = cam[6] * d * px - D(obs[0]);
  err[1] = cam[6] * d * py - D(obs[1]);
}
template <typename T> inline void user_model_residual(int kind, const T *cam, const T *pt, const T *obs, const T *data, T *res) {
  user_model_error<T, T>(kind, cam, pt, obs, data, res);
}
// residual + blocks (E x d column-major, the pose block zero-padded to 9 columns)
template <typename T>
inline void user_model_residual_jacobian(int kind, const T *cam, const T *pt, const T *obs, const T *data, T *res, T *Jc, T *Jp) {
  using D = DualNumber<T>;
  const int dc = kind == MODEL_PINHOLE ? 6 : 9;
  user_model_error<T, T>(kind, cam, pt, obs, data, res);
  for (int i = 0; i < 18; ++i) Jc[i] = T(0);
  for (int col = 0; col < dc + 3; ++col) {
    D c[9], p[3], e[2];
    for (int k = 0; k < 9; ++k) c[k] = D(cam[k]);
    for (int k = 0; k < 3; ++k) p[k] = D(pt[k]);
    (col < dc ? c[col] : p[col - dc]).dual = T(1);
    user_model_error<D, T>(kind, c, p, obs, data, e);
    T *dst = col < dc ? &Jc[2 * col] : &Jp[2 * (col - dc)];
    dst[0] = e[0].dual; dst[1] = e[1].dual;
  }
}

} // namespace gro
