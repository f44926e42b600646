// ORACLE — test infrastructure only.  Never linked into or called from the
// product path (graphite_amd/, libgraphite_mi355x.so).  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
//
// CPU restatement of the BAL camera model used on the reference's hot path.
//   residual : /root/reference/examples/reprojection_error.cuh:61-99
//              (bal_reprojection_error_simple: Eigen::AngleAxis rotation
//              matrix when theta > 0, identity otherwise, translate,
//              -P.xy/P.z, radial distortion, times f, minus observation)
//   jacobian : /root/reference/examples/projection_jacobians.cuh:2-322 is a
//              wrenfold-generated straight-line program for the analytic
//              derivative of that same projection (quaternion half-angle
//              form), with the rotation derivative forced to ZERO when
//              theta == 0 (:175-212).  The derivative is restated here from
//              the closed form d(R X)/dr = -R [X]x Jr(r) (right Jacobian of
//              SO(3)); it is the same function, so values agree to rounding.
//              The theta == 0 quirk is kept.
//   layout   : Jacobian blocks are E x d column-major, E = 2
//              (/root/reference/examples/reprojection_error.cuh:117-125).
//
// Third-party arithmetic restated: Eigen 3.4 AngleAxis::toRotationMatrix
// (un-vendored dependency, CMakeLists.txt:25) — the published Rodrigues form
//   R = c I + s [a]x + (1-c) a a^T  evaluated entry-wise as Eigen does.
// Parity status of the residual VALUES: "parity unpinned" against the
// reference (its tests only assert chi2 != 0, tests/schur.cu:144); pinned here
// by central finite differences and committed fixtures (tests/golden/).
#pragma once
#include <cmath>

namespace gro {

template <typename T> struct SO3Coeffs {
  T a; // sin(theta)/theta
  T b; // (1-cos(theta))/theta^2
  T c; // (theta - sin(theta))/theta^3
};

// Coefficients of the right Jacobian Jr(r) = a I - b [r]x + c r r^T.
// Series below theta^2 < 0.25 avoids the cancellation in c; product code
// (graphite_amd/csrc/bal_model.hpp) uses the same split so fp32 parity is tight.
template <typename T> inline SO3Coeffs<T> so3_coeffs(T theta2) {
  SO3Coeffs<T> k;
  const T theta = std::sqrt(theta2);
  if (theta2 < T(0.25)) {
    const T t2 = theta2;
    k.a = T(1) + t2 * (T(-1.0 / 6) + t2 * (T(1.0 / 120) + t2 * (T(-1.0 / 5040) + t2 * (T(1.0 / 362880) + t2 * T(-1.0 / 39916800)))));
    k.b = T(0.5) + t2 * (T(-1.0 / 24) + t2 * (T(1.0 / 720) + t2 * (T(-1.0 / 40320) + t2 * (T(1.0 / 3628800) + t2 * T(-1.0 / 479001600)))));
    k.c = T(1.0 / 6) + t2 * (T(-1.0 / 120) + t2 * (T(1.0 / 5040) + t2 * (T(-1.0 / 362880) + t2 * (T(1.0 / 39916800) + t2 * T(-1.0 / 6227020800.0)))));
  } else {
    const T s = std::sin(theta), c = std::cos(theta);
    k.a = s / theta;
    k.b = (T(1) - c) / theta2;
    k.c = (theta - s) / (theta2 * theta);
  }
  return k;
}

// Rotation matrix, row-major R[3*i+j].  theta > 0: Eigen AngleAxis form;
// theta == 0: identity (reprojection_error.cuh:72-77).
template <typename T> inline void bal_rotation(const T *rvec, T *R) {
  const T theta2 = rvec[0] * rvec[0] + rvec[1] * rvec[1] + rvec[2] * rvec[2];
  const T theta = std::sqrt(theta2);
  if (theta > T(0)) {
    const T ax = rvec[0] / theta, ay = rvec[1] / theta, az = rvec[2] / theta;
    const T c = std::cos(theta), s = std::sin(theta);
    const T sx = s * ax, sy = s * ay, sz = s * az;
    const T cx = (T(1) - c) * ax, cy = (T(1) - c) * ay, cz = (T(1) - c) * az;
    T tmp;
    tmp = cx * ay; R[1] = tmp - sz; R[3] = tmp + sz;
    tmp = cx * az; R[2] = tmp + sy; R[6] = tmp - sy;
    tmp = cy * az; R[5] = tmp - sx; R[7] = tmp + sx;
    R[0] = cx * ax + c; R[4] = cy * ay + c; R[8] = cz * az + c;
  } else {
    R[0] = 1; R[1] = 0; R[2] = 0;
    R[3] = 0; R[4] = 1; R[5] = 0;
    R[6] = 0; R[7] = 0; R[8] = 1;
  }
}

// residual (2) of one observation.  cam = [r(3) t(3) f k1 k2], pt = X(3).
template <typename T>
inline void bal_residual(const T *cam, const T *pt, const T *obs, T *res) {
  T R[9];
  bal_rotation(cam, R);
  const T Px = R[0] * pt[0] + R[1] * pt[1] + R[2] * pt[2] + cam[3];
  const T Py = R[3] * pt[0] + R[4] * pt[1] + R[5] * pt[2] + cam[4];
  const T Pz = R[6] * pt[0] + R[7] * pt[1] + R[8] * pt[2] + cam[5];
  const T px = -Px / Pz, py = -Py / Pz;
  const T f = cam[6], k1 = cam[7], k2 = cam[8];
  const T r2 = px * px + py * py;
  const T d = T(1) + k1 * r2 + k2 * r2 * r2;
  res[0] = f * d * px - obs[0];
  res[1] = f * d * py - obs[1];
}

// residual + Jacobians.  Jc: 2x9 column-major (18), Jp: 2x3 column-major (6).
template <typename T>
inline void bal_residual_jacobian(const T *cam, const T *pt, const T *obs,
                                  T *res, T *Jc, T *Jp) {
  T R[9];
  bal_rotation(cam, R);
  const T X = pt[0], Y = pt[1], Z = pt[2];
  const T Px = R[0] * X + R[1] * Y + R[2] * Z + cam[3];
  const T Py = R[3] * X + R[4] * Y + R[5] * Z + cam[4];
  const T Pz = R[6] * X + R[7] * Y + R[8] * Z + cam[5];
  const T iz = T(1) / Pz;
  const T px = -Px * iz, py = -Py * iz;
  const T f = cam[6], k1 = cam[7], k2 = cam[8];
  const T r2 = px * px + py * py;
  const T d = T(1) + k1 * r2 + k2 * r2 * r2;
  if (res) {
    res[0] = f * d * px - obs[0];
    res[1] = f * d * py - obs[1];
  }
  // B = d(res)/dp = f (d I + 2 (k1 + 2 k2 r2) p p^T)   (2x2, symmetric)
  const T g = T(2) * (k1 + T(2) * k2 * r2);
  const T B00 = f * (d + g * px * px), B01 = f * g * px * py, B11 = f * (d + g * py * py);
  // dp/dP = [ -iz 0 -px*iz ; 0 -iz -py*iz ]   (since px = -Px/Pz)
  // A = B * dp/dP  (2x3)
  const T A00 = -B00 * iz, A01 = -B01 * iz, A02 = -(B00 * px + B01 * py) * iz;
  const T A10 = -B01 * iz, A11 = -B11 * iz, A12 = -(B01 * px + B11 * py) * iz;
  // Jp = A R   (2x3)
  const T Q00 = A00 * R[0] + A01 * R[3] + A02 * R[6];
  const T Q01 = A00 * R[1] + A01 * R[4] + A02 * R[7];
  const T Q02 = A00 * R[2] + A01 * R[5] + A02 * R[8];
  const T Q10 = A10 * R[0] + A11 * R[3] + A12 * R[6];
  const T Q11 = A10 * R[1] + A11 * R[4] + A12 * R[7];
  const T Q12 = A10 * R[2] + A11 * R[5] + A12 * R[8];
  Jp[0] = Q00; Jp[1] = Q10; Jp[2] = Q01; Jp[3] = Q11; Jp[4] = Q02; Jp[5] = Q12;
  // rotation block: d(res)/dr = -(A R) [X]x Jr(r),  zero when theta == 0.
  const T rx = cam[0], ry = cam[1], rz = cam[2];
  const T theta2 = rx * rx + ry * ry + rz * rz;
  if (std::sqrt(theta2) > T(0)) {
    // M = -(Q [X]x) ;  [X]x = [0 -Z Y; Z 0 -X; -Y X 0]
    const T M00 = -(Q01 * Z - Q02 * Y), M01 = -(-Q00 * Z + Q02 * X), M02 = -(Q00 * Y - Q01 * X);
    const T M10 = -(Q11 * Z - Q12 * Y), M11 = -(-Q10 * Z + Q12 * X), M12 = -(Q10 * Y - Q11 * X);
    const SO3Coeffs<T> k = so3_coeffs(theta2);
    // G = a I - b [r]x + c r r^T
    const T G00 = k.a + k.c * rx * rx, G01 = k.b * rz + k.c * rx * ry, G02 = -k.b * ry + k.c * rx * rz;
    const T G10 = -k.b * rz + k.c * ry * rx, G11 = k.a + k.c * ry * ry, G12 = k.b * rx + k.c * ry * rz;
    const T G20 = k.b * ry + k.c * rz * rx, G21 = -k.b * rx + k.c * rz * ry, G22 = k.a + k.c * rz * rz;
    Jc[0] = M00 * G00 + M01 * G10 + M02 * G20;
    Jc[1] = M10 * G00 + M11 * G10 + M12 * G20;
    Jc[2] = M00 * G01 + M01 * G11 + M02 * G21;
    Jc[3] = M10 * G01 + M11 * G11 + M12 * G21;
    Jc[4] = M00 * G02 + M01 * G12 + M02 * G22;
    Jc[5] = M10 * G02 + M11 * G12 + M12 * G22;
  } else {
    for (int i = 0; i < 6; ++i) Jc[i] = T(0);
  }
  // translation block = A
  Jc[6] = A00; Jc[7] = A10; Jc[8] = A01; Jc[9] = A11; Jc[10] = A02; Jc[11] = A12;
  // intrinsics: f, k1, k2
  Jc[12] = d * px; Jc[13] = d * py;
  Jc[14] = f * r2 * px; Jc[15] = f * r2 * py;
  Jc[16] = f * r2 * r2 * px; Jc[17] = f * r2 * r2 * py;
}

} // namespace gro
