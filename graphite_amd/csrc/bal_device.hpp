// Device-side BAL camera model for gfx950.
//
// Replaces, for BAL graphs, the user-trait calls made by
//   ops::compute_error      /root/reference/include/graphite/ops/error.hpp:253-323
//   ops::compute_jacobians  /root/reference/include/graphite/ops/linearize.hpp:10-138
// with the residual of examples/reprojection_error.cuh:61-99 and the analytic
// Jacobian that examples/projection_jacobians.cuh:2-322 encodes (restated in
// closed form: d(R X)/dr = -R [X]x Jr(r); zero rotation block at theta == 0,
// as :175-212 does).
//
// MI355X design: the reference recomputes sin/cos/sqrt and ~514 flops per
// observation per slot.  Here everything that depends on the camera alone is
// hoisted into a 24-scalar "camera pack" (R, t, f, k1, k2, G = Jr(r)) computed
// once per linearisation by Nc threads; a per-observation evaluation is then
// ~150 FMAs with no transcendental, cheap enough that kernels RECOMPUTE J
// instead of streaming 24 stored scalars per observation from HBM
// (the reference's own set_jacobian_storage(false) mode, factor.hpp:632).
#pragma once
#include <hip/hip_runtime.h>

namespace gr {

constexpr int PACK = 24; // R[0..8] t[9..11] f k1 k2 [12..14] G[15..23]

template <typename T> __device__ __forceinline__ T t_sqrt(T x);
template <> __device__ __forceinline__ float t_sqrt<float>(float x) { return sqrtf(x); }
template <> __device__ __forceinline__ double t_sqrt<double>(double x) { return sqrt(x); }
template <typename T> __device__ __forceinline__ void t_sincos(T x, T *s, T *c);
template <> __device__ __forceinline__ void t_sincos<float>(float x, float *s, float *c) { sincosf(x, s, c); }
template <> __device__ __forceinline__ void t_sincos<double>(double x, double *s, double *c) { sincos(x, s, c); }

// One camera -> pack.  cam = [r(3) t(3) f k1 k2].
template <typename T> __device__ __forceinline__ void make_campack(const T *cam, T *pk) {
  const T rx = cam[0], ry = cam[1], rz = cam[2];
  const T theta2 = rx * rx + ry * ry + rz * rz;
  const T theta = t_sqrt(theta2);
  if (theta > T(0)) {
    // Rodrigues, entry-wise as Eigen::AngleAxis::toRotationMatrix evaluates it
    const T ax = rx / theta, ay = ry / theta, az = rz / theta;
    T s, c;
    t_sincos(theta, &s, &c);
    const T sx = s * ax, sy = s * ay, sz = s * az;
    const T cx = (T(1) - c) * ax, cy = (T(1) - c) * ay, cz = (T(1) - c) * az;
    T tmp;
    tmp = cx * ay; pk[1] = tmp - sz; pk[3] = tmp + sz;
    tmp = cx * az; pk[2] = tmp + sy; pk[6] = tmp - sy;
    tmp = cy * az; pk[5] = tmp - sx; pk[7] = tmp + sx;
    pk[0] = cx * ax + c; pk[4] = cy * ay + c; pk[8] = cz * az + c;
    // G = a I - b [r]x + c r r^T ; series below theta^2 < 0.25 (no cancellation)
    T ka, kb, kc;
    if (theta2 < T(0.25)) {
      const T t2 = theta2;
      ka = T(1) + t2 * (T(-1.0 / 6) + t2 * (T(1.0 / 120) + t2 * (T(-1.0 / 5040) + t2 * (T(1.0 / 362880) + t2 * T(-1.0 / 39916800)))));
      kb = T(0.5) + t2 * (T(-1.0 / 24) + t2 * (T(1.0 / 720) + t2 * (T(-1.0 / 40320) + t2 * (T(1.0 / 3628800) + t2 * T(-1.0 / 479001600)))));
      kc = T(1.0 / 6) + t2 * (T(-1.0 / 120) + t2 * (T(1.0 / 5040) + t2 * (T(-1.0 / 362880) + t2 * (T(1.0 / 39916800) + t2 * T(-1.0 / 6227020800.0)))));
    } else {
      ka = s / theta;
      kb = (T(1) - c) / theta2;
      kc = (theta - s) / (theta2 * theta);
    }
    pk[15] = ka + kc * rx * rx; pk[16] = kb * rz + kc * rx * ry;  pk[17] = -kb * ry + kc * rx * rz;
    pk[18] = -kb * rz + kc * ry * rx; pk[19] = ka + kc * ry * ry; pk[20] = kb * rx + kc * ry * rz;
    pk[21] = kb * ry + kc * rz * rx;  pk[22] = -kb * rx + kc * rz * ry; pk[23] = ka + kc * rz * rz;
  } else {
    pk[0] = 1; pk[1] = 0; pk[2] = 0; pk[3] = 0; pk[4] = 1; pk[5] = 0; pk[6] = 0; pk[7] = 0; pk[8] = 1;
#pragma unroll
    for (int i = 15; i < 24; ++i) pk[i] = T(0); // zero rotation derivative at theta == 0
  }
  pk[9] = cam[3]; pk[10] = cam[4]; pk[11] = cam[5];
  pk[12] = cam[6]; pk[13] = cam[7]; pk[14] = cam[8];
}

// Intermediate of one projection; everything later kernels need.
template <typename T> struct Proj {
  T px, py, r2, d, iz; // normalised image point, radius^2, distortion, 1/Pz
};

// residual only (chi2 pass).  pk may live in registers, LDS or global.
template <typename T>
__device__ __forceinline__ void bal_residual(const T *pk, T X, T Y, T Z, T ox, T oy, T &e0, T &e1) {
  const T Px = pk[0] * X + pk[1] * Y + pk[2] * Z + pk[9];
  const T Py = pk[3] * X + pk[4] * Y + pk[5] * Z + pk[10];
  const T Pz = pk[6] * X + pk[7] * Y + pk[8] * Z + pk[11];
  const T px = -Px / Pz, py = -Py / Pz;
  const T r2 = px * px + py * py;
  const T d = T(1) + pk[13] * r2 + pk[14] * r2 * r2;
  e0 = pk[12] * d * px - ox;
  e1 = pk[12] * d * py - oy;
}

// residual + Jacobians (E x d column-major: Jc[2*col+row], Jp[2*col+row]).
template <typename T>
__device__ __forceinline__ void bal_linearize(const T *pk, T X, T Y, T Z, T ox, T oy, T &e0, T &e1,
                                              T *Jc, T *Jp) {
  const T Px = pk[0] * X + pk[1] * Y + pk[2] * Z + pk[9];
  const T Py = pk[3] * X + pk[4] * Y + pk[5] * Z + pk[10];
  const T Pz = pk[6] * X + pk[7] * Y + pk[8] * Z + pk[11];
  const T iz = T(1) / Pz;
  const T px = -Px * iz, py = -Py * iz;
  const T f = pk[12], k1 = pk[13], k2 = pk[14];
  const T r2 = px * px + py * py;
  const T d = T(1) + k1 * r2 + k2 * r2 * r2;
  e0 = f * d * px - ox;
  e1 = f * d * py - oy;
  const T g = T(2) * (k1 + T(2) * k2 * r2);
  const T B00 = f * (d + g * px * px), B01 = f * g * px * py, B11 = f * (d + g * py * py);
  const T A00 = -B00 * iz, A01 = -B01 * iz, A02 = -(B00 * px + B01 * py) * iz;
  const T A10 = -B01 * iz, A11 = -B11 * iz, A12 = -(B01 * px + B11 * py) * iz;
  const T Q00 = A00 * pk[0] + A01 * pk[3] + A02 * pk[6];
  const T Q01 = A00 * pk[1] + A01 * pk[4] + A02 * pk[7];
  const T Q02 = A00 * pk[2] + A01 * pk[5] + A02 * pk[8];
  const T Q10 = A10 * pk[0] + A11 * pk[3] + A12 * pk[6];
  const T Q11 = A10 * pk[1] + A11 * pk[4] + A12 * pk[7];
  const T Q12 = A10 * pk[2] + A11 * pk[5] + A12 * pk[8];
  Jp[0] = Q00; Jp[1] = Q10; Jp[2] = Q01; Jp[3] = Q11; Jp[4] = Q02; Jp[5] = Q12;
  const T M00 = -(Q01 * Z - Q02 * Y), M01 = -(-Q00 * Z + Q02 * X), M02 = -(Q00 * Y - Q01 * X);
  const T M10 = -(Q11 * Z - Q12 * Y), M11 = -(-Q10 * Z + Q12 * X), M12 = -(Q10 * Y - Q11 * X);
  Jc[0] = M00 * pk[15] + M01 * pk[18] + M02 * pk[21];
  Jc[1] = M10 * pk[15] + M11 * pk[18] + M12 * pk[21];
  Jc[2] = M00 * pk[16] + M01 * pk[19] + M02 * pk[22];
  Jc[3] = M10 * pk[16] + M11 * pk[19] + M12 * pk[22];
  Jc[4] = M00 * pk[17] + M01 * pk[20] + M02 * pk[23];
  Jc[5] = M10 * pk[17] + M11 * pk[20] + M12 * pk[23];
  Jc[6] = A00; Jc[7] = A10; Jc[8] = A01; Jc[9] = A11; Jc[10] = A02; Jc[11] = A12;
  Jc[12] = d * px; Jc[13] = d * py;
  Jc[14] = f * r2 * px; Jc[15] = f * r2 * py;
  Jc[16] = f * r2 * r2 * px; Jc[17] = f * r2 * r2 * py;
}

// Mixed precision (the reference's Graph<T = double, S = float>, bal.cu --precision FP64-FP32): the
// residual stays in T (ops/error.hpp evaluates the error in the graph precision), the Jacobian
// entries are evaluated in JT and promoted; every accumulation downstream stays in T.
template <typename T, typename JT>
__device__ __forceinline__ void bal_linearize_j(const T *pk, T X, T Y, T Z, T ox, T oy, T &e0, T &e1, T *Jc, T *Jp) {
  if constexpr (sizeof(JT) == sizeof(T)) {
    bal_linearize<T>(pk, X, Y, Z, ox, oy, e0, e1, Jc, Jp);
  } else {
    bal_residual<T>(pk, X, Y, Z, ox, oy, e0, e1);
    JT pj[PACK], f0, f1, Jcj[18], Jpj[6];
#pragma unroll
    for (int i = 0; i < PACK; ++i) pj[i] = (JT)pk[i];
    bal_linearize<JT>(pj, (JT)X, (JT)Y, (JT)Z, (JT)ox, (JT)oy, f0, f1, Jcj, Jpj);
#pragma unroll
    for (int i = 0; i < 18; ++i) Jc[i] = (T)Jcj[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) Jp[i] = (T)Jpj[i];
  }
}

// rho'(raw chi2)  — loss.hpp:15-51 (DefaultLoss kind 0, HuberLoss kind 1)
template <typename T> __device__ __forceinline__ T loss_rho(int kind, T delta, T raw) {
  if (kind == 1 && !(raw <= delta * delta)) return T(2) * t_sqrt(raw) * delta - delta * delta;
  return raw;
}
template <typename T> __device__ __forceinline__ T loss_drho(int kind, T delta, T raw) {
  if (kind == 1 && !(raw <= delta * delta)) return delta / t_sqrt(raw);
  return T(1);
}

} // namespace gr
