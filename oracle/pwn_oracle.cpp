/*
 * pwn_oracle.cpp -- CPU ORACLE: restatement of the reference's PWN hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see pwn_oracle.h).  PARITY UNPINNED: the reference cannot be built
 * here (Eigen3/OpenCV missing) and holds no golden vectors; this file follows the reference
 * sources line by line and restates the Eigen routines they call.
 *
 * Arithmetic conventions (must be compiled with -ffp-contract=off, no -ffast-math):
 *   - every inner product is evaluated left to right, ((a0*b0 + a1*b1) + a2*b2) + a3*b3, which is
 *     what Eigen's fixed-size coefficient-based products do and what its SSE3 4-wide reductions
 *     reduce to when the w-term is 0 (true for every dot on this path);
 *   - no fused multiply-add;
 *   - Eigen 3.2.x expression semantics where versions differ (noted at the site).
 *
 * References are relative to /root/reference/g2o_frontend/pwn_core/ unless a directory is given.
 */
#include "pwn_oracle.h"

#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

struct V4 { float v[4]; };
struct M3 { float m[9];  inline float& operator()(int r, int c) { return m[r + 3 * c]; } inline float operator()(int r, int c) const { return m[r + 3 * c]; } };
struct M4 { float m[16]; inline float& operator()(int r, int c) { return m[r + 4 * c]; } inline float operator()(int r, int c) const { return m[r + 4 * c]; } };

inline M4 m4_zero() { M4 r; std::memset(r.m, 0, sizeof(r.m)); return r; }
inline M4 m4_identity() { M4 r = m4_zero(); r(0,0) = r(1,1) = r(2,2) = r(3,3) = 1.0f; return r; }
inline M4 m4_load(const float* p) { M4 r; std::memcpy(r.m, p, sizeof(r.m)); return r; }

/* Matrix4f * Vector4f, coefficient-based: sum over k ascending. */
inline V4 m4_mul_v4(const M4& A, const V4& x) {
  V4 r;
  for (int i = 0; i < 4; ++i) {
    float s = A(i,0) * x.v[0];
    s = s + A(i,1) * x.v[1];
    s = s + A(i,2) * x.v[2];
    s = s + A(i,3) * x.v[3];
    r.v[i] = s;
  }
  return r;
}
/* Isometry3f * 4-vector (Eigen transform_right_product_impl): top 3 rows = affine(3x4)*x, last row copied. */
inline V4 iso_mul_v4(const M4& T, const V4& x) {
  V4 r;
  for (int i = 0; i < 3; ++i) {
    float s = T(i,0) * x.v[0];
    s = s + T(i,1) * x.v[1];
    s = s + T(i,2) * x.v[2];
    s = s + T(i,3) * x.v[3];
    r.v[i] = s;
  }
  r.v[3] = x.v[3];
  return r;
}
inline M4 m4_mul(const M4& A, const M4& B) {
  M4 R;
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 4; ++i) {
      float s = A(i,0) * B(0,j);
      s = s + A(i,1) * B(1,j);
      s = s + A(i,2) * B(2,j);
      s = s + A(i,3) * B(3,j);
      R(i,j) = s;
    }
  return R;
}
inline M4 m4_transpose(const M4& A) { M4 R; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) R(i,j) = A(j,i); return R; }
inline M3 m3_mul(const M3& A, const M3& B) {
  M3 R;
  for (int j = 0; j < 3; ++j)
    for (int i = 0; i < 3; ++i) {
      float s = A(i,0) * B(0,j);
      s = s + A(i,1) * B(1,j);
      s = s + A(i,2) * B(2,j);
      R(i,j) = s;
    }
  return R;
}
inline void m3_mul_v3(const M3& A, const float x[3], float r[3]) {
  for (int i = 0; i < 3; ++i) {
    float s = A(i,0) * x[0];
    s = s + A(i,1) * x[1];
    s = s + A(i,2) * x[2];
    r[i] = s;
  }
}
inline float dot4(const V4& a, const V4& b) {
  float s = a.v[0] * b.v[0];
  s = s + a.v[1] * b.v[1];
  s = s + a.v[2] * b.v[2];
  s = s + a.v[3] * b.v[3];
  return s;
}
inline float sqnorm3(const float a[3]) { float s = a[0]*a[0]; s = s + a[1]*a[1]; s = s + a[2]*a[2]; return s; }

inline M3 iso_linear(const M4& T) { M3 R; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R(i,j) = T(i,j); return R; }
inline void iso_set(M4& T, const M3& R, const float t[3]) {
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) T(i,j) = R(i,j); T(i,3) = t[i]; }
  T(3,0) = 0.f; T(3,1) = 0.f; T(3,2) = 0.f; T(3,3) = 1.f;
}
inline void force_last_row(M4& T) { T(3,0) = 0.f; T(3,1) = 0.f; T(3,2) = 0.f; T(3,3) = 1.f; }

/* Eigen Transform<float,3,Isometry>::inverse(Isometry): R' = R^T, t' = (-R^T) * t, makeAffine. */
inline M4 iso_inverse(const M4& T) {
  M3 Rt; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rt(i,j) = -T(j,i);
  float t[3] = { T(0,3), T(1,3), T(2,3) }, ti[3];
  m3_mul_v3(Rt, t, ti);
  M3 R; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R(i,j) = T(j,i);
  M4 r; iso_set(r, R, ti); return r;
}
/* Eigen transform_transform_product_impl (same affine mode): linear = Rl*Rr, translation = Rl*tr + tl. */
inline M4 iso_mul(const M4& A, const M4& B) {
  M3 Ra = iso_linear(A), Rb = iso_linear(B);
  M3 R = m3_mul(Ra, Rb);
  float tb[3] = { B(0,3), B(1,3), B(2,3) }, t[3];
  m3_mul_v3(Ra, tb, t);
  for (int i = 0; i < 3; ++i) t[i] = t[i] + A(i,3);
  M4 r; iso_set(r, R, t); return r;
}

/* Eigen Matrix3f::inverse() (compute_inverse_size3_helper: cofactors, det along column 0). */
inline float cofactor3(const M3& m, int i, int j) {
  const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m(i1,j1) * m(i2,j2) - m(i1,j2) * m(i2,j1);
}
inline M3 m3_inverse(const M3& m) {
  float c0[3] = { cofactor3(m,0,0), cofactor3(m,1,0), cofactor3(m,2,0) };
  float det = c0[0] * m(0,0); det = det + c0[1] * m(1,0); det = det + c0[2] * m(2,0);
  const float invdet = 1.0f / det;
  M3 r;
  r(0,0) = c0[0] * invdet; r(0,1) = c0[1] * invdet; r(0,2) = c0[2] * invdet;
  r(1,0) = cofactor3(m,0,1) * invdet; r(1,1) = cofactor3(m,1,1) * invdet; r(1,2) = cofactor3(m,2,1) * invdet;
  r(2,0) = cofactor3(m,0,2) * invdet; r(2,1) = cofactor3(m,1,2) * invdet; r(2,2) = cofactor3(m,2,2) * invdet;
  return r;
}

/* ---------------------------------------------------------------- bm_se3.h:9-52 ------------- */
inline M3 quat2mat(const float q[3]) {           /* bm_se3.h:9-22 */
  const float qx = q[0], qy = q[1], qz = q[2];
  const float qw = std::sqrt(1.f - sqnorm3(q));
  M3 R;
  R(0,0) = qw*qw + qx*qx - qy*qy - qz*qz; R(0,1) = 2*(qx*qy - qw*qz);           R(0,2) = 2*(qx*qz + qw*qy);
  R(1,0) = 2*(qx*qy + qz*qw);            R(1,1) = qw*qw - qx*qx + qy*qy - qz*qz; R(1,2) = 2*(qy*qz - qx*qw);
  R(2,0) = 2*(qx*qz - qy*qw);            R(2,1) = 2*(qy*qz + qx*qw);            R(2,2) = qw*qw - qx*qx - qy*qy + qz*qz;
  return R;
}
/* Eigen Quaternionf(Matrix3f) (quaternionbase_assign_impl<Other,3,3>) + normalize(); bm_se3.h:25-35 */
inline void mat2quat(const M3& mat, float rq[3]) {
  float q[4];  /* x y z w */
  float t = (mat(0,0) + mat(1,1)) + mat(2,2);
  if (t > 0.f) {
    t = std::sqrt(t + 1.0f);
    q[3] = 0.5f * t;
    t = 0.5f / t;
    q[0] = (mat(2,1) - mat(1,2)) * t;
    q[1] = (mat(0,2) - mat(2,0)) * t;
    q[2] = (mat(1,0) - mat(0,1)) * t;
  } else {
    int i = 0;
    if (mat(1,1) > mat(0,0)) i = 1;
    if (mat(2,2) > mat(i,i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(mat(i,i) - mat(j,j) - mat(k,k) + 1.0f);
    q[i] = 0.5f * t;
    t = 0.5f / t;
    q[3] = (mat(k,j) - mat(j,k)) * t;
    q[j] = (mat(j,i) + mat(i,j)) * t;
    q[k] = (mat(k,i) + mat(i,k)) * t;
  }
  /* normalize(): coeffs /= sqrt(squaredNorm), SSE3 hadd order (x2+y2)+(z2+w2) */
  const float n2 = (q[0]*q[0] + q[1]*q[1]) + (q[2]*q[2] + q[3]*q[3]);
  const float n = std::sqrt(n2);
  for (int a = 0; a < 4; ++a) q[a] = q[a] / n;
  rq[0] = q[0]; rq[1] = q[1]; rq[2] = q[2];
  if (q[3] < 0.f) { rq[0] = -rq[0]; rq[1] = -rq[1]; rq[2] = -rq[2]; }
}
inline M4 v2t(const float x[6]) {                 /* bm_se3.h:37-43 */
  M3 R = quat2mat(x + 3);
  M4 T; iso_set(T, R, x); return T;
}
inline void t2v(const M4& T, float v[6]) {        /* bm_se3.h:45-52 */
  v[0] = T(0,3); v[1] = T(1,3); v[2] = T(2,3);
  mat2quat(iso_linear(T), v + 3);
}
/* bm_se3.h:54-66 skew on a 4-vector: S = -2 [v]x embedded in 4x4 */
inline M4 skew4(const V4& v) {
  const float tx = 2 * v.v[0], ty = 2 * v.v[1], tz = 2 * v.v[2];
  M4 S = m4_zero();
  S(0,1) = tz;  S(1,0) = -tz;
  S(0,2) = -ty; S(2,0) = ty;
  S(1,2) = tx;  S(2,1) = -tx;
  return S;
}

/* --------------------------------------- Eigen SelfAdjointEigenSolver<Matrix3f>::computeDirect */
/* The reference calls std::atan2/cos/sin on floats, i.e. the host libm's atan2f/cosf/sinf, whose last
 * bit depends on the libm version.  Canonical mode 0 (default) evaluates them in double precision by the
 * fixed + - * / algorithms below and rounds once to float, which any platform reproduces bit for bit
 * (and which is the correctly rounded value except about once in 1e8 calls); mode 1 calls the float libm
 * literally, mode 2 the double libm.  tests/ quantify the differences between the modes. */
#define PWN_TRIG_HD static inline
#define PWN_TRIG_SIGNBIT(x) std::signbit(x)

// ---- trig of the 3x3 eigensolver, double precision, + - * / only -----------------------------------------------------
// The eigensolver needs atan2(sqrt(q), half_b) / 3 and cos / sin of that angle as FLOATS (the reference calls the float libm).
// Canonical evaluation: in double by the fixed algorithms below (no libm, no FMA: every operation is an IEEE + - * /, so host
// and device produce the same bits), rounded once to float.  Their error is < 1e-15, so the rounded float is the correctly
// rounded value -- what a current glibc's atan2f / cosf / sinf return -- except when the true value lies within 1e-15 of a
// rounding boundary (about one call in 1e8).
//   atan2(y, x), y >= 0:  t = min/max in [0, 1]; centre c_k = tan(k pi/16), k = 0..4, nearest in angle;
//                         u = (min - c max) / (max + c min), |u| <= tan(pi/32); atan(u) by its series to u^19; unfold.
//   sin / cos on [0, pi/3]: Taylor series to x^21 / x^22.
PWN_TRIG_HD double orc_atan2_pos(double y, double x) {
  const double kPi = 3.141592653589793, kPi2 = 1.5707963267948966;
  if (y == 0.0) return (x < 0.0 || (x == 0.0 && PWN_TRIG_SIGNBIT(x))) ? kPi : 0.0;
  const double ax = x < 0.0 ? -x : x;
  const bool swap = ax < y;
  const double num = swap ? ax : y, den = swap ? y : ax;
  double c, ac;
  if (num > 0.8206787908286602 * den)       { c = 1.0;                 ac = 0.7853981633974483; }
  else if (num > 0.5345111359507916 * den)  { c = 0.6681786379192989;  ac = 0.5890486225480862; }
  else if (num > 0.3033466836073424 * den)  { c = 0.41421356237309503; ac = 0.39269908169872414; }
  else if (num > 0.09849140335716425 * den) { c = 0.198912367379658;   ac = 0.19634954084936207; }
  else                                      { c = 0.0;                 ac = 0.0; }
  const double u = (num - c * den) / (den + c * num);
  const double w = u * u;
  double p = -0.05263157894736842;
  p = p * w + 0.058823529411764705;
  p = p * w + -0.06666666666666667;
  p = p * w + 0.07692307692307693;
  p = p * w + -0.09090909090909091;
  p = p * w + 0.1111111111111111;
  p = p * w + -0.14285714285714285;
  p = p * w + 0.2;
  p = p * w + -0.3333333333333333;
  p = p * w + 1.0;
  double r = ac + u * p;
  if (swap) r = kPi2 - r;
  if (x < 0.0) r = kPi - r;
  return r;
}
PWN_TRIG_HD void orc_sincos_small(double x, double& s, double& c) {      // 0 <= x <= pi/3 (a little beyond is fine)
  const double z = x * x;
  double pc = -8.896791392450574e-22;
  pc = pc * z + 4.110317623312165e-19;
  pc = pc * z + -1.5619206968586225e-16;
  pc = pc * z + 4.779477332387385e-14;
  pc = pc * z + -1.1470745597729725e-11;
  pc = pc * z + 2.08767569878681e-09;
  pc = pc * z + -2.755731922398589e-07;
  pc = pc * z + 2.48015873015873e-05;
  pc = pc * z + -0.001388888888888889;
  pc = pc * z + 0.041666666666666664;
  pc = pc * z + -0.5;
  pc = pc * z + 1.0;
  double ps = 1.9572941063391263e-20;
  ps = ps * z + -8.22063524662433e-18;
  ps = ps * z + 2.8114572543455206e-15;
  ps = ps * z + -7.647163731819816e-13;
  ps = ps * z + 1.6059043836821613e-10;
  ps = ps * z + -2.505210838544172e-08;
  ps = ps * z + 2.7557319223985893e-06;
  ps = ps * z + -0.0001984126984126984;
  ps = ps * z + 0.008333333333333333;
  ps = ps * z + -0.16666666666666666;
  ps = ps * z + 1.0;
  s = x * ps; c = pc;
}

/* mode 0 (canonical): the fixed double-precision algorithms above, rounded once to float (what the kernels evaluate, operation for
 * operation); mode 1: the float libm literally, as the reference; mode 2: the double libm rounded to float (the correctly rounded value
 * up to libm's own error; tests/ check that mode 0 agrees with it). */
int g_trig_mode = 0;
inline float trig_atan2(float y, float x) {
  if (g_trig_mode == 1) return std::atan2(y, x);
  if (g_trig_mode == 2) return (float)std::atan2((double)y, (double)x);
  return (float)orc_atan2_pos((double)y, (double)x);
}
inline float trig_cos(float x) {
  if (g_trig_mode == 1) return std::cos(x);
  if (g_trig_mode == 2) return (float)std::cos((double)x);
  double s, c; orc_sincos_small((double)x, s, c); return (float)c;
}
inline float trig_sin(float x) {
  if (g_trig_mode == 1) return std::sin(x);
  if (g_trig_mode == 2) return (float)std::sin((double)x);
  double s, c; orc_sincos_small((double)x, s, c); return (float)s;
}
inline void cross3(const float a[3], const float b[3], float r[3]) {
  r[0] = a[1]*b[2] - a[2]*b[1];
  r[1] = a[2]*b[0] - a[0]*b[2];
  r[2] = a[0]*b[1] - a[1]*b[0];
}
inline void eig3_roots(const M3& m, float roots[3]) {
  const float s_inv3 = 1.0f / 3.0f;
  const float s_sqrt3 = std::sqrt(3.0f);
  float c0 = m(0,0)*m(1,1)*m(2,2) + 2.0f*m(1,0)*m(2,0)*m(2,1) - m(0,0)*m(2,1)*m(2,1) - m(1,1)*m(2,0)*m(2,0) - m(2,2)*m(1,0)*m(1,0);
  float c1 = m(0,0)*m(1,1) - m(1,0)*m(1,0) + m(0,0)*m(2,2) - m(2,0)*m(2,0) + m(1,1)*m(2,2) - m(2,1)*m(2,1);
  float c2 = m(0,0) + m(1,1) + m(2,2);
  float c2_over_3 = c2 * s_inv3;
  float a_over_3 = (c2 * c2_over_3 - c1) * s_inv3;
  a_over_3 = std::max(a_over_3, 0.0f);
  float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
  float q = a_over_3 * a_over_3 * a_over_3 - half_b * half_b;
  q = std::max(q, 0.0f);
  float rho = std::sqrt(a_over_3);
  float theta = trig_atan2(std::sqrt(q), half_b) * s_inv3;
  float cos_theta = trig_cos(theta);
  float sin_theta = trig_sin(theta);
  roots[0] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
  roots[1] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
  roots[2] = c2_over_3 + 2.0f * rho * cos_theta;
}
inline void eig3_extract_kernel(const M3& mat, float res[3], float representative[3]) {
  int i0 = 0; float best = std::fabs(mat(0,0));
  if (std::fabs(mat(1,1)) > best) { best = std::fabs(mat(1,1)); i0 = 1; }
  if (std::fabs(mat(2,2)) > best) { best = std::fabs(mat(2,2)); i0 = 2; }
  for (int i = 0; i < 3; ++i) representative[i] = mat(i, i0);
  float col1[3], col2[3], c0[3], c1[3];
  for (int i = 0; i < 3; ++i) { col1[i] = mat(i, (i0 + 1) % 3); col2[i] = mat(i, (i0 + 2) % 3); }
  cross3(representative, col1, c0);
  cross3(representative, col2, c1);
  const float n0 = sqnorm3(c0), n1 = sqnorm3(c1);
  if (n0 > n1) { const float s = std::sqrt(n0); for (int i = 0; i < 3; ++i) res[i] = c0[i] / s; }
  else         { const float s = std::sqrt(n1); for (int i = 0; i < 3; ++i) res[i] = c1[i] / s; }
}
/* A: symmetric, lower triangle read.  evals ascending, evecs column-major (column k = eigenvector k). */
void eig3_direct(const M3& A, float evals[3], M3& evecs) {
  const float eps = FLT_EPSILON;
  const float shift = ((A(0,0) + A(1,1)) + A(2,2)) / 3.0f;
  M3 S;
  for (int i = 0; i < 3; ++i) for (int j = 0; j <= i; ++j) { S(i,j) = A(i,j); S(j,i) = A(i,j); }
  for (int i = 0; i < 3; ++i) S(i,i) = S(i,i) - shift;
  float scale = 0.f;
  for (int k = 0; k < 9; ++k) scale = std::max(scale, std::fabs(S.m[k]));
  if (scale > 0.f) for (int k = 0; k < 9; ++k) S.m[k] = S.m[k] / scale;
  eig3_roots(S, evals);
  if ((evals[2] - evals[0]) <= eps) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) evecs(i,j) = (i == j) ? 1.f : 0.f;
  } else {
    M3 tmp = S;
    float d0 = evals[2] - evals[1];
    float d1 = evals[1] - evals[0];
    int k = 0, l = 2;
    if (d0 > d1) { std::swap(k, l); d0 = d1; }
    float vk[3], vl[3];
    for (int i = 0; i < 3; ++i) tmp(i,i) = tmp(i,i) - evals[k];
    eig3_extract_kernel(tmp, vk, vl);
    if (d0 <= 2 * eps * d1) {
      float d = vk[0]*vl[0]; d = d + vk[1]*vl[1]; d = d + vk[2]*vl[2];
      for (int i = 0; i < 3; ++i) vl[i] = vl[i] - d * vl[i];
      const float n = std::sqrt(sqnorm3(vl));
      for (int i = 0; i < 3; ++i) vl[i] = vl[i] / n;
    } else {
      tmp = S;
      for (int i = 0; i < 3; ++i) tmp(i,i) = tmp(i,i) - evals[l];
      float dummy[3];
      eig3_extract_kernel(tmp, vl, dummy);
    }
    for (int i = 0; i < 3; ++i) { evecs(i,k) = vk[i]; evecs(i,l) = vl[i]; }
    float v2[3] = { evecs(0,2), evecs(1,2), evecs(2,2) }, v0[3] = { evecs(0,0), evecs(1,0), evecs(2,0) }, v1[3];
    cross3(v2, v0, v1);
    const float z = sqnorm3(v1);
    if (z > 0.f) { const float n = std::sqrt(z); for (int i = 0; i < 3; ++i) v1[i] = v1[i] / n; }
    for (int i = 0; i < 3; ++i) evecs(i,1) = v1[i];
  }
  for (int i = 0; i < 3; ++i) { evals[i] = evals[i] * scale; evals[i] = evals[i] + shift; }
}

/* ----------------------------------------------------- Eigen LDLT<Matrix6f>::solve (fp32) ---- */
void ldlt_solve6(const float Hin[36], const float bin[6], float x[6]) {
  const int n = 6;
  float A[36]; std::memcpy(A, Hin, sizeof(A));
  auto a = [&](int r, int c) -> float& { return A[r + n * c]; };
  int tr[6]; float temp[6];
  float cutoff = 0.f;
  for (int k = 0; k < n; ++k) {
    int p = k; float big = std::fabs(a(k,k));
    for (int i = k + 1; i < n; ++i) if (std::fabs(a(i,i)) > big) { big = std::fabs(a(i,i)); p = i; }
    if (k == 0) cutoff = std::fabs(FLT_EPSILON * big);
    if (big < cutoff) { for (int i = k; i < n; ++i) tr[i] = i; break; }
    tr[k] = p;
    if (k != p) {
      const int s = n - p - 1;
      for (int j = 0; j < k; ++j) std::swap(a(k,j), a(p,j));
      for (int i = 0; i < s; ++i) std::swap(a(p + 1 + i, k), a(p + 1 + i, p));
      std::swap(a(k,k), a(p,p));
      for (int i = k + 1; i < p; ++i) { float t = a(i,k); a(i,k) = a(p,i); a(p,i) = t; }
    }
    const int rs = n - k - 1;
    if (k > 0) {
      for (int j = 0; j < k; ++j) temp[j] = a(j,j) * a(k,j);
      float d = 0.f;
      for (int j = 0; j < k; ++j) { if (j == 0) d = a(k,0) * temp[0]; else d = d + a(k,j) * temp[j]; }
      a(k,k) = a(k,k) - d;
      for (int i = 0; i < rs; ++i) {
        float s2 = 0.f;
        for (int j = 0; j < k; ++j) { if (j == 0) s2 = a(k+1+i,0) * temp[0]; else s2 = s2 + a(k+1+i,j) * temp[j]; }
        a(k+1+i,k) = a(k+1+i,k) - s2;
      }
    }
    if (rs > 0 && std::fabs(a(k,k)) > cutoff)
      for (int i = 0; i < rs; ++i) a(k+1+i,k) = a(k+1+i,k) / a(k,k);
  }
  float d[6]; std::memcpy(d, bin, sizeof(d));
  for (int k = 0; k < n; ++k) if (tr[k] != k) std::swap(d[k], d[tr[k]]);
  for (int i = 0; i < n; ++i) {                       /* L^-1 (unit lower) */
    float s = 0.f;
    for (int j = 0; j < i; ++j) { if (j == 0) s = a(i,0) * d[0]; else s = s + a(i,j) * d[j]; }
    if (i > 0) d[i] = d[i] - s;
  }
  float dmax = 0.f; for (int i = 0; i < n; ++i) dmax = std::max(dmax, std::fabs(a(i,i)));
  const float tol = std::max(dmax * FLT_EPSILON, 1.0f / FLT_MAX);
  for (int i = 0; i < n; ++i) { if (std::fabs(a(i,i)) > tol) d[i] = d[i] / a(i,i); else d[i] = 0.f; }
  for (int i = n - 1; i >= 0; --i) {                  /* L^-T */
    float s = 0.f; bool first = true;
    for (int j = i + 1; j < n; ++j) { if (first) { s = a(j,i) * d[j]; first = false; } else s = s + a(j,i) * d[j]; }
    if (!first) d[i] = d[i] - s;
  }
  for (int k = n - 1; k >= 0; --k) if (tr[k] != k) std::swap(d[k], d[tr[k]]);
  std::memcpy(x, d, sizeof(d));
}

/* ------------------------------------------------------------------- stats.h:13-121 --------- */
struct Stats {
  M4 m; float eig[3]; int n; mutable bool curvatureComputed; mutable float curv;
  Stats() { n = 0; m = m4_identity(); eig[0] = eig[1] = eig[2] = 0.f; curvatureComputed = false; curv = 1.0f; }  /* stats.h:21-27 */
  float curvature() const {                                                                  /* stats.h:98-103 */
    if (!curvatureComputed) curv = (float)((double)eig[0] / ((double)(eig[0] + eig[1] + eig[2]) + 1e-9));
    curvatureComputed = true;
    return curv;
  }
};

/* pinholepointprojector.cpp:17-31 */
struct Projector {
  M3 K, iK; M4 T, KRt, iKRt; float minD, maxD;
  void update() {
    M4 t = iso_inverse(T);
    force_last_row(t);
    iK = m3_inverse(K);
    M3 KR = m3_mul(K, iso_linear(t));
    float tt[3] = { t(0,3), t(1,3), t(2,3) }, Kt[3];
    m3_mul_v3(K, tt, Kt);
    M3 iKR = m3_mul(iso_linear(T), iK);
    KRt = m4_identity(); iKRt = m4_identity();
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) { KRt(i,j) = KR(i,j); iKRt(i,j) = iKR(i,j); } KRt(i,3) = Kt[i]; iKRt(i,3) = T(i,3); }
  }
  /* pinholepointprojector.h:224-233 */
  inline bool project(int& x, int& y, float& d, const V4& p) const {
    V4 ip = m4_mul_v4(KRt, p);
    d = ip.v[2];
    if (d < minD || d > maxD) return false;
    const float inv = 1.0f / d;
    x = (int)std::round(ip.v[0] * inv);
    y = (int)std::round(ip.v[1] * inv);
    return true;
  }
  /* pinholepointprojector.h:246-251 */
  inline bool unProject(V4& p, int x, int y, float d) const {
    if (d < minD || d > maxD) return false;
    V4 in = { { x * d, y * d, d, 1.0f } };
    p = m4_mul_v4(iKRt, in);
    p.v[3] = 1.0f;           /* Point::operator= forces w (homogeneousvector4f.h:56-60) */
    return true;
  }
  /* pinholepointprojector.h:264-274 */
  inline int projectInterval(float d, float worldRadius) const {
    if (d < minD || d > maxD) return -1;
    float v[3] = { worldRadius, worldRadius, 0.f }, p[3];
    m3_mul_v3(K, v, p);
    const float inv = 1.0f / d;
    p[0] = p[0] * inv; p[1] = p[1] * inv;
    if (p[0] > p[1]) return (int)p[0];
    return (int)p[1];
  }
};

}  // namespace

/* basemath/gaussian.h:9-94 (Gaussian<float,3>): moments and information form, each cached lazily behind a flag */
struct Gauss {
  M3 cov, info; float mean[3], infoVec[3]; bool momentsUpdated, infoUpdated;
  void updateMoments() {                               /* gaussian.h:73-79 */
    if (momentsUpdated) return;
    cov = m3_inverse(info);
    m3_mul_v3(cov, infoVec, mean);
    momentsUpdated = true;
  }
  void updateInfo() {                                  /* gaussian.h:81-87 */
    if (infoUpdated) return;
    info = m3_inverse(cov);
    m3_mul_v3(info, mean, infoVec);
    infoUpdated = true;
  }
  void addInformation(Gauss& g) {                      /* gaussian.h:47-53 (g's information form is computed and cached in g) */
    updateInfo();
    g.updateInfo();
    for (int k = 0; k < 9; ++k) info.m[k] = info.m[k] + g.info.m[k];
    for (int k = 0; k < 3; ++k) infoVec[k] = infoVec[k] + g.infoVec[k];
    momentsUpdated = false;
  }
  static Gauss fromMoments(const float m[3], const M3& c) {   /* gaussian.h:34-45, useInfoForm = false */
    Gauss g; std::memset(&g, 0, sizeof(g));
    for (int k = 0; k < 3; ++k) g.mean[k] = m[k];
    g.cov = c; g.momentsUpdated = true; g.infoUpdated = false;
    return g;
  }
};

struct orc_cloud {
  std::vector<V4> points, normals;
  std::vector<Stats> stats;
  std::vector<M4> omegaP, omegaN;
  std::vector<Gauss> gaussians;      /* sensor-noise Gaussians of unProject (pinholepointprojector.cpp:114-123); empty unless requested */
};

namespace {

/* 0 (default): CorrespondenceFinder::compute and Linearizer::update run their canonical one-thread loops whatever the OpenMP thread
 * count is (parity tests, golden vectors).  1: they use the reference's own thread partition without its remainder dropping -- the
 * timed CPU baseline of bench.py only (orc_set_parallel_align). */
int g_parallel_align = 0;

/* ----------------------------------------- pointaccumulator.h / pointintegralimage.cpp ------ */
/* 10 unique channels of (sum, squaredSum): x y z n xx xy xz yy yz zz.  The other 10 entries of the
 * reference's 4+16 floats are bitwise duplicates (p_i*p_j commutes, x*1 == x). */
enum { CH = 10 };
struct Acc { float c[CH]; };

void integral_image(const int* index, const V4* points, int rows, int cols, std::vector<Acc>& I) {
  I.assign((size_t)rows * cols, Acc{ {0,0,0,0,0,0,0,0,0,0} });
  /* pointintegralimage.cpp:16-27: accumulator(r=img x, c=img y) += point */
#pragma omp parallel for
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) {
      const int idx = index[(size_t)r * cols + c];
      if (idx < 0) continue;
      const float* p = points[idx].v;
      Acc& a = I[(size_t)r * cols + c];
      a.c[0] += p[0]; a.c[1] += p[1]; a.c[2] += p[2]; a.c[3] += p[3];
      a.c[4] += p[0]*p[0]; a.c[5] += p[0]*p[1]; a.c[6] += p[0]*p[2];
      a.c[7] += p[1]*p[1]; a.c[8] += p[1]*p[2]; a.c[9] += p[2]*p[2];
    }
  /* :30-35 prefix along image x inside every image row, sequential */
#pragma omp parallel for
  for (int r = 0; r < rows; ++r)
    for (int c = 1; c < cols; ++c) {
      Acc& a = I[(size_t)r * cols + c]; const Acc& b = I[(size_t)r * cols + c - 1];
      for (int k = 0; k < CH; ++k) a.c[k] = a.c[k] + b.c[k];
    }
  /* :38-43 prefix along image y inside every image column, sequential */
#pragma omp parallel for
  for (int c = 0; c < cols; ++c)
    for (int r = 1; r < rows; ++r) {
      Acc& a = I[(size_t)r * cols + c]; const Acc& b = I[(size_t)(r - 1) * cols + c];
      for (int k = 0; k < CH; ++k) a.c[k] = a.c[k] + b.c[k];
    }
}
inline int clampi(int v, int lo, int hi) { v = (v < lo) ? lo : v; v = (v > hi) ? hi : v; return v; }
/* pointintegralimage.cpp:53-66; x indexes image columns, y image rows */
inline Acc get_region(const std::vector<Acc>& I, int rows, int cols, int xmin, int xmax, int ymin, int ymax) {
  xmin = clampi(xmin - 1, 0, cols - 1);
  xmax = clampi(xmax - 1, 0, cols - 1);
  ymin = clampi(ymin - 1, 0, rows - 1);
  ymax = clampi(ymax - 1, 0, rows - 1);
  Acc pa = I[(size_t)ymax * cols + xmax];
  const Acc& b = I[(size_t)ymin * cols + xmin];
  const Acc& c = I[(size_t)ymax * cols + xmin];
  const Acc& d = I[(size_t)ymin * cols + xmax];
  for (int k = 0; k < CH; ++k) { pa.c[k] = pa.c[k] + b.c[k]; pa.c[k] = pa.c[k] - c.c[k]; pa.c[k] = pa.c[k] - d.c[k]; }
  return pa;
}

Projector make_projector(const float K[9], const float T[16], float minD, float maxD) {
  Projector pr; std::memcpy(pr.K.m, K, sizeof(pr.K.m)); pr.T = m4_load(T); pr.minD = minD; pr.maxD = maxD; pr.update(); return pr;
}

/* statscalculatorintegralimage.cpp:14-82 */
void stats_compute(const orc_converter_params* P, const int* index, const int* interval, int rows, int cols,
                   const std::vector<V4>& points, std::vector<V4>& normals, std::vector<Stats>& stats) {
  const size_t M = points.size();
  normals.assign(M, V4{ {0,0,0,0} });
  stats.assign(M, Stats());
  std::vector<Acc> I;
  integral_image(index, points.data(), rows, cols, I);
#pragma omp parallel for schedule(dynamic, 4)
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) {
      const int idx = index[(size_t)r * cols + c], itv = interval[(size_t)r * cols + c];
      if (idx < 0 || itv < 0) continue;
      int rad = itv;
      if (rad < P->min_image_radius) rad = P->min_image_radius;
      if (rad > P->max_image_radius) rad = P->max_image_radius;
      const Acc acc = get_region(I, rows, cols, c - rad, c + rad, r - rad, r + rad);
      const int n = (int)acc.c[3];
      if (n < P->min_points) continue;
      /* pointaccumulator.h:66-86 */
      float d = acc.c[3];
      M3 cov; float mean[3] = {0,0,0};
      if (d) {
        d = 1.0f / d;
        mean[0] = acc.c[0] * d; mean[1] = acc.c[1] * d; mean[2] = acc.c[2] * d;
        const float sq[3][3] = { { acc.c[4], acc.c[5], acc.c[6] }, { acc.c[5], acc.c[7], acc.c[8] }, { acc.c[6], acc.c[8], acc.c[9] } };
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) cov(i,j) = sq[i][j] * d - mean[i] * mean[j];
      } else { for (int k = 0; k < 9; ++k) cov.m[k] = 0.f; }
      float ev[3]; M3 U;
      eig3_direct(cov, ev, U);
      Stats& s = stats[idx];
      s.m = m4_zero();
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) s.m(i,j) = U(i,j);
      s.m(0,3) = mean[0]; s.m(1,3) = mean[1]; s.m(2,3) = mean[2]; s.m(3,3) = 1.0f;
      if (ev[0] < 0.0f) ev[0] = 0.0f;
      s.eig[0] = ev[0]; s.eig[1] = ev[1]; s.eig[2] = ev[2];
      s.n = n;
      V4 nrm = { { s.m(0,0), s.m(1,0), s.m(2,0), 0.f } };
      if (s.curvature() < P->stats_curvature_threshold) {
        if (dot4(nrm, points[idx]) > 0) { nrm.v[0] = -nrm.v[0]; nrm.v[1] = -nrm.v[1]; nrm.v[2] = -nrm.v[2]; nrm.v[3] = 0.f; }
      } else { nrm = V4{ {0,0,0,0} }; }
      normals[idx] = nrm;
    }
}

inline M4 diag4(const float d[3]) { M4 r = m4_zero(); r(0,0) = d[0]; r(1,1) = d[1]; r(2,2) = d[2]; return r; }
inline void info_zero_border(M4& m) { for (int i = 0; i < 4; ++i) { m(3,i) = 0.f; m(i,3) = 0.f; } }

/* informationmatrixcalculator.cpp:9-58 */
void info_compute(const orc_converter_params* P, const std::vector<Stats>& stats, const std::vector<V4>& normals,
                  std::vector<M4>& omegaP, std::vector<M4>& omegaN) {
  const size_t M = stats.size();
  omegaP.assign(M, m4_zero()); omegaN.assign(M, m4_zero());
  const M4 flatP = diag4(P->point_flat_diag), flatN = diag4(P->normal_flat_diag), nonflatN = diag4(P->normal_nonflat_diag);
#pragma omp parallel for
  for (long i = 0; i < (long)M; ++i) {
    const Stats& s = stats[i];
    const V4& nm = normals[i];
    const float sq = dot4(nm, nm);
    if (sq > 0) {
      M4 U = m4_zero();
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) U(a,b) = s.m(a,b);
      M4 D;
      if (s.curvature() < P->point_info_curvature_threshold) D = flatP;
      else { M4 nf = diag4(P->point_nonflat_diag); nf(0,0) = 1.0f / s.eig[0]; nf(1,1) = 1.0f / s.eig[1]; nf(2,2) = 1.0f / s.eig[2]; D = nf; }
      M4 O = m4_mul(m4_mul(U, D), m4_transpose(U));
      info_zero_border(O);
      omegaP[i] = O;
      omegaN[i] = (s.curvature() < P->normal_info_curvature_threshold) ? flatN : nonflatN;
    }
  }
}

/* cloud.cpp:173-186 */
void cloud_transform_in_place(orc_cloud* c, const float Tin[16]) {
  M4 m = m4_load(Tin);
  force_last_row(m);
  const M4 I = m4_identity();
  bool ident = true; for (int k = 0; k < 16; ++k) ident = ident && (m.m[k] == I.m[k]);   /* cloud.cpp:176 */
  if (ident) return;
  const size_t M = c->points.size();
  for (size_t i = 0; i < M; ++i) { V4 p = m4_mul_v4(m, c->points[i]); p.v[3] = 1.0f; c->points[i] = p; }
  for (size_t i = 0; i < c->normals.size(); ++i) { V4 n = m4_mul_v4(m, c->normals[i]); n.v[3] = 0.0f; c->normals[i] = n; }
  for (size_t i = 0; i < c->stats.size(); ++i) c->stats[i].m = m4_mul(m, c->stats[i].m);   /* stats.h:125-131 */
  {                                                                                        /* gaussian3.h:65-73 */
    M3 R; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R(i,j) = m(i,j);
    M3 Rt; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rt(i,j) = R(j,i);
    for (size_t i = 0; i < c->gaussians.size(); ++i) {
      Gauss& g = c->gaussians[i];
      g.updateMoments();
      float mu[3]; m3_mul_v3(R, g.mean, mu);
      for (int k = 0; k < 3; ++k) mu[k] = mu[k] + m(k,3);
      g = Gauss::fromMoments(mu, m3_mul(m3_mul(R, g.cov), Rt));
    }
  }
  M4 T = m; for (int i = 0; i < 4; ++i) { T(3,i) = 0.f; T(i,3) = 0.f; }
  const M4 Tt = m4_transpose(T);
  for (size_t i = 0; i < c->omegaP.size(); ++i) { M4 o = m4_mul(m4_mul(T, c->omegaP[i]), Tt); info_zero_border(o); c->omegaP[i] = o; }   /* informationmatrix.h:111-121 */
  for (size_t i = 0; i < c->omegaN.size(); ++i) { M4 o = m4_mul(m4_mul(T, c->omegaN[i]), Tt); info_zero_border(o); c->omegaN[i] = o; }
}

void project_points(const Projector& pr, int rows, int cols, const V4* points, int n, int* indexImage, float* depthImage) {
  /* pinholepointprojector.cpp:33-66 */
  const size_t N = (size_t)rows * cols;
  for (size_t i = 0; i < N; ++i) { depthImage[i] = FLT_MAX; indexImage[i] = -1; }
  for (int i = 0; i < n; ++i) {
    int x, y; float d;
    if (!pr.project(x, y, d, points[i]) || d < pr.minD || d > pr.maxD || x < 0 || x >= cols || y < 0 || y >= rows) continue;
    float& od = depthImage[(size_t)y * cols + x];
    int& oi = indexImage[(size_t)y * cols + x];
    if (!od || od > d) { od = d; oi = i; }
  }
}

/* acceptance tests of one pixel (correspondencefinder.cpp:60-99); returns 1 and the pair when accepted */
inline bool correspondence_pixel(const orc_aligner_params* P, const orc_cloud* ref, const orc_cloud* cur, const M4& T, int ri, int ci,
                                 float squaredThreshold, float minCurvatureRatio, float maxCurvatureRatio) {
  const V4& cn = cur->normals[ci]; const V4& rn0 = ref->normals[ri];
  const V4& cp = cur->points[ci];  const V4& rp0 = ref->points[ri];
  if (dot4(cn, cn) == 0.0f || dot4(rn0, rn0) == 0.0f) return false;
  V4 rp = iso_mul_v4(T, rp0); rp.v[3] = 1.0f;
  V4 rn = iso_mul_v4(T, rn0); rn.v[3] = 0.0f;
  if (dot4(cn, rn) < P->inlier_normal_angular_threshold) return false;
  V4 dd = { { cp.v[0] - rp.v[0], cp.v[1] - rp.v[1], cp.v[2] - rp.v[2], cp.v[3] - rp.v[3] } };
  if (dot4(dd, dd) > squaredThreshold) return false;
  float rc = ref->stats[ri].curvature(), cc = cur->stats[ci].curvature();
  if (rc < P->flat_curvature_threshold) rc = P->flat_curvature_threshold;
  if (cc < P->flat_curvature_threshold) cc = P->flat_curvature_threshold;
  const float ratio = (float)(((double)rc + 1e-5) / ((double)cc + 1e-5));
  if (ratio < minCurvatureRatio || ratio > maxCurvatureRatio) return false;
  return true;
}
int correspondences(const orc_aligner_params* P, const orc_cloud* ref, const orc_cloud* cur,
                    const int* refIndex, const int* curIndex, M4 T, int* corr, int* Kout) {
  /* correspondencefinder.cpp:20-118.  One thread: the canonical loop.  More threads (the timed CPU baseline only): the reference's
   * own scheme -- every thread takes a contiguous block of rows and fills its own segment, the segments are concatenated in thread
   * order (:38-51,110-114) -- except that the blocks cover ALL rows (the reference drops the trailing rows % numThreads rows), so the
   * list is the one-thread list whatever the thread count. */
  force_last_row(T);
  const float squaredThreshold = P->inlier_distance_threshold * P->inlier_distance_threshold;
  const float minCurvatureRatio = 1.0f / P->inlier_curvature_ratio_threshold;
  const float maxCurvatureRatio = P->inlier_curvature_ratio_threshold;
  const int rows = P->rows, cols = P->cols;
  const int nt = g_parallel_align ? std::max(1, std::min(omp_get_max_threads(), rows)) : 1;
  if (nt == 1) {
    int C = 0, K = 0;
    for (int r = 0; r < rows; ++r)
      for (int c = 0; c < cols; ++c) {
        const int ri = refIndex[(size_t)r * cols + c], ci = curIndex[(size_t)r * cols + c];
        if (ri < 0 || ci < 0) continue;
        ++K;
        if (!correspondence_pixel(P, ref, cur, T, ri, ci, squaredThreshold, minCurvatureRatio, maxCurvatureRatio)) continue;
        corr[2 * C] = ri; corr[2 * C + 1] = ci; ++C;
      }
    if (Kout) *Kout = K;
    return C;
  }
  std::vector<std::vector<int> > seg(nt);
  std::vector<int> Ks(nt, 0);
#pragma omp parallel for num_threads(nt) schedule(static, 1)
  for (int t = 0; t < nt; ++t) {
    const int r0 = (int)((long long)rows * t / nt), r1 = (int)((long long)rows * (t + 1) / nt);
    std::vector<int>& out = seg[t];
    int K = 0;
    for (int r = r0; r < r1; ++r)
      for (int c = 0; c < cols; ++c) {
        const int ri = refIndex[(size_t)r * cols + c], ci = curIndex[(size_t)r * cols + c];
        if (ri < 0 || ci < 0) continue;
        ++K;
        if (!correspondence_pixel(P, ref, cur, T, ri, ci, squaredThreshold, minCurvatureRatio, maxCurvatureRatio)) continue;
        out.push_back(ri); out.push_back(ci);
      }
    Ks[t] = K;
  }
  int C = 0, K = 0;
  for (int t = 0; t < nt; ++t) {
    std::memcpy(corr + 2 * (size_t)C, seg[t].data(), seg[t].size() * sizeof(int));
    C += (int)(seg[t].size() / 2); K += Ks[t];
  }
  if (Kout) *Kout = K;
  return C;
}

template <typename ACC> struct LinAcc {
  ACC Htt[16], Htr[16], Hrr[16], bt[4], br[4], error; double errd; int inliers;
  LinAcc() { for (int k = 0; k < 16; ++k) Htt[k] = Htr[k] = Hrr[k] = 0; for (int k = 0; k < 4; ++k) bt[k] = br[k] = 0; error = 0; errd = 0.0; inliers = 0; }
};
/* the loop body of linearizer.cpp:41-90 over the correspondences [i0, i1) */
template <typename ACC>
void linearize_range(const orc_aligner_params* P, const orc_cloud* ref, const orc_cloud* cur, const int* corr, int i0, int i1, const M4& T, LinAcc<ACC>& a) {
  for (int i = i0; i < i1; ++i) {
    const int ri = corr[2 * i], ci = corr[2 * i + 1];
    V4 rp = iso_mul_v4(T, ref->points[ri]);  rp.v[3] = 1.0f;
    V4 rn = iso_mul_v4(T, ref->normals[ri]); rn.v[3] = 0.0f;
    const V4& cp = cur->points[ci]; const V4& cn = cur->normals[ci];
    const M4& oP = cur->omegaP[ci]; const M4& oN = cur->omegaN[ci];
    V4 pe = { { rp.v[0] - cp.v[0], rp.v[1] - cp.v[1], rp.v[2] - cp.v[2], rp.v[3] - cp.v[3] } };
    V4 ne = { { rn.v[0] - cn.v[0], rn.v[1] - cn.v[1], rn.v[2] - cn.v[2], rn.v[3] - cn.v[3] } };
    const V4 ep = m4_mul_v4(oP, pe), en = m4_mul_v4(oN, ne);
    float localError = dot4(pe, ep) + dot4(ne, en);
    float kscale = 1;
    if (localError > P->inlier_max_chi2) {
      if (P->robust_kernel) kscale = std::sqrt(P->inlier_max_chi2 / localError);
      else continue;
    }
    ++a.inliers;
    const float term = kscale * localError;
    a.error = a.error + (ACC)term; a.errd += (double)term;
    const M4 Sp = skew4(rp), Sn = skew4(rn);
    const M4 oPSp = m4_mul(oP, Sp);
    const M4 A = m4_mul(m4_mul(m4_transpose(Sp), oP), Sp);
    const M4 B = m4_mul(m4_mul(m4_transpose(Sn), oN), Sn);
    const V4 spe = m4_mul_v4(m4_transpose(Sp), ep), sne = m4_mul_v4(m4_transpose(Sn), en);
    for (int k = 0; k < 16; ++k) {
      a.Htt[k] = a.Htt[k] + (ACC)oP.m[k];
      a.Htr[k] = a.Htr[k] + (ACC)oPSp.m[k];
      a.Hrr[k] = a.Hrr[k] + (ACC)(A.m[k] + B.m[k]);        /* Eigen 3.2: Hrr += (tmpA + tmpB) */
    }
    for (int k = 0; k < 4; ++k) {
      a.bt[k] = a.bt[k] + (ACC)(kscale * ep.v[k]);
      a.br[k] = a.br[k] + (ACC)(kscale * (spe.v[k] + sne.v[k]));
    }
  }
}
template <typename ACC>
void linearize_impl(const orc_aligner_params* P, const orc_cloud* ref, const orc_cloud* cur, const int* corr, int C, M4 T,
                    float* Hout, float* bout, float* chi2, double* chi2d, int* inliersOut) {
  /* linearizer.cpp:17-115.  One thread: the canonical serial sums.  More threads (the timed CPU baseline only): the reference's own
   * scheme -- contiguous chunks of the list, per-thread partial sums, reduced serially in thread order (:33-39,93-108) -- with chunks
   * that cover ALL correspondences (the reference drops the last C % numThreads).  Sums then differ from the one-thread ones in the
   * last bits, exactly as the reference's do between thread counts. */
  force_last_row(T);
  const int nt = g_parallel_align ? std::max(1, std::min(omp_get_max_threads(), std::max(1, C / 256))) : 1;
  LinAcc<ACC> tot;
  if (nt == 1) {
    linearize_range<ACC>(P, ref, cur, corr, 0, C, T, tot);
  } else {
    std::vector<LinAcc<ACC> > part(nt);
#pragma omp parallel for num_threads(nt) schedule(static, 1)
    for (int t = 0; t < nt; ++t)
      linearize_range<ACC>(P, ref, cur, corr, (int)((long long)C * t / nt), (int)((long long)C * (t + 1) / nt), T, part[t]);
    for (int t = 0; t < nt; ++t) {
      for (int k = 0; k < 16; ++k) { tot.Htt[k] += part[t].Htt[k]; tot.Htr[k] += part[t].Htr[k]; tot.Hrr[k] += part[t].Hrr[k]; }
      for (int k = 0; k < 4; ++k) { tot.bt[k] += part[t].bt[k]; tot.br[k] += part[t].br[k]; }
      tot.error += part[t].error; tot.errd += part[t].errd; tot.inliers += part[t].inliers;
    }
  }
  ACC* Htt = tot.Htt; ACC* Htr = tot.Htr; ACC* Hrr = tot.Hrr; ACC* bt = tot.bt; ACC* br = tot.br;
  const ACC error = tot.error; const double errd = tot.errd; const int inliers = tot.inliers;
  /* linearizer.cpp:109-114 */
  float H[36];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    H[i + 6 * j] = (float)Htt[i + 4 * j];
    H[i + 6 * (j + 3)] = (float)Htr[i + 4 * j];
    H[(i + 3) + 6 * (j + 3)] = (float)Hrr[i + 4 * j];
  }
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) H[(i + 3) + 6 * j] = H[j + 6 * (i + 3)];
  std::memcpy(Hout, H, sizeof(H));
  for (int i = 0; i < 3; ++i) { bout[i] = (float)bt[i]; bout[i + 3] = (float)br[i]; }
  *chi2 = (float)error; if (chi2d) *chi2d = errd; *inliersOut = inliers;
}

}  // namespace

extern "C" {

void orc_default_converter_params(orc_converter_params* p) {
  std::memset(p, 0, sizeof(*p));
  const float K[9] = { 525.f, 0.f, 0.f, 0.f, 525.f, 0.f, 319.5f, 239.5f, 1.f };   /* pwn_simple_aligner.cpp:226-229 */
  std::memcpy(p->K, K, sizeof(K));
  p->min_distance = 0.01f; p->max_distance = 6.0f;
  p->world_radius = 0.1f; p->min_image_radius = 10; p->max_image_radius = 30; p->min_points = 50;
  p->stats_curvature_threshold = 0.02f;
  p->point_info_curvature_threshold = 0.02f; p->normal_info_curvature_threshold = 0.02f;
  p->point_flat_diag[0] = 1000.f; p->point_flat_diag[1] = 1.f; p->point_flat_diag[2] = 1.f;
  p->point_nonflat_diag[0] = p->point_nonflat_diag[1] = p->point_nonflat_diag[2] = 1.f;
  p->normal_flat_diag[0] = p->normal_flat_diag[1] = p->normal_flat_diag[2] = 100.f;
  p->normal_nonflat_diag[0] = p->normal_nonflat_diag[1] = p->normal_nonflat_diag[2] = 1.f;
  const M4 I = m4_identity(); std::memcpy(p->sensor_offset, I.m, sizeof(I.m));
}
void orc_default_aligner_params(orc_aligner_params* p) {
  std::memset(p, 0, sizeof(*p));
  const float K[9] = { 525.f, 0.f, 0.f, 0.f, 525.f, 0.f, 319.5f, 239.5f, 1.f };
  std::memcpy(p->K, K, sizeof(K));
  p->min_distance = 0.01f; p->max_distance = 6.0f; p->rows = 480; p->cols = 640;
  p->inlier_distance_threshold = 0.5f;
  p->inlier_normal_angular_threshold = (float)std::cos(M_PI / 6);
  p->flat_curvature_threshold = 0.02f; p->inlier_curvature_ratio_threshold = 1.3f;
  p->inlier_max_chi2 = 9e3f; p->robust_kernel = 1; p->outer_iterations = 10; p->inner_iterations = 1;
  const M4 I = m4_identity();
  std::memcpy(p->reference_sensor_offset, I.m, sizeof(I.m));
  std::memcpy(p->current_sensor_offset, I.m, sizeof(I.m));
  std::memcpy(p->initial_guess, I.m, sizeof(I.m));
  p->accumulate_fp64 = 0;
}

void orc_convert_16u_to_32f(const uint16_t* src, float* dst, int n, float scale) {
  for (int i = 0; i < n; ++i) { dst[i] = 0.0f; if (src[i]) dst[i] = scale * src[i]; }
}
void orc_convert_32f_to_16u(const float* src, uint16_t* dst, int n, float scale) {
  for (int i = 0; i < n; ++i) { dst[i] = 0; if (src[i] < FLT_MAX) dst[i] = (uint16_t)(scale * src[i]); }
}
void orc_depth_scale(const float* src, int srows, int scols, int step, float maxDepthCov, float* dst) {
  const int rows = srows / step, cols = scols / step;
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) {
      dst[(size_t)r * cols + c] = 0.f;
      float acc = 0, acc2 = 0; int np = 0;
      const int sr = r * step, sc = c * step;
      for (int i = 0; i < step; ++i)
        for (int j = 0; j < step; ++j)
          if (sr + i < srows && sc + j < scols) {
            const float f = src[(size_t)(sr + i) * scols + sc + j];
            acc += f; acc2 += f * f; np += f > 0;
          }
      if (np) {
        const float mu = acc / np;
        const float sigma = acc2 / np - mu * mu;
        if (sigma > maxDepthCov) continue;
        dst[(size_t)r * cols + c] = mu;
      }
    }
}

void orc_projector_matrices(const float K[9], const float T[16], float KRt[16], float iKRt[16], float iK[9]) {
  Projector pr = make_projector(K, T, 0.f, 0.f);
  if (KRt) std::memcpy(KRt, pr.KRt.m, sizeof(pr.KRt.m));
  if (iKRt) std::memcpy(iKRt, pr.iKRt.m, sizeof(pr.iKRt.m));
  if (iK) std::memcpy(iK, pr.iK.m, sizeof(pr.iK.m));
}

orc_cloud* orc_cloud_create(void) { return new orc_cloud(); }
void orc_cloud_destroy(orc_cloud* c) { delete c; }
int orc_cloud_size(const orc_cloud* c) { return (int)c->points.size(); }
void orc_cloud_get(const orc_cloud* c, float* points, float* normals, float* curvature, float* stats,
                   float* eigenvalues, int* npoints, float* omega_p, float* omega_n) {
  const size_t M = c->points.size();
  for (size_t i = 0; i < M; ++i) {
    if (points) std::memcpy(points + 4 * i, c->points[i].v, 16);
    if (normals) std::memcpy(normals + 4 * i, c->normals[i].v, 16);
    if (curvature) curvature[i] = c->stats[i].curvature();
    if (stats) std::memcpy(stats + 16 * i, c->stats[i].m.m, 64);
    if (eigenvalues) std::memcpy(eigenvalues + 3 * i, c->stats[i].eig, 12);
    if (npoints) npoints[i] = c->stats[i].n;
    /* a loaded cloud has no information matrices (cloud.cpp:25-82 fills points, normals and stats only): report zeros */
    if (omega_p) { if (i < c->omegaP.size()) std::memcpy(omega_p + 16 * i, c->omegaP[i].m, 64); else std::memset(omega_p + 16 * i, 0, 64); }
    if (omega_n) { if (i < c->omegaN.size()) std::memcpy(omega_n + 16 * i, c->omegaN[i].m, 64); else std::memset(omega_n + 16 * i, 0, 64); }
  }
}
void orc_cloud_set(orc_cloud* c, int n, const float* points, const float* normals, const float* curvature,
                   const float* omega_p, const float* omega_n) {
  c->points.resize(n); c->normals.resize(n); c->stats.assign(n, Stats()); c->omegaP.resize(n); c->omegaN.resize(n);
  for (int i = 0; i < n; ++i) {
    std::memcpy(c->points[i].v, points + 4 * i, 16);
    std::memcpy(c->normals[i].v, normals + 4 * i, 16);
    c->stats[i].curv = curvature[i]; c->stats[i].curvatureComputed = true;   /* stats.h:105-108 setCurvature */
    std::memcpy(c->omegaP[i].m, omega_p + 16 * i, 64);
    std::memcpy(c->omegaN[i].m, omega_n + 16 * i, 64);
  }
}

/* The Gaussian part of PinholePointProjector::unProject(points, gaussians, index, depth) (pinholepointprojector.cpp:104-123,
 * baseline / alpha defaults :10-11): per valid pixel the sensor-noise covariance (iK J) diag(3, 3, zVariation) (iK J)^T around the point. */
static int g_with_gaussians = 0;
static float g_baseline = 0.075f, g_alpha = 0.1f;
static void unproject_gaussians(const orc_converter_params* p, const float* depth, int rows, int cols, const std::vector<V4>& points,
                                std::vector<Gauss>& gaussians) {
  const M4 I = m4_identity();
  Projector pr = make_projector(p->K, I.m, p->min_distance, p->max_distance);
  gaussians.resize(points.size());
  const float fB = g_baseline * pr.K(0,0);
  size_t count = 0;
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) {
      const float z = depth[(size_t)r * cols + c];
      if (z < pr.minD || z > pr.maxD) continue;
      const float zVariation = (g_alpha * z * z) / (fB + z * g_alpha);
      M3 J; J(0,0) = z; J(0,1) = 0.f; J(0,2) = (float)c; J(1,0) = 0.f; J(1,1) = z; J(1,2) = (float)r; J(2,0) = 0.f; J(2,1) = 0.f; J(2,2) = 1.f;
      J = m3_mul(pr.iK, J);
      const float d[3] = { 3.0f, 3.0f, zVariation };
      M3 JD; for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) JD(i,k) = J(i,k) * d[k];
      M3 Jt; for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) Jt(i,k) = J(k,i);
      gaussians[count] = Gauss::fromMoments(points[count].v, m3_mul(JD, Jt));
      ++count;
    }
}

int orc_unproject(const orc_converter_params* p, const float* depth, int rows, int cols, float* points, int* index_image) {
  const M4 I = m4_identity();
  Projector pr = make_projector(p->K, I.m, p->min_distance, p->max_distance);
  int count = 0;
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) {
      V4 pt;
      if (!pr.unProject(pt, c, r, depth[(size_t)r * cols + c])) { index_image[(size_t)r * cols + c] = -1; continue; }
      std::memcpy(points + 4 * (size_t)count, pt.v, 16);
      index_image[(size_t)r * cols + c] = count++;
    }
  return count;
}
void orc_project_intervals(const orc_converter_params* p, const float* depth, int rows, int cols, int* interval_image) {
  const M4 I = m4_identity();
  Projector pr = make_projector(p->K, I.m, p->min_distance, p->max_distance);
  for (size_t i = 0; i < (size_t)rows * cols; ++i) interval_image[i] = pr.projectInterval(depth[i], p->world_radius);
}
void orc_integral_image(const int* index_image, const float* points, int rows, int cols, float* out) {
  std::vector<Acc> I;
  integral_image(index_image, reinterpret_cast<const V4*>(points), rows, cols, I);
  const size_t N = (size_t)rows * cols;
  for (int k = 0; k < CH; ++k) for (size_t i = 0; i < N; ++i) out[k * N + i] = I[i].c[k];
}

void orc_convert(const orc_converter_params* p, const float* depth, int rows, int cols, orc_cloud* cloud,
                 int* index_image, int* interval_image) {
  /* depthimageconverterintegralimage.cpp:15-55 */
  const size_t N = (size_t)rows * cols;
  std::vector<int> idx(N), itv(N);
  cloud->points.resize(N);
  const int M = orc_unproject(p, depth, rows, cols, reinterpret_cast<float*>(cloud->points.data()), idx.data());
  cloud->points.resize(M);
  cloud->gaussians.clear();
  if (g_with_gaussians) unproject_gaussians(p, depth, rows, cols, cloud->points, cloud->gaussians);
  orc_project_intervals(p, depth, rows, cols, itv.data());
  stats_compute(p, idx.data(), itv.data(), rows, cols, cloud->points, cloud->normals, cloud->stats);
  info_compute(p, cloud->stats, cloud->normals, cloud->omegaP, cloud->omegaN);
  cloud_transform_in_place(cloud, p->sensor_offset);
  if (index_image) std::memcpy(index_image, idx.data(), N * sizeof(int));
  if (interval_image) std::memcpy(interval_image, itv.data(), N * sizeof(int));
}

void orc_project(const float K[9], const float T[16], float min_distance, float max_distance, int rows, int cols,
                 const float* points, int n, int* index_image, float* depth_image) {
  Projector pr = make_projector(K, T, min_distance, max_distance);
  project_points(pr, rows, cols, reinterpret_cast<const V4*>(points), n, index_image, depth_image);
}

int orc_correspondences(const orc_aligner_params* p, const orc_cloud* ref, const orc_cloud* cur, const int* ref_index,
                        const int* cur_index, const float T[16], int* corr, int* K_out) {
  return correspondences(p, ref, cur, ref_index, cur_index, m4_load(T), corr, K_out);
}
void orc_linearize(const orc_aligner_params* p, const orc_cloud* ref, const orc_cloud* cur, const int* corr, int C,
                   const float T[16], float* H, float* b, float* chi2, double* chi2_fp64, int* inliers) {
  if (p->accumulate_fp64) linearize_impl<double>(p, ref, cur, corr, C, m4_load(T), H, b, chi2, chi2_fp64, inliers);
  else                    linearize_impl<float>(p, ref, cur, corr, C, m4_load(T), H, b, chi2, chi2_fp64, inliers);
}

/* --------------------------------------------- se3_prior.{h,cpp} + aligner.cpp:96-108 -------------------------------- */
namespace {
bool inverse_n(int n, const float* Ain, float* out);      /* defined with the statistics helpers below */
struct Prior { int kind; M4 mean; M4 invRef; float info[36]; };
std::vector<Prior> g_priors;      /* priors of the next orc_align calls (Aligner::addRelativePrior / addAbsolutePrior) */
void prior_error(const Prior& pr, const M4& mean, const M4& invT, float e[6]) {                 /* se3_prior.cpp:58-60, 69-71 */
  const M4 X = pr.kind == 0 ? iso_mul(invT, mean) : iso_mul(iso_mul(invT, pr.invRef), mean);
  t2v(X, e);
}
void m6_mul(const float* A, const float* B, float* R) {
  for (int j = 0; j < 6; ++j) for (int i = 0; i < 6; ++i) { float s = A[i] * B[6 * j]; for (int k = 1; k < 6; ++k) s = s + A[i + 6 * k] * B[k + 6 * j]; R[i + 6 * j] = s; }
}
void m6_t(const float* A, float* R) { for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) R[i + 6 * j] = A[j + 6 * i]; }
void prior_add(const Prior& pr, const M4& invT, float H[36], float b[6]) {
  const float epsilon = 1e-3f, iEpsilon = 0.5f / epsilon;                                       /* se3_prior.cpp:10-11 */
  float e[6]; prior_error(pr, pr.mean, invT, e);
  float J[36], Jz[36];
  for (int i = 0; i < 6; ++i) {
    float up[6] = {0,0,0,0,0,0}, dn[6] = {0,0,0,0,0,0}, eu[6], ed[6];
    up[i] = epsilon; dn[i] = -epsilon;
    prior_error(pr, pr.mean, iso_mul(v2t(up), invT), eu); prior_error(pr, pr.mean, iso_mul(v2t(dn), invT), ed);          /* :18 */
    for (int r = 0; r < 6; ++r) J[r + 6 * i] = iEpsilon * (eu[r] - ed[r]);
    prior_error(pr, iso_mul(pr.mean, v2t(up)), invT, eu); prior_error(pr, iso_mul(pr.mean, v2t(dn)), invT, ed);          /* :34-40 */
    for (int r = 0; r < 6; ++r) Jz[r + 6 * i] = iEpsilon * (eu[r] - ed[r]);
  }
  float iJz[36], iJzT[36], t1[36], Om[36], Jt[36], t2[36], Hp[36];
  if (!inverse_n(6, Jz, iJz)) return;                                                           /* :50 */
  m6_t(iJz, iJzT); m6_mul(iJzT, pr.info, t1); m6_mul(t1, iJz, Om);                              /* :51 */
  m6_t(J, Jt); m6_mul(Jt, Om, t2); m6_mul(t2, J, Hp);                                           /* aligner.cpp:103 */
  for (int i = 0; i < 36; ++i) H[i] = H[i] + Hp[i];                                             /* :106 */
  for (int i = 0; i < 6; ++i) { float sacc = t2[i] * e[0]; for (int k = 1; k < 6; ++k) sacc = sacc + t2[i + 6 * k] * e[k]; b[i] = b[i] + sacc; }   /* :104,107 */
}
}  // namespace
void orc_clear_priors(void) { g_priors.clear(); }
void orc_add_prior(int kind, const float mean[16], const float reference_transform[16], const float information[36]) {
  Prior p; p.kind = kind; p.mean = m4_load(mean); p.invRef = kind == 1 ? iso_inverse(m4_load(reference_transform)) : m4_identity();
  std::memcpy(p.info, information, sizeof(p.info)); g_priors.push_back(p);
}

static std::vector<int> g_last_corr;     /* correspondences of the last outer iteration of the last orc_align (single-threaded test use) */
static int g_last_C_tmp = 0;
/* Aligner::_computeStatistics after orc_align: the 11th Linearizer::update at the final transform on the finder's last
 * correspondences (aligner.cpp:165-170), then orc_compute_statistics. */
void orc_align_statistics(const orc_aligner_params* p, const orc_cloud* ref, const orc_cloud* cur, const float T[16], float H_out[36],
                          float mean[6], float omega[36], float* translational_ratio, float* rotational_ratio) {
  M4 invT = iso_inverse(m4_load(T)); force_last_row(invT);
  float H[36], b[6], chi2; double chi2d; int inl;
  orc_linearize(p, ref, cur, g_last_corr.data(), (int)(g_last_corr.size() / 2), invT.m, H, b, &chi2, &chi2d, &inl);
  if (H_out) std::memcpy(H_out, H, sizeof(H));
  orc_compute_statistics(H, T, mean, omega, translational_ratio, rotational_ratio);
}
void orc_align(const orc_aligner_params* p, const orc_cloud* ref, const orc_cloud* cur, float T_out[16], float* error_out,
               int* inliers_out, orc_iter_trace* trace, int* ref_index_out, float* ref_depth_out, int* cur_index_out,
               float* cur_depth_out) {
  /* aligner.cpp:49-125 */
  const size_t N = (size_t)p->rows * p->cols;
  std::vector<int> refIdx(N), curIdx(N), corr(2 * N);
  std::vector<float> refDepth(N), curDepth(N);
  Projector pr = make_projector(p->K, p->current_sensor_offset, p->min_distance, p->max_distance);
  project_points(pr, p->rows, p->cols, cur->points.data(), (int)cur->points.size(), curIdx.data(), curDepth.data());
  M4 T = m4_load(p->initial_guess);
  const M4 refOff = m4_load(p->reference_sensor_offset);
  float err = 0.f; int inl = 0; int it = 0;
  for (int i = 0; i < p->outer_iterations; ++i) {
    force_last_row(T);
    pr.T = iso_mul(T, refOff); pr.update();
    project_points(pr, p->rows, p->cols, ref->points.data(), (int)ref->points.size(), refIdx.data(), refDepth.data());
    int K = 0;
    const int C = correspondences(p, ref, cur, refIdx.data(), curIdx.data(), iso_inverse(T), corr.data(), &K);
    g_last_C_tmp = C;
    M4 invT = iso_inverse(T);
    for (int k = 0; k < p->inner_iterations; ++k, ++it) {
      force_last_row(invT);
      float H[36], b[6]; float chi2; double chi2d;
      orc_linearize(p, ref, cur, corr.data(), C, invT.m, H, b, &chi2, &chi2d, &inl);
      err = chi2;
      if (trace) {
        orc_iter_trace& t = trace[it];
        t.K = K; t.C = C; t.inliers = inl; t.chi2 = chi2; t.chi2_fp64 = chi2d;
        std::memcpy(t.H, H, sizeof(H)); std::memcpy(t.b, b, sizeof(b)); std::memcpy(t.T_before, T.m, sizeof(T.m));
      }
      for (int d = 0; d < 6; ++d) { H[d + 6 * d] = H[d + 6 * d] + 1.0f; }        /* aligner.cpp:92 */
      for (int d = 0; d < 6; ++d) { H[d + 6 * d] = H[d + 6 * d] + 1000.0f; }     /* aligner.cpp:94 */
      for (const Prior& pr : g_priors) prior_add(pr, invT, H, b);                /* aligner.cpp:97-108 */
      float nb[6], dx[6];
      for (int d = 0; d < 6; ++d) nb[d] = -b[d];
      ldlt_solve6(H, nb, dx);
      const M4 dT = v2t(dx);
      invT = iso_mul(dT, invT);
    }
    T = iso_inverse(invT);
    float v[6]; t2v(T, v); T = v2t(v);
    force_last_row(T);
  }
  g_last_corr.assign(corr.begin(), corr.begin() + 2 * (size_t)g_last_C_tmp);
  std::memcpy(T_out, T.m, sizeof(T.m));
  if (error_out) *error_out = err;
  if (inliers_out) *inliers_out = inl;
  if (ref_index_out) std::memcpy(ref_index_out, refIdx.data(), N * sizeof(int));
  if (ref_depth_out) std::memcpy(ref_depth_out, refDepth.data(), N * sizeof(float));
  if (cur_index_out) std::memcpy(cur_index_out, curIdx.data(), N * sizeof(int));
  if (cur_depth_out) std::memcpy(cur_depth_out, curDepth.data(), N * sizeof(float));
}

/* pwn_tracker/pwn_matcher_base.cpp:153-182.  cv::Mat semantics restated: (a>0)&(b>0) is a 0/255 uchar mask, converted
 * to float it is 0.0f/255.0f, and `abs(cur-ref) & mask` on CV_32F data is a BITWISE and of the float words. */
void orc_match_score(const float* ref_depth, const float* cur_depth, int n, float threshold, int* non_zeros, int* outliers,
                     int* inliers, float* reprojection_distance) {
  std::vector<uint16_t> c(n), r(n);
  orc_convert_32f_to_16u(cur_depth, c.data(), n, 1000.0f);     /* :157 */
  orc_convert_32f_to_16u(ref_depth, r.data(), n, 1000.0f);     /* :160 */
  int nz = 0, inl = 0; float sum = 0;
  for (int i = 0; i < n; ++i) {
    const bool m = c[i] > 0 && r[i] > 0;                        /* :163 */
    const float maskf = m ? 255.0f : 0.0f;                      /* :166 */
    const float ad = std::fabs((float)c[i] - (float)r[i]);      /* :164-165,167 */
    uint32_t a, b; std::memcpy(&a, &ad, 4); std::memcpy(&b, &maskf, 4);
    const uint32_t w = a & b; float d; std::memcpy(&d, &w, 4);  /* :167 bitwise and */
    nz += m;                                                     /* :168 countNonZero(mask) */
    if (maskf && d < threshold) ++inl;                          /* :174-175 */
    sum += d;                                                    /* :176 */
  }
  *non_zeros = nz; *inliers = inl; *outliers = nz - inl;        /* :180-182 */
  *reprojection_distance = sum / nz;                            /* :179 */
}

/* ------------------------------------------------ Aligner::_computeStatistics (aligner.cpp:152-199) ------------ */
namespace {
/* Symmetric eigen-decomposition by cyclic Jacobi rotations (float).  Stands in for Eigen's JacobiSVD on the symmetric
 * positive (semi)definite matrices of this function: singular values = eigenvalues, U = V = eigenvectors. */
void jacobi_sym(int n, std::vector<float>& A, std::vector<float>& V, std::vector<float>& ev) {
  V.assign(n * n, 0.f); for (int i = 0; i < n; ++i) V[i + n * i] = 1.f;
  for (int sweep = 0; sweep < 30; ++sweep) {
    float off = 0.f;
    for (int p = 0; p < n; ++p) for (int q = p + 1; q < n; ++q) off += A[p + n * q] * A[p + n * q];
    if (off < 1e-30f) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        const float apq = A[p + n * q];
        if (std::fabs(apq) < 1e-30f) continue;
        const float theta = (A[q + n * q] - A[p + n * p]) / (2.f * apq);
        const float t = (theta >= 0.f ? 1.f : -1.f) / (std::fabs(theta) + std::sqrt(theta * theta + 1.f));
        const float c = 1.f / std::sqrt(t * t + 1.f), sn = t * c;
        for (int k = 0; k < n; ++k) { const float akp = A[k + n * p], akq = A[k + n * q]; A[k + n * p] = c * akp - sn * akq; A[k + n * q] = sn * akp + c * akq; }
        for (int k = 0; k < n; ++k) { const float apk = A[p + n * k], aqk = A[q + n * k]; A[p + n * k] = c * apk - sn * aqk; A[q + n * k] = sn * apk + c * aqk; }
        for (int k = 0; k < n; ++k) { const float vkp = V[k + n * p], vkq = V[k + n * q]; V[k + n * p] = c * vkp - sn * vkq; V[k + n * q] = sn * vkp + c * vkq; }
      }
  }
  ev.resize(n); for (int i = 0; i < n; ++i) ev[i] = A[i + n * i];
}
/* general inverse by Gauss-Jordan with partial pivoting (Eigen Matrix6f::inverse() = PartialPivLU) */
bool inverse_n(int n, const float* Ain, float* out) {
  std::vector<float> a(Ain, Ain + n * n), b(n * n, 0.f);
  for (int i = 0; i < n; ++i) b[i + n * i] = 1.f;
  for (int c = 0; c < n; ++c) {
    int piv = c; float best = std::fabs(a[c + n * c]);
    for (int r = c + 1; r < n; ++r) if (std::fabs(a[r + n * c]) > best) { best = std::fabs(a[r + n * c]); piv = r; }
    if (best == 0.f) return false;
    if (piv != c) for (int k = 0; k < n; ++k) { std::swap(a[c + n * k], a[piv + n * k]); std::swap(b[c + n * k], b[piv + n * k]); }
    const float d = a[c + n * c];
    for (int k = 0; k < n; ++k) { a[c + n * k] /= d; b[c + n * k] /= d; }
    for (int r = 0; r < n; ++r) if (r != c) {
      const float f = a[r + n * c];
      if (f != 0.f) for (int k = 0; k < n; ++k) { a[r + n * k] -= f * a[c + n * k]; b[r + n * k] -= f * b[c + n * k]; }
    }
  }
  std::memcpy(out, b.data(), sizeof(float) * n * n);
  return true;
}
}  // namespace

/* H: the linearizer's 6x6 at the final transform (no damping); T: Aligner::_T.  Outputs as aligner.cpp:152-199. */
void orc_compute_statistics(const float Hin[36], const float T[16], float mean[6], float omega[36], float* translational_ratio, float* rotational_ratio) {
  const int n = 6;
  std::vector<float> H(Hin, Hin + 36), V, ev;
  for (int i = 0; i < n; ++i) H[i + n * i] += 1.0f;                                   /* :169 H += linearizer.H + I */
  std::vector<float> A = H;
  jacobi_sym(n, A, V, ev);                                                            /* :172 JacobiSVD */
  float smax = 0.f; for (float v : ev) smax = std::max(smax, std::fabs(v));
  std::vector<float> sigma(36, 0.f);                                                  /* :173 svd.solve(I): pseudo-inverse */
  for (int k = 0; k < n; ++k) {
    if (std::fabs(ev[k]) <= FLT_EPSILON * n * smax) continue;
    const float inv = 1.0f / ev[k];
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) sigma[i + n * j] += V[i + n * k] * inv * V[j + n * k];
  }
  /* unscented.h:23-50 sampleUnscented(mean = 0, covariance = sigma) */
  const double alpha = 1e-3, beta = 2., lambda = alpha * alpha * n;
  const double wi = 1. / (2. * (n + lambda));
  std::vector<float> L(36, 0.f), C(36);
  for (int k = 0; k < 36; ++k) C[k] = sigma[k] * (float)(n + lambda);                 /* covariance*(dim+lambda), float matrix */
  for (int j = 0; j < n; ++j) {                                                       /* LLT */
    float d = C[j + n * j];
    for (int k = 0; k < j; ++k) d -= L[j + n * k] * L[j + n * k];
    d = std::sqrt(d);
    L[j + n * j] = d;
    for (int i = j + 1; i < n; ++i) { float v = C[i + n * j]; for (int k = 0; k < j; ++k) v -= L[i + n * k] * L[j + n * k]; L[i + n * j] = v / d; }
  }
  struct SP { float s[6]; double wi, wp; };
  std::vector<SP> sp(2 * n + 1);
  for (int k = 0; k < 6; ++k) sp[0].s[k] = 0.f;
  sp[0].wi = lambda / (n + lambda); sp[0].wp = lambda / (n + lambda) + (1. - alpha * alpha + beta);
  for (int i = 0, k = 1; i < n; ++i) {
    for (int r = 0; r < n; ++r) { sp[k].s[r] = L[r + n * i]; sp[k + 1].s[r] = -L[r + n * i]; }
    sp[k].wi = sp[k].wp = wi; sp[k + 1].wi = sp[k + 1].wp = wi; k += 2;
  }
  /* :178-185 remap: p = t2v(dT * v2t(p).inverse()) */
  const M4 dT = m4_load(T);
  for (auto& p : sp) { const M4 X = iso_mul(dT, iso_inverse(v2t(p.s))); t2v(X, p.s); }
  /* unscented.h:52-65 reconstructGaussian */
  float m[6] = {0, 0, 0, 0, 0, 0}; std::vector<float> cov(36, 0.f);
  for (auto& p : sp) for (int r = 0; r < n; ++r) m[r] += (float)p.wi * p.s[r];
  for (auto& p : sp) { float dlt[6]; for (int r = 0; r < n; ++r) dlt[r] = p.s[r] - m[r];
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) cov[i + n * j] += (float)p.wp * (dlt[i] * dlt[j]); }
  std::memcpy(mean, m, sizeof(m));
  if (!inverse_n(n, cov.data(), omega)) for (int k = 0; k < 36; ++k) omega[k] = 0.f;  /* :190 */
  /* :193-198 singular values of the two 3x3 diagonal blocks */
  for (int blk = 0; blk < 2; ++blk) {
    std::vector<float> B(9), Vb, eb;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) B[i + 3 * j] = omega[(i + 3 * blk) + n * (j + 3 * blk)];
    /* singular values of a general 3x3 = sqrt(eig(B^T B)) */
    std::vector<float> BtB(9, 0.f);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) BtB[i + 3 * j] += B[k + 3 * i] * B[k + 3 * j];
    jacobi_sym(3, BtB, Vb, eb);
    float s0 = 0.f, s2 = FLT_MAX;
    for (float v : eb) { const float sv = std::sqrt(std::max(v, 0.f)); s0 = std::max(s0, sv); s2 = std::min(s2, sv); }
    (blk == 0 ? *translational_ratio : *rotational_ratio) = s0 / s2;
  }
}

void orc_iso_inverse(const float T[16], float out[16]) { const M4 r = iso_inverse(m4_load(T)); std::memcpy(out, r.m, sizeof(r.m)); }
void orc_iso_mul(const float A[16], const float B[16], float out[16]) { const M4 r = iso_mul(m4_load(A), m4_load(B)); std::memcpy(out, r.m, sizeof(r.m)); }
/* PwnTracker::processFrame's periodic clean-up of the accumulated rotation (pwn_tracker/pwn_tracker.cpp:154-159):
 *   R = globalT.linear(); E = R^T R; E.diagonal() -= 1; globalT.linear() -= 0.5 * R * E
 * (0.5 * R is exact in fp32, so (0.5 R) E and 0.5 (R E) are the same bits whichever way Eigen groups the expression). */
void orc_reorthonormalize(const float T[16], float out[16]) {
  M4 t = m4_load(T);
  const M3 R = iso_linear(t);
  M3 Rt; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rt(i,j) = R(j,i);
  M3 E = m3_mul(Rt, R);
  E(0,0) -= 1.0f; E(1,1) -= 1.0f; E(2,2) -= 1.0f;
  M3 hR; for (int k = 0; k < 9; ++k) hR.m[k] = 0.5f * R.m[k];
  const M3 D = m3_mul(hR, E);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) t(i,j) = R(i,j) - D(i,j);
  std::memcpy(out, t.m, sizeof(t.m));
}
void orc_v2t(const float v[6], float T[16]) { const M4 t = v2t(v); std::memcpy(T, t.m, sizeof(t.m)); }
void orc_t2v(const float T[16], float v[6]) { t2v(m4_load(T), v); }
void orc_eigen3(const float A[9], float evals[3], float evecs[9]) {
  M3 a; std::memcpy(a.m, A, sizeof(a.m)); M3 U; eig3_direct(a, evals, U); std::memcpy(evecs, U.m, sizeof(U.m));
}
void orc_ldlt_solve6(const float H[36], const float b[6], float x[6]) { ldlt_solve6(H, b, x); }
void orc_set_trig_mode(int mode) { g_trig_mode = (mode == 1 || mode == 2) ? mode : 0; }
/* the eigensolver's three trig values for (y = sqrt(q), x = half_b) in the given mode: theta = atan2(y, x) / 3 (float), cos, sin */
void orc_trig_eval(int mode, int n, const float* y, const float* x, float* theta, float* c, float* s) {
  const int saved = g_trig_mode; g_trig_mode = (mode == 1 || mode == 2) ? mode : 0;
  const float s_inv3 = 1.0f / 3.0f;
  for (int i = 0; i < n; ++i) { theta[i] = trig_atan2(y[i], x[i]) * s_inv3; c[i] = trig_cos(theta[i]); s[i] = trig_sin(theta[i]); }
  g_trig_mode = saved;
}
void orc_set_parallel_align(int enabled) { g_parallel_align = enabled ? 1 : 0; }
void orc_set_num_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n > 0 ? n : 1);
#else
  (void)n;
#endif
}
int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

}  // extern "C"

/* =============================================================================================================
 * Scene maintenance: Cloud::add, Merger::merge, VoxelCalculator::compute, Cloud::save/load
 * ============================================================================================================= */
#include <fstream>
#include <sstream>
#include <map>
#include <string>

extern "C" {

/* convert() also fills the cloud's Gaussians (the reference always does: depthimageconverterintegralimage.cpp:39; here it is a
 * switch so that the timed CPU baseline measures the same work as the GPU path, which produces them on request only) */
void orc_set_gaussians(int enabled, float baseline, float alpha) { g_with_gaussians = enabled ? 1 : 0; g_baseline = baseline; g_alpha = alpha; }
int orc_cloud_num_gaussians(const orc_cloud* c) { return (int)c->gaussians.size(); }
/* per Gaussian: mean[3] cov[9] (column-major) infoVec[3] info[9] flags (1 = moments valid, 2 = information valid); any pointer may be NULL */
void orc_cloud_get_gaussians(const orc_cloud* c, float* mean, float* cov, float* info_vec, float* info, int* flags) {
  for (size_t i = 0; i < c->gaussians.size(); ++i) {
    const Gauss& g = c->gaussians[i];
    if (mean) std::memcpy(mean + 3 * i, g.mean, 12);
    if (cov) std::memcpy(cov + 9 * i, g.cov.m, 36);
    if (info_vec) std::memcpy(info_vec + 3 * i, g.infoVec, 12);
    if (info) std::memcpy(info + 9 * i, g.info.m, 36);
    if (flags) flags[i] = (g.momentsUpdated ? 1 : 0) | (g.infoUpdated ? 2 : 0);
  }
}
void orc_cloud_transform_in_place(orc_cloud* c, const float T[16]) { cloud_transform_in_place(c, T); }

/* Cloud::add (cloud.cpp:145-171): append a transformed copy */
void orc_cloud_add(orc_cloud* dst, const orc_cloud* src, const float T[16]) {
  orc_cloud tmp = *src;
  cloud_transform_in_place(&tmp, T);
  const size_t k = dst->points.size();
  dst->points.resize(k + tmp.points.size()); dst->normals.resize(k + tmp.normals.size()); dst->stats.resize(k + tmp.stats.size());
  dst->omegaP.resize(k + tmp.omegaP.size()); dst->omegaN.resize(k + tmp.omegaN.size());
  dst->gaussians.resize(k + tmp.gaussians.size());
  for (size_t i = 0; i < tmp.points.size(); ++i) {
    dst->points[k + i] = tmp.points[i]; dst->normals[k + i] = tmp.normals[i]; dst->stats[k + i] = tmp.stats[i];
    if (!tmp.omegaP.empty()) { dst->omegaP[k + i] = tmp.omegaP[i]; dst->omegaN[k + i] = tmp.omegaN[i]; }
    if (!tmp.gaussians.empty()) dst->gaussians[k + i] = tmp.gaussians[i];
  }
}

/* Merger::merge (merger.cpp:15-119).  K, min/max distance: the converter's projector; T: the projector pose.  collapsed (optional,
 * size of the cloud before the merge) receives _collapsedIndices.  Returns the new size.  As in the reference the Gaussian vector is
 * compacted in place but NOT resized (merger.cpp:108-112): its tail keeps stale entries. */
int orc_merge(orc_cloud* c, const float K[9], const float T[16], float min_distance, float max_distance, int rows, int cols,
              float distance_threshold, float normal_threshold, float max_point_depth, int* collapsed_out) {
  Projector pr = make_projector(K, T, min_distance, max_distance);
  const int n = (int)c->points.size();
  std::vector<int> indexImage((size_t)rows * cols); std::vector<float> depthImage((size_t)rows * cols);
  project_points(pr, rows, cols, c->points.data(), n, indexImage.data(), depthImage.data());
  std::vector<int> collapsed(n, -1);
  for (int i = 0; i < n; ++i) {
    const V4 currentPoint = c->points[i];
    const V4 currentNormal = c->normals[i];
    int r = -1, cc = -1; float depth = 0.0f;
    pr.project(cc, r, depth, currentPoint);
    if (depth < 0 || depth > max_point_depth || r < 0 || r >= rows || cc < 0 || cc >= cols) continue;
    const float targetZ = depthImage[(size_t)r * cols + cc];
    const int targetIndex = indexImage[(size_t)r * cols + cc];
    if (targetIndex < 0) continue;
    const V4& targetNormal = c->normals[targetIndex];
    if (targetIndex == i) collapsed[i] = i;
    else if (std::fabs(depth - targetZ) < distance_threshold && dot4(currentNormal, targetNormal) > normal_threshold) {
      c->gaussians[targetIndex].addInformation(c->gaussians[i]);
      collapsed[i] = targetIndex;
    }
  }
  int k = 0;
  for (int i = 0; i < n; ++i) {
    const int ci = collapsed[i];
    if (ci == i) { Gauss& g = c->gaussians[i]; g.updateMoments(); for (int a = 0; a < 3; ++a) c->points[i].v[a] = g.mean[a]; }
    if (ci < 0 || ci == i) {
      c->points[k] = c->points[i]; c->normals[k] = c->normals[i]; c->stats[k] = c->stats[i];
      c->omegaP[k] = c->omegaP[i]; c->omegaN[k] = c->omegaN[i]; c->gaussians[k] = c->gaussians[i];
      ++k;
    }
  }
  c->points.resize(k); c->normals.resize(k); c->stats.resize(k); c->omegaP.resize(k); c->omegaN.resize(k);
  if (collapsed_out) std::memcpy(collapsed_out, collapsed.data(), sizeof(int) * (size_t)n);
  return k;
}

/* VoxelCalculator::compute (voxelcalculator.cpp:15-73): keeps the first point (lowest index) that falls into each voxel, in the
 * iteration order of a std::map keyed by the voxel indices.
 * literal = 1: std::map with the reference's IndexComparator (voxelcalculator.h:41-48), which is NOT a strict weak ordering (its third
 *   clause lacks indeces[0] == s.indeces[0]); what the map then does is a property of libstdc++'s red-black tree, reproduced here by
 *   using that very container.  It can fail to find an existing voxel, so more than one point per voxel can survive.
 * literal = 0 (canonical, what the GPU path implements): the intended lexicographic order (one point per voxel).
 * kept (optional) receives the original indices of the surviving points, in output order.  Returns the new size. */
}  // extern "C"
struct VoxelKeyLiteral { int i[3]; bool operator<(const VoxelKeyLiteral& s) const {
  if (i[0] < s.i[0]) return true;
  if (i[0] == s.i[0] && i[1] < s.i[1]) return true;
  if (i[1] == s.i[1] && i[2] < s.i[2]) return true;
  return false; } };
struct VoxelKeyCanonical { int i[3]; bool operator<(const VoxelKeyCanonical& s) const {
  if (i[0] != s.i[0]) return i[0] < s.i[0];
  if (i[1] != s.i[1]) return i[1] < s.i[1];
  return i[2] < s.i[2]; } };
template <typename KEY> static void voxel_survivors(const orc_cloud* c, float res, std::vector<int>& kept) {
  std::map<KEY, int> acc;
  const float inverseResolution = 1.0f / res;
  for (size_t i = 0; i < c->points.size(); ++i) {
    KEY s; for (int a = 0; a < 3; ++a) s.i[a] = (int)(c->points[i].v[a] * inverseResolution);
    if (acc.find(s) == acc.end()) acc.insert(std::make_pair(s, (int)i));
  }
  kept.clear();
  for (auto it = acc.begin(); it != acc.end(); ++it) kept.push_back(it->second);
}
extern "C" {
int orc_voxelize(orc_cloud* c, float resolution, int literal, int* kept_out) {
  std::vector<int> kept;
  if (literal) voxel_survivors<VoxelKeyLiteral>(c, resolution, kept); else voxel_survivors<VoxelKeyCanonical>(c, resolution, kept);
  orc_cloud t;
  const bool info = c->omegaP.size() == c->points.size() && c->omegaN.size() == c->points.size();     /* voxelcalculator.cpp:54-58 */
  const bool gauss = c->gaussians.size() == c->points.size();                                         /* :62-64 */
  for (int idx : kept) {
    t.points.push_back(c->points[idx]); t.normals.push_back(c->normals[idx]); t.stats.push_back(c->stats[idx]);
    if (info) { t.omegaP.push_back(c->omegaP[idx]); t.omegaN.push_back(c->omegaN[idx]); }
    if (gauss) t.gaussians.push_back(c->gaussians[idx]);
  }
  *c = t;
  if (kept_out) std::memcpy(kept_out, kept.data(), sizeof(int) * kept.size());
  return (int)kept.size();
}

/* Cloud::save (cloud.cpp:84-136).  Text records exactly as the reference writes them (operator<< on floats, precision 6).  The
 * reference's binary records are raw dumps of Point / Normal / Stats objects (32 + 32 + 112 bytes on x86-64 Itanium ABI, including
 * each object's vptr and padding); here the same offsets are written with the non-data bytes zeroed. */
static void put_obj(std::ostream& os, const float* v4) { char rec[32]; std::memset(rec, 0, 32); std::memcpy(rec + 16, v4, 16); os.write(rec, 32); }
static void put_stats(std::ostream& os, const Stats& st) {
  char rec[112]; std::memset(rec, 0, 112);
  std::memcpy(rec + 16, st.m.m, 64); std::memcpy(rec + 80, &st.n, 4); std::memcpy(rec + 84, st.eig, 12);
  rec[96] = st.curvatureComputed ? 1 : 0; std::memcpy(rec + 100, &st.curv, 4);
  os.write(rec, 112);
}
int orc_cloud_save(const orc_cloud* c, const char* filename, const float T[16], int step, int binary) {
  std::ofstream os(filename);
  if (!os) return 0;
  os << "PWNCLOUD " << c->points.size() / step << " " << (binary ? true : false) << std::endl;
  float tv[6]; t2v(m4_load(T), tv);
  os << tv[0] << " " << tv[1] << " " << tv[2] << " " << tv[3] << " " << tv[4] << " " << tv[5] << " " << std::endl;
  for (size_t i = 0; i < c->points.size(); i += step) {
    if (!binary) {
      os << "POINTWITHSTATS ";
      for (int k = 0; k < 3; ++k) os << c->points[i].v[k] << " ";
      for (int k = 0; k < 3; ++k) os << c->normals[i].v[k] << " ";
      for (int r = 0; r < 4; ++r) for (int cc = 0; cc < 4; ++cc) os << c->stats[i].m(r, cc) << " ";
      os << std::endl;
    } else {
      put_obj(os, c->points[i].v); put_obj(os, c->normals[i].v); put_stats(os, c->stats[i]);
    }
  }
  return os.good() ? 1 : 0;
}
/* Cloud::load (cloud.cpp:25-82) */
int orc_cloud_load(orc_cloud* c, const char* filename, float T_out[16]) {
  std::ifstream is(filename);
  if (!is) return 0;
  c->points.clear(); c->normals.clear();
  char buf[1024];
  is.getline(buf, 1024);
  std::istringstream ls(buf);
  std::string tag; size_t numPoints = 0; bool binary = false;
  ls >> tag;
  if (tag != "PWNCLOUD") return 0;
  ls >> numPoints >> binary;
  c->points.assign(numPoints, V4{ {0, 0, 0, 1} }); c->normals.assign(numPoints, V4{ {0, 0, 0, 0} }); c->stats.assign(numPoints, Stats());
  is.getline(buf, 1024);
  std::istringstream lst(buf);
  float tv[6] = {0, 0, 0, 0, 0, 0};
  lst >> tv[0] >> tv[1] >> tv[2] >> tv[3] >> tv[4] >> tv[5];
  const M4 T = v2t(tv); std::memcpy(T_out, T.m, sizeof(T.m));
  size_t k = 0;
  while (k < c->points.size() && is.good()) {
    if (!binary) {
      is.getline(buf, 1024);
      std::istringstream l2(buf);
      std::string s2; l2 >> s2;
      if (s2 != "POINTWITHSTATS") continue;
      for (int i = 0; i < 3 && l2; ++i) l2 >> c->points[k].v[i];
      for (int i = 0; i < 3 && l2; ++i) l2 >> c->normals[k].v[i];
      for (int r = 0; r < 4 && l2; ++r) for (int cc = 0; cc < 4 && l2; ++cc) l2 >> c->stats[k].m(r, cc);
    } else {
      char rec[176];
      is.read(rec, 176);
      std::memcpy(c->points[k].v, rec + 16, 16); std::memcpy(c->normals[k].v, rec + 32 + 16, 16);
      const char* st = rec + 64; Stats& S = c->stats[k];
      std::memcpy(S.m.m, st + 16, 64); std::memcpy(&S.n, st + 80, 4); std::memcpy(S.eig, st + 84, 12);
      S.curvatureComputed = st[96] != 0; std::memcpy(&S.curv, st + 100, 4);
    }
    ++k;
  }
  return is.good() ? 1 : 0;
}

}  // extern "C"
