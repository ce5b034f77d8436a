// pwn_math.h -- small fixed-size fp32 math shared by host code and gfx950 kernels.
//
// The reference does this math with Eigen (fixed-size Matrix3f/Matrix4f/Isometry3f, Quaternionf,
// SelfAdjointEigenSolver<Matrix3f>::computeDirect, Matrix6f::ldlt()).  Eigen is not a dependency
// here; the routines below reproduce its evaluation order (inner products left to right, no FMA:
// this file must be compiled with -ffp-contract=off) so that host and device agree bit for bit
// and both agree with the CPU path they replace.  Matrices are column-major like Eigen's.
#pragma once

#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>

#define PWN_HD __host__ __device__ __forceinline__

namespace pwnhip {

struct Mat3 {
  float m[9];
  PWN_HD float& operator()(int r, int c) { return m[r + 3 * c]; }
  PWN_HD float operator()(int r, int c) const { return m[r + 3 * c]; }
};
struct Mat4 {
  float m[16];
  PWN_HD float& operator()(int r, int c) { return m[r + 4 * c]; }
  PWN_HD float operator()(int r, int c) const { return m[r + 4 * c]; }
};
struct Vec3 { float x, y, z; };

PWN_HD Mat4 mat4_identity() {
  Mat4 r;
  for (int i = 0; i < 16; ++i) r.m[i] = 0.f;
  r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.f;
  return r;
}
PWN_HD Mat4 mat4_from(const float* p) { Mat4 r; for (int i = 0; i < 16; ++i) r.m[i] = p[i]; return r; }
PWN_HD Mat3 mat3_from(const float* p) { Mat3 r; for (int i = 0; i < 9; ++i) r.m[i] = p[i]; return r; }
PWN_HD void set_last_row(Mat4& T) { T(3,0) = 0.f; T(3,1) = 0.f; T(3,2) = 0.f; T(3,3) = 1.f; }

// 3-term / 4-term inner products, strictly left to right
PWN_HD float dot3seq(float a0, float b0, float a1, float b1, float a2, float b2) {
  float s = a0 * b0; s = s + a1 * b1; s = s + a2 * b2; return s;
}
PWN_HD float dot4seq(float a0, float b0, float a1, float b1, float a2, float b2, float a3, float b3) {
  float s = a0 * b0; s = s + a1 * b1; s = s + a2 * b2; s = s + a3 * b3; return s;
}

PWN_HD Mat3 mat3_mul(const Mat3& A, const Mat3& B) {
  Mat3 R;
  for (int j = 0; j < 3; ++j)
    for (int i = 0; i < 3; ++i) R(i,j) = dot3seq(A(i,0), B(0,j), A(i,1), B(1,j), A(i,2), B(2,j));
  return R;
}
PWN_HD Vec3 mat3_mul_vec(const Mat3& A, const Vec3& v) {
  Vec3 r;
  r.x = dot3seq(A(0,0), v.x, A(0,1), v.y, A(0,2), v.z);
  r.y = dot3seq(A(1,0), v.x, A(1,1), v.y, A(1,2), v.z);
  r.z = dot3seq(A(2,0), v.x, A(2,1), v.y, A(2,2), v.z);
  return r;
}
PWN_HD Mat4 mat4_mul(const Mat4& A, const Mat4& B) {
  Mat4 R;
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 4; ++i) R(i,j) = dot4seq(A(i,0), B(0,j), A(i,1), B(1,j), A(i,2), B(2,j), A(i,3), B(3,j));
  return R;
}
PWN_HD Mat3 iso_linear(const Mat4& T) { Mat3 R; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R(i,j) = T(i,j); return R; }
PWN_HD Mat4 iso_make(const Mat3& R, const Vec3& t) {
  Mat4 T;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T(i,j) = R(i,j);
  T(0,3) = t.x; T(1,3) = t.y; T(2,3) = t.z;
  set_last_row(T);
  return T;
}
// Isometry3f::inverse(): linear = R^T, translation = (-R^T) * t
PWN_HD Mat4 iso_inverse(const Mat4& T) {
  Mat3 Rt, nRt;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { Rt(i,j) = T(j,i); nRt(i,j) = -T(j,i); }
  Vec3 t = { T(0,3), T(1,3), T(2,3) };
  return iso_make(Rt, mat3_mul_vec(nRt, t));
}
// Isometry3f * Isometry3f: linear = Ra*Rb, translation = Ra*tb + ta
PWN_HD Mat4 iso_mul(const Mat4& A, const Mat4& B) {
  Mat3 Ra = iso_linear(A);
  Mat3 R = mat3_mul(Ra, iso_linear(B));
  Vec3 tb = { B(0,3), B(1,3), B(2,3) };
  Vec3 t = mat3_mul_vec(Ra, tb);
  t.x = t.x + A(0,3); t.y = t.y + A(1,3); t.z = t.z + A(2,3);
  return iso_make(R, t);
}
PWN_HD float cof3(const Mat3& m, int i, int j) {
  const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m(i1,j1) * m(i2,j2) - m(i1,j2) * m(i2,j1);
}
// Matrix3f::inverse(): cofactor expansion, determinant along column 0
PWN_HD Mat3 mat3_inverse(const Mat3& m) {
  const float c00 = cof3(m,0,0), c10 = cof3(m,1,0), c20 = cof3(m,2,0);
  const float invdet = 1.0f / dot3seq(c00, m(0,0), c10, m(1,0), c20, m(2,0));
  Mat3 r;
  r(0,0) = c00 * invdet; r(0,1) = c10 * invdet; r(0,2) = c20 * invdet;
  r(1,0) = cof3(m,0,1) * invdet; r(1,1) = cof3(m,1,1) * invdet; r(1,2) = cof3(m,2,1) * invdet;
  r(2,0) = cof3(m,0,2) * invdet; r(2,1) = cof3(m,1,2) * invdet; r(2,2) = cof3(m,2,2) * invdet;
  return r;
}

// PinholePointProjector::_updateMatrices (reference pwn_core/pinholepointprojector.cpp:17-31)
PWN_HD void projector_matrices(const Mat3& K, const Mat4& T, Mat4& KRt, Mat4& iKRt, Mat3& iK) {
  Mat4 t = iso_inverse(T);
  set_last_row(t);
  iK = mat3_inverse(K);
  const Mat3 KR = mat3_mul(K, iso_linear(t));
  const Vec3 tt = { t(0,3), t(1,3), t(2,3) };
  const Vec3 Kt = mat3_mul_vec(K, tt);
  const Mat3 iKR = mat3_mul(iso_linear(T), iK);
  KRt = mat4_identity(); iKRt = mat4_identity();
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { KRt(i,j) = KR(i,j); iKRt(i,j) = iKR(i,j); }
  KRt(0,3) = Kt.x; KRt(1,3) = Kt.y; KRt(2,3) = Kt.z;
  iKRt(0,3) = T(0,3); iKRt(1,3) = T(1,3); iKRt(2,3) = T(2,3);
}

// pwn_core/bm_se3.h:9-22
PWN_HD Mat3 quat2mat(float qx, float qy, float qz) {
  const float qw = sqrtf(1.f - dot3seq(qx, qx, qy, qy, qz, qz));
  Mat3 R;
  R(0,0) = qw*qw + qx*qx - qy*qy - qz*qz; R(0,1) = 2*(qx*qy - qw*qz);            R(0,2) = 2*(qx*qz + qw*qy);
  R(1,0) = 2*(qx*qy + qz*qw);             R(1,1) = qw*qw - qx*qx + qy*qy - qz*qz; R(1,2) = 2*(qy*qz - qx*qw);
  R(2,0) = 2*(qx*qz - qy*qw);             R(2,1) = 2*(qy*qz + qx*qw);             R(2,2) = qw*qw - qx*qx - qy*qy + qz*qz;
  return R;
}
// Quaternionf(Matrix3f) + normalize() + sign fix (pwn_core/bm_se3.h:25-35)
PWN_HD Vec3 mat2quat(const Mat3& R) {
  float x, y, z, w;
  float t = (R(0,0) + R(1,1)) + R(2,2);
  if (t > 0.f) {
    t = sqrtf(t + 1.0f);
    w = 0.5f * t;
    t = 0.5f / t;
    x = (R(2,1) - R(1,2)) * t;
    y = (R(0,2) - R(2,0)) * t;
    z = (R(1,0) - R(0,1)) * t;
  } else {
    // Eigen: i = index of the largest diagonal entry, j = (i+1)%3, k = (j+1)%3; q[i] = t/2, w = (R(k,j)-R(j,k)) t', q[j] = (R(j,i)+R(i,j)) t',
    // q[k] = (R(k,i)+R(i,k)) t' -- written out per i so that no index is a run-time value (registers, not scratch, on the device)
    int i = 0;
    if (R(1,1) > R(0,0)) i = 1;
    if (R(2,2) > (i == 0 ? R(0,0) : R(1,1))) i = 2;
    if (i == 0) {
      t = sqrtf(R(0,0) - R(1,1) - R(2,2) + 1.0f);
      x = 0.5f * t; t = 0.5f / t;
      w = (R(2,1) - R(1,2)) * t; y = (R(1,0) + R(0,1)) * t; z = (R(2,0) + R(0,2)) * t;
    } else if (i == 1) {
      t = sqrtf(R(1,1) - R(2,2) - R(0,0) + 1.0f);
      y = 0.5f * t; t = 0.5f / t;
      w = (R(0,2) - R(2,0)) * t; z = (R(2,1) + R(1,2)) * t; x = (R(0,1) + R(1,0)) * t;
    } else {
      t = sqrtf(R(2,2) - R(0,0) - R(1,1) + 1.0f);
      z = 0.5f * t; t = 0.5f / t;
      w = (R(1,0) - R(0,1)) * t; x = (R(0,2) + R(2,0)) * t; y = (R(1,2) + R(2,1)) * t;
    }
  }
  const float n = sqrtf((x*x + y*y) + (z*z + w*w));
  x = x / n; y = y / n; z = z / n; w = w / n;
  Vec3 r = { x, y, z };
  if (w < 0.f) { r.x = -r.x; r.y = -r.y; r.z = -r.z; }
  return r;
}
// pwn_core/bm_se3.h:37-43
PWN_HD Mat4 v2t(const float x[6]) {
  const Vec3 t = { x[0], x[1], x[2] };
  return iso_make(quat2mat(x[3], x[4], x[5]), t);
}
// pwn_core/bm_se3.h:45-52
PWN_HD void t2v(const Mat4& T, float v[6]) {
  v[0] = T(0,3); v[1] = T(1,3); v[2] = T(2,3);
  const Vec3 q = mat2quat(iso_linear(T));
  v[3] = q.x; v[4] = q.y; v[5] = q.z;
}

#define PWN_TRIG_HD PWN_HD
#define PWN_TRIG_SIGNBIT(x) __builtin_signbit(x)
// ---- trig of the 3x3 eigensolver, double precision, + - * / only -----------------------------------------------------
// The eigensolver needs atan2(sqrt(q), half_b) / 3 and cos / sin of that angle as FLOATS (the reference calls the float libm).
// Canonical evaluation: in double by the fixed algorithms below (no libm, no FMA: every operation is an IEEE + - * /, so host
// and device produce the same bits), rounded once to float.  Their error is < 1e-15, so the rounded float is the correctly
// rounded value -- what a current glibc's atan2f / cosf / sinf return -- except when the true value lies within 1e-15 of a
// rounding boundary (about one call in 1e8).
//   atan2(y, x), y >= 0:  t = min/max in [0, 1]; centre c_k = tan(k pi/16), k = 0..4, nearest in angle;
//                         u = (min - c max) / (max + c min), |u| <= tan(pi/32); atan(u) by its series to u^19; unfold.
//   sin / cos on [0, pi/3]: Taylor series to x^21 / x^22.
PWN_TRIG_HD double pwn_atan2_pos(double y, double x) {
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
PWN_TRIG_HD void pwn_sincos_small(double x, double& s, double& c) {      // 0 <= x <= pi/3 (a little beyond is fine)
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

// ---- SelfAdjointEigenSolver<Matrix3f>::computeDirect(A, ComputeEigenvectors) -------------------------
// (closed-form roots of the characteristic polynomial on the shifted+scaled matrix, eigenvectors by
// kernel extraction with cross products).  Reads the lower triangle.  evals ascending.
PWN_HD Vec3 cross3(const Vec3& a, const Vec3& b) {
  Vec3 r; r.x = a.y*b.z - a.z*b.y; r.y = a.z*b.x - a.x*b.z; r.z = a.x*b.y - a.y*b.x; return r;
}
PWN_HD float sqn3(const Vec3& a) { return dot3seq(a.x, a.x, a.y, a.y, a.z, a.z); }
PWN_HD Vec3 col3(const Mat3& m, int c) { Vec3 r = { m(0,c), m(1,c), m(2,c) }; return r; }

PWN_HD void eig3_kernel(const Mat3& mat, Vec3& res, Vec3& representative) {
  int i0 = 0; float best = fabsf(mat(0,0));
  if (fabsf(mat(1,1)) > best) { best = fabsf(mat(1,1)); i0 = 1; }
  if (fabsf(mat(2,2)) > best) { i0 = 2; }
  // columns selected without dynamic indexing
  const Vec3 a0 = col3(mat, 0), a1 = col3(mat, 1), a2 = col3(mat, 2);
  const Vec3 rep = (i0 == 0) ? a0 : (i0 == 1 ? a1 : a2);
  const Vec3 n1 = (i0 == 0) ? a1 : (i0 == 1 ? a2 : a0);
  const Vec3 n2 = (i0 == 0) ? a2 : (i0 == 1 ? a0 : a1);
  representative = rep;
  const Vec3 c0 = cross3(rep, n1), c1 = cross3(rep, n2);
  const float s0 = sqn3(c0), s1 = sqn3(c1);
  if (s0 > s1) { const float s = sqrtf(s0); res.x = c0.x / s; res.y = c0.y / s; res.z = c0.z / s; }
  else         { const float s = sqrtf(s1); res.x = c1.x / s; res.y = c1.y / s; res.z = c1.z / s; }
}

// a00..a22: lower triangle of the symmetric input (a10 = A(1,0) ...). Outputs eigenvalues e[3] and
// eigenvectors v0,v1,v2 (columns).
PWN_HD void eig3_direct(float a00, float a10, float a20, float a11, float a21, float a22,
                        float e[3], Vec3& v0, Vec3& v1, Vec3& v2) {
  const float eps = FLT_EPSILON;
  const float shift = ((a00 + a11) + a22) / 3.0f;
  Mat3 S;
  S(0,0) = a00 - shift; S(1,1) = a11 - shift; S(2,2) = a22 - shift;
  S(1,0) = a10; S(0,1) = a10; S(2,0) = a20; S(0,2) = a20; S(2,1) = a21; S(1,2) = a21;
  float scale = 0.f;
  for (int k = 0; k < 9; ++k) scale = fmaxf(scale, fabsf(S.m[k]));
  if (scale > 0.f) for (int k = 0; k < 9; ++k) S.m[k] = S.m[k] / scale;
  {  // roots
    const float s_inv3 = 1.0f / 3.0f;
    const float s_sqrt3 = sqrtf(3.0f);
    const float m00 = S(0,0), m11 = S(1,1), m22 = S(2,2), m10 = S(1,0), m20 = S(2,0), m21 = S(2,1);
    const float c0 = m00*m11*m22 + 2.0f*m10*m20*m21 - m00*m21*m21 - m11*m20*m20 - m22*m10*m10;
    const float c1 = m00*m11 - m10*m10 + m00*m22 - m20*m20 + m11*m22 - m21*m21;
    const float c2 = m00 + m11 + m22;
    const float c2_over_3 = c2 * s_inv3;
    float a_over_3 = (c2 * c2_over_3 - c1) * s_inv3;
    a_over_3 = fmaxf(a_over_3, 0.0f);
    const float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
    float q = a_over_3 * a_over_3 * a_over_3 - half_b * half_b;
    q = fmaxf(q, 0.0f);
    const float rho = sqrtf(a_over_3);
    // The three trig calls: fixed double-precision algorithms rounded once to float (see pwn_atan2_pos): the same bits on the host and on
    // the device, a third of the instructions of the ocml double routines (this kernel is VALU-bound).
    const float theta = (float)pwn_atan2_pos((double)sqrtf(q), (double)half_b) * s_inv3;
    double sd, cd;
    pwn_sincos_small((double)theta, sd, cd);
    const float cos_theta = (float)cd;
    const float sin_theta = (float)sd;
    e[0] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
    e[1] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
    e[2] = c2_over_3 + 2.0f * rho * cos_theta;
  }
  if ((e[2] - e[0]) <= eps) {
    v0.x = 1.f; v0.y = 0.f; v0.z = 0.f; v1.x = 0.f; v1.y = 1.f; v1.z = 0.f; v2.x = 0.f; v2.y = 0.f; v2.z = 1.f;
  } else {
    float d0 = e[2] - e[1];
    const float d1 = e[1] - e[0];
    const bool swapped = d0 > d1;        // k = swapped ? 2 : 0 ; l = swapped ? 0 : 2
    if (swapped) d0 = d1;
    const float ek = swapped ? e[2] : e[0], el = swapped ? e[0] : e[2];
    Mat3 tmp = S;
    tmp(0,0) = tmp(0,0) - ek; tmp(1,1) = tmp(1,1) - ek; tmp(2,2) = tmp(2,2) - ek;
    Vec3 vk, vl;
    eig3_kernel(tmp, vk, vl);
    if (d0 <= 2 * eps * d1) {
      const float d = dot3seq(vk.x, vl.x, vk.y, vl.y, vk.z, vl.z);
      vl.x = vl.x - d * vl.x; vl.y = vl.y - d * vl.y; vl.z = vl.z - d * vl.z;
      const float n = sqrtf(sqn3(vl));
      vl.x = vl.x / n; vl.y = vl.y / n; vl.z = vl.z / n;
    } else {
      tmp = S;
      tmp(0,0) = tmp(0,0) - el; tmp(1,1) = tmp(1,1) - el; tmp(2,2) = tmp(2,2) - el;
      Vec3 dummy;
      eig3_kernel(tmp, vl, dummy);
    }
    v0 = swapped ? vl : vk;
    v2 = swapped ? vk : vl;
    v1 = cross3(v2, v0);
    const float z = sqn3(v1);
    if (z > 0.f) { const float n = sqrtf(z); v1.x = v1.x / n; v1.y = v1.y / n; v1.z = v1.z / n; }
  }
  for (int i = 0; i < 3; ++i) { e[i] = e[i] * scale; e[i] = e[i] + shift; }
}

// ---- Matrix6f::ldlt().solve(b): pivoted (largest |diagonal|) LDL^T, fp32 -------------------------------
// Every index is a compile-time constant (the step k is a template parameter, the pivot row p selects one of the
// statically indexed swap bodies), so that the 6x6 factorisation lives in registers on the device: the one thread per pair
// that runs it would otherwise walk a chain of dependent LDS (or scratch) round trips.  The arithmetic -- the operations and
// their order -- is that of Eigen's ldlt_inplace<Lower>::unblocked followed by LDLT::_solve_impl.
#define PWN_A(r, c) A[(r) + 6 * (c)]
// conditional swap as two selects: the addresses stay compile-time constants whatever the pivot row is (a branch per pivot row
// gets its stores merged into one store through a selected address, which puts the matrix into scratch memory)
PWN_HD void ldlt_cswap(bool m, float& a, float& b) { const float ta = m ? b : a, tb = m ? a : b; a = ta; b = tb; }
template <int K, int P> PWN_HD void ldlt_pivot_swap(float (&A)[36], bool m) {    // symmetric row/column swap K <-> P, lower triangle
#pragma unroll
  for (int j = 0; j < K; ++j) ldlt_cswap(m, PWN_A(K, j), PWN_A(P, j));
#pragma unroll
  for (int i = P + 1; i < 6; ++i) ldlt_cswap(m, PWN_A(i, K), PWN_A(i, P));
  ldlt_cswap(m, PWN_A(K, K), PWN_A(P, P));
#pragma unroll
  for (int i = K + 1; i < P; ++i) ldlt_cswap(m, PWN_A(i, K), PWN_A(P, i));
}
template <int K, int P> PWN_HD void ldlt_pivot_dispatch(float (&A)[36], int p) {
  if constexpr (P < 6) {
    ldlt_pivot_swap<K, P>(A, p == P);
    ldlt_pivot_dispatch<K, P + 1>(A, p);
  }
}
template <int K, int P> PWN_HD void ldlt_perm_dispatch(float (&d)[6], int p) {   // swap d[K] <-> d[p]
  if constexpr (P < 6) {
    ldlt_cswap(p == P, d[K], d[P]);
    ldlt_perm_dispatch<K, P + 1>(d, p);
  }
}
// one step of the factorisation; false = the remaining diagonal is below the cutoff (the factorisation stops)
template <int K> PWN_HD bool ldlt_step(float (&A)[36], int (&tr)[6], float& cutoff) {
  int p = K; float big = fabsf(PWN_A(K, K));
#pragma unroll
  for (int i = K + 1; i < 6; ++i) if (fabsf(PWN_A(i, i)) > big) { big = fabsf(PWN_A(i, i)); p = i; }
  if (K == 0) cutoff = fabsf(FLT_EPSILON * big);
  if (big < cutoff) return false;
  tr[K] = p;
  ldlt_pivot_dispatch<K, K + 1>(A, p);
  if constexpr (K > 0) {
    float temp[K];
#pragma unroll
    for (int j = 0; j < K; ++j) temp[j] = PWN_A(j, j) * PWN_A(K, j);
    float d = PWN_A(K, 0) * temp[0];
#pragma unroll
    for (int j = 1; j < K; ++j) d = d + PWN_A(K, j) * temp[j];
    PWN_A(K, K) = PWN_A(K, K) - d;
#pragma unroll
    for (int i = K + 1; i < 6; ++i) {
      float s = PWN_A(i, 0) * temp[0];
#pragma unroll
      for (int j = 1; j < K; ++j) s = s + PWN_A(i, j) * temp[j];
      PWN_A(i, K) = PWN_A(i, K) - s;
    }
  }
  if constexpr (K + 1 < 6) {
    if (fabsf(PWN_A(K, K)) > cutoff) {
#pragma unroll
      for (int i = K + 1; i < 6; ++i) PWN_A(i, K) = PWN_A(i, K) / PWN_A(K, K);
    }
  }
  return true;
}
PWN_HD void ldlt_solve6(const float Hin[36], const float bin[6], float x[6]) {
  float A[36];
#pragma unroll
  for (int i = 0; i < 36; ++i) A[i] = Hin[i];
  int tr[6] = {0, 1, 2, 3, 4, 5};
  float cutoff = 0.f;
  (void)(ldlt_step<0>(A, tr, cutoff) && ldlt_step<1>(A, tr, cutoff) && ldlt_step<2>(A, tr, cutoff) && ldlt_step<3>(A, tr, cutoff) &&
         ldlt_step<4>(A, tr, cutoff) && ldlt_step<5>(A, tr, cutoff));
  float d[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) d[i] = bin[i];
  ldlt_perm_dispatch<0, 1>(d, tr[0]);
  ldlt_perm_dispatch<1, 2>(d, tr[1]);
  ldlt_perm_dispatch<2, 3>(d, tr[2]);
  ldlt_perm_dispatch<3, 4>(d, tr[3]);
  ldlt_perm_dispatch<4, 5>(d, tr[4]);
#pragma unroll
  for (int i = 1; i < 6; ++i) {                          // L^-1 (unit lower)
    float s = PWN_A(i, 0) * d[0];
#pragma unroll
    for (int j = 1; j < i; ++j) s = s + PWN_A(i, j) * d[j];
    d[i] = d[i] - s;
  }
  float dmax = 0.f;
#pragma unroll
  for (int i = 0; i < 6; ++i) dmax = fmaxf(dmax, fabsf(PWN_A(i, i)));
  const float tol = fmaxf(dmax * FLT_EPSILON, 1.0f / FLT_MAX);
#pragma unroll
  for (int i = 0; i < 6; ++i) { if (fabsf(PWN_A(i, i)) > tol) d[i] = d[i] / PWN_A(i, i); else d[i] = 0.f; }
#pragma unroll
  for (int i = 4; i >= 0; --i) {                         // L^-T
    float s = PWN_A(i + 1, i) * d[i + 1];
#pragma unroll
    for (int j = i + 2; j < 6; ++j) s = s + PWN_A(j, i) * d[j];
    d[i] = d[i] - s;
  }
  ldlt_perm_dispatch<4, 5>(d, tr[4]);
  ldlt_perm_dispatch<3, 4>(d, tr[3]);
  ldlt_perm_dispatch<2, 3>(d, tr[2]);
  ldlt_perm_dispatch<1, 2>(d, tr[1]);
  ldlt_perm_dispatch<0, 1>(d, tr[0]);
#pragma unroll
  for (int i = 0; i < 6; ++i) x[i] = d[i];
}
#undef PWN_A

}  // namespace pwnhip
