// pwn_stats.h -- host-side 6x6 math of Aligner::_computeStatistics (reference pwn_core/aligner.cpp:152-199,
// pwn_core/unscented.h:23-65): covariance of the estimate from the final normal equations, unscented remap into the
// (t, q) chart of the solution, information matrix and the translational / rotational eigen-ratios.
// The reference uses Eigen's JacobiSVD / LLT / Matrix6f::inverse; symmetric cyclic Jacobi, Cholesky and Gauss-Jordan
// stand in for them here (results agree to fp32 round-off).  Host only.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include "pwn_math.h"

namespace pwnhip {

// eigen-decomposition of a symmetric n x n matrix (column-major, n <= 6) by cyclic Jacobi rotations
inline void sym_eigen(int n, float* A, float* V, float* w) {
  for (int i = 0; i < n * n; ++i) V[i] = 0.f;
  for (int i = 0; i < n; ++i) V[i + n * i] = 1.f;
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
        const float c = 1.f / std::sqrt(t * t + 1.f), s = t * c;
        for (int k = 0; k < n; ++k) { const float x = A[k + n * p], y = A[k + n * q]; A[k + n * p] = c * x - s * y; A[k + n * q] = s * x + c * y; }
        for (int k = 0; k < n; ++k) { const float x = A[p + n * k], y = A[q + n * k]; A[p + n * k] = c * x - s * y; A[q + n * k] = s * x + c * y; }
        for (int k = 0; k < n; ++k) { const float x = V[k + n * p], y = V[k + n * q]; V[k + n * p] = c * x - s * y; V[k + n * q] = s * x + c * y; }
      }
  }
  for (int i = 0; i < n; ++i) w[i] = A[i + n * i];
}
inline bool gauss_jordan_inverse(int n, const float* Ain, float* out) {
  float a[36], b[36];
  for (int i = 0; i < n * n; ++i) { a[i] = Ain[i]; b[i] = 0.f; }
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
  for (int i = 0; i < n * n; ++i) out[i] = b[i];
  return true;
}

// H: Linearizer::H() at the final transform (no damping), column-major; T: Aligner::_T.
inline void compute_statistics(const float Hin[36], const Mat4& T, float mean[6], float omega[36], float* translationalRatio, float* rotationalRatio) {
  const int n = 6;
  float H[36], V[36], w[6];
  for (int i = 0; i < 36; ++i) H[i] = Hin[i];
  for (int i = 0; i < n; ++i) H[i + n * i] += 1.0f;                       // aligner.cpp:169
  sym_eigen(n, H, V, w);                                                  // :172 (JacobiSVD of a symmetric PSD matrix)
  float smax = 0.f;
  for (int i = 0; i < n; ++i) smax = std::max(smax, std::fabs(w[i]));
  float sigma[36];
  for (int i = 0; i < 36; ++i) sigma[i] = 0.f;
  for (int k = 0; k < n; ++k) {                                           // :173 svd.solve(Identity): pseudo-inverse
    if (std::fabs(w[k]) <= FLT_EPSILON * n * smax) continue;
    const float inv = 1.0f / w[k];
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) sigma[i + n * j] += V[i + n * k] * inv * V[j + n * k];
  }
  // unscented.h:23-50
  const double alpha = 1e-3, beta = 2., lambda = alpha * alpha * n;
  const double wi = 1. / (2. * (n + lambda));
  float L[36], C[36];
  for (int i = 0; i < 36; ++i) { L[i] = 0.f; C[i] = sigma[i] * (float)(n + lambda); }
  for (int j = 0; j < n; ++j) {
    float d = C[j + n * j];
    for (int k = 0; k < j; ++k) d -= L[j + n * k] * L[j + n * k];
    d = std::sqrt(d);
    L[j + n * j] = d;
    for (int i = j + 1; i < n; ++i) { float v = C[i + n * j]; for (int k = 0; k < j; ++k) v -= L[i + n * k] * L[j + n * k]; L[i + n * j] = v / d; }
  }
  float samples[13][6]; double wI[13], wP[13];
  for (int r = 0; r < n; ++r) samples[0][r] = 0.f;
  wI[0] = lambda / (n + lambda); wP[0] = lambda / (n + lambda) + (1. - alpha * alpha + beta);
  for (int i = 0, k = 1; i < n; ++i, k += 2) {
    for (int r = 0; r < n; ++r) { samples[k][r] = L[r + n * i]; samples[k + 1][r] = -L[r + n * i]; }
    wI[k] = wP[k] = wi; wI[k + 1] = wP[k + 1] = wi;
  }
  for (int k = 0; k < 13; ++k) {                                          // aligner.cpp:182-185
    const Mat4 X = iso_mul(T, iso_inverse(v2t(samples[k])));
    t2v(X, samples[k]);
  }
  float cov[36];                                                          // unscented.h:52-65
  for (int r = 0; r < n; ++r) mean[r] = 0.f;
  for (int i = 0; i < 36; ++i) cov[i] = 0.f;
  for (int k = 0; k < 13; ++k) for (int r = 0; r < n; ++r) mean[r] += (float)wI[k] * samples[k][r];
  for (int k = 0; k < 13; ++k) {
    float d[6];
    for (int r = 0; r < n; ++r) d[r] = samples[k][r] - mean[r];
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) cov[i + n * j] += (float)wP[k] * (d[i] * d[j]);
  }
  if (!gauss_jordan_inverse(n, cov, omega)) for (int i = 0; i < 36; ++i) omega[i] = 0.f;      // aligner.cpp:190
  for (int blk = 0; blk < 2; ++blk) {                                     // :193-198
    float B[9], BtB[9], Vb[9], eb[3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) B[i + 3 * j] = omega[(i + 3 * blk) + n * (j + 3 * blk)];
    for (int i = 0; i < 9; ++i) BtB[i] = 0.f;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) BtB[i + 3 * j] += B[k + 3 * i] * B[k + 3 * j];
    sym_eigen(3, BtB, Vb, eb);
    float s0 = 0.f, s2 = FLT_MAX;
    for (int i = 0; i < 3; ++i) { const float sv = std::sqrt(std::max(eb[i], 0.f)); s0 = std::max(s0, sv); s2 = std::min(s2, sv); }
    (blk == 0 ? *translationalRatio : *rotationalRatio) = s0 / s2;
  }
}

// ---- SE(3) priors (reference pwn_core/se3_prior.{h,cpp}, consumed by aligner.cpp:96-108) ---------------------------
struct PriorHost {
  int kind;            // 0: SE3RelativePrior (error = t2v(invT * mean)), 1: SE3AbsolutePrior (error = t2v(invT * ref^-1 * mean))
  Mat4 mean;           // _priorMean
  Mat4 invReference;   // _inverseReferenceTransform (identity for relative priors)
  float information[36];
};
inline void prior_error(const PriorHost& pr, const Mat4& mean, const Mat4& invT, float e[6]) {      // se3_prior.cpp:58-60,69-71
  const Mat4 X = pr.kind == 0 ? iso_mul(invT, mean) : iso_mul(iso_mul(invT, pr.invReference), mean);
  t2v(X, e);
}
inline void mat6_mul(const float* A, const float* B, float* R) {      // column-major, left-to-right inner products
  for (int j = 0; j < 6; ++j) for (int i = 0; i < 6; ++i) { float s = A[i] * B[6 * j]; for (int k = 1; k < 6; ++k) s = s + A[i + 6 * k] * B[k + 6 * j]; R[i + 6 * j] = s; }
}
inline void mat6_transpose(const float* A, float* R) { for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) R[i + 6 * j] = A[j + 6 * i]; }
// Adds J^T * Omega' * J to H and J^T * Omega' * e to b (aligner.cpp:99-107) with the numeric Jacobians of se3_prior.cpp:8-52.
inline void prior_accumulate(const PriorHost& pr, const Mat4& invT, float H[36], float b[6]) {
  const float epsilon = 1e-3f, iEpsilon = 0.5f / epsilon;
  float e[6]; prior_error(pr, pr.mean, invT, e);
  float J[36], Jz[36];
  for (int i = 0; i < 6; ++i) {
    float up[6] = {0, 0, 0, 0, 0, 0}, dn[6] = {0, 0, 0, 0, 0, 0}, eu[6], ed[6];
    up[i] = epsilon; dn[i] = -epsilon;
    prior_error(pr, pr.mean, iso_mul(v2t(up), invT), eu); prior_error(pr, pr.mean, iso_mul(v2t(dn), invT), ed);        // jacobian
    for (int r = 0; r < 6; ++r) J[r + 6 * i] = iEpsilon * (eu[r] - ed[r]);
    prior_error(pr, iso_mul(pr.mean, v2t(up)), invT, eu); prior_error(pr, iso_mul(pr.mean, v2t(dn)), invT, ed);        // jacobianZ
    for (int r = 0; r < 6; ++r) Jz[r + 6 * i] = iEpsilon * (eu[r] - ed[r]);
  }
  float iJz[36], iJzT[36], t1[36], Om[36], Jt[36], t2[36], Hp[36];
  if (!gauss_jordan_inverse(6, Jz, iJz)) return;
  mat6_transpose(iJz, iJzT); mat6_mul(iJzT, pr.information, t1); mat6_mul(t1, iJz, Om);                                 // errorInformation
  mat6_transpose(J, Jt); mat6_mul(Jt, Om, t2); mat6_mul(t2, J, Hp);
  for (int i = 0; i < 36; ++i) H[i] = H[i] + Hp[i];
  for (int i = 0; i < 6; ++i) { float sacc = t2[i] * e[0]; for (int k = 1; k < 6; ++k) sacc = sacc + t2[i + 6 * k] * e[k]; b[i] = b[i] + sacc; }
}

}  // namespace pwnhip
