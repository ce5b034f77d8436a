// pwn_scene_kernels.h -- gfx950 kernels of the scene-maintenance stage that follows the registration path
// (SURVEY.md section 8(f) row 4): the per-point sensor-noise Gaussians of PinholePointProjector::unProject, Cloud::add,
// Merger::merge and VoxelCalculator::compute.  Reference: g2o_frontend/pwn_core/{pinholepointprojector.cpp:93-133, gaussian3.h,
// cloud.cpp:145-186, merger.cpp:15-119, voxelcalculator.cpp:15-73} and g2o_frontend/basemath/gaussian.h.
//
// Layout: a Gaussian is 24 floats per point, AoS (it is always read and written whole, by the thread that owns the point):
//   mean[3] cov[9] infoVec[3] info[9], 3x3 blocks column-major like Eigen's; a separate int per point holds the reference's two
//   lazy-evaluation flags (1 = moments valid, 2 = information form valid; gaussian.h:89-94).  Both forms are kept because the
//   reference caches both and fp32 inverse(inverse(A)) != A.
#pragma once

#include "pwn_kernels.h"

namespace pwnhip {

constexpr int kGaussFloats = 24;
struct GaussD { float mean[3]; float cov[9]; float infoVec[3]; float info[9]; };
constexpr int kGaussMoments = 1, kGaussInfo = 2;

struct SceneBuffers {        // optional per-cloud arrays of the scene stage
  GaussD* G;                 // [capacity]
  int* Gf;                   // [capacity] flags
};

__device__ __forceinline__ Mat3 mat3_of(const float* m) { Mat3 r; for (int k = 0; k < 9; ++k) r.m[k] = m[k]; return r; }
__device__ __forceinline__ Mat3 mat3_transpose(const Mat3& a) { Mat3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r(i,j) = a(j,i); return r; }
// Gaussian::_updateInfo / _updateMoments (basemath/gaussian.h:73-87)
__device__ __forceinline__ void gauss_update_info(GaussD& g, int& flags) {
  if (flags & kGaussInfo) return;
  const Mat3 I = mat3_inverse(mat3_of(g.cov));
  for (int k = 0; k < 9; ++k) g.info[k] = I.m[k];
  const Vec3 mu = { g.mean[0], g.mean[1], g.mean[2] };
  const Vec3 v = mat3_mul_vec(I, mu);
  g.infoVec[0] = v.x; g.infoVec[1] = v.y; g.infoVec[2] = v.z;
  flags |= kGaussInfo;
}
__device__ __forceinline__ void gauss_update_moments(GaussD& g, int& flags) {
  if (flags & kGaussMoments) return;
  const Mat3 C = mat3_inverse(mat3_of(g.info));
  for (int k = 0; k < 9; ++k) g.cov[k] = C.m[k];
  const Vec3 iv = { g.infoVec[0], g.infoVec[1], g.infoVec[2] };
  const Vec3 v = mat3_mul_vec(C, iv);
  g.mean[0] = v.x; g.mean[1] = v.y; g.mean[2] = v.z;
  flags |= kGaussMoments;
}
// Gaussian3fVector::transformInPlace (gaussian3.h:65-73): Gaussian3f(R*mean + t, R*cov*R^T, false)
__device__ __forceinline__ void gauss_transform(GaussD& g, int& flags, const Mat4& m) {
  gauss_update_moments(g, flags);
  const Mat3 R = iso_linear(m);
  const Vec3 mu = { g.mean[0], g.mean[1], g.mean[2] };
  Vec3 v = mat3_mul_vec(R, mu);
  g.mean[0] = v.x + m(0,3); g.mean[1] = v.y + m(1,3); g.mean[2] = v.z + m(2,3);
  const Mat3 C = mat3_mul(mat3_mul(R, mat3_of(g.cov)), mat3_transpose(R));
  for (int k = 0; k < 9; ++k) { g.cov[k] = C.m[k]; g.info[k] = 0.f; }
  g.infoVec[0] = g.infoVec[1] = g.infoVec[2] = 0.f;
  flags = kGaussMoments;
}

// ------------------------------------------------------------------------------------------------------------------
// The Gaussian half of PinholePointProjector::unProject(points, gaussians, index, depth) (pinholepointprojector.cpp:104-123)
// followed by Cloud::transformInPlace(sensorOffset) on the Gaussians (cloud.cpp:180).  Same ordered compaction as k_unproject
// (the point index is the row-major rank of the valid pixel); f.rowoff holds the exclusive row offsets (k_row_count + k_row_offsets).
// grid = (rows, 1), block = 256.
__global__ void __launch_bounds__(256) k_gaussians(const FrameDesc* __restrict__ frames, ConvertParams cp, Mat3 iK, float fB, float alpha,
                                                   SceneBuffers sb) {
  const FrameDesc& f = frames[blockIdx.y];
  const int r = blockIdx.x;
  __shared__ int wcount[4];
  int base = f.rowoff[r];
  const int wave = threadIdx.x >> 6, lane = lane_id();
  for (int c0 = 0; c0 < cp.cols; c0 += 256) {
    const int c = c0 + threadIdx.x;
    const bool in = c < cp.cols;
    const float d = in ? frame_depth(f, (size_t)r * cp.cols + c) : 0.f;
    const bool valid = in && !(d < cp.minD || d > cp.maxD);
    const unsigned long long bal = __ballot(valid);
    const int rank = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wcount[wave] = __popcll(bal);
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const int n = wcount[w]; if (w < wave) woff += n; tot += n; }
    if (valid) {
      const int idx = base + woff + rank;
      if (idx < f.cloud.capacity) {
        GaussD g;
        const float a = (float)c * d, b = (float)r * d;
        g.mean[0] = dot4seq(cp.iKRt(0,0), a, cp.iKRt(0,1), b, cp.iKRt(0,2), d, cp.iKRt(0,3), 1.0f);
        g.mean[1] = dot4seq(cp.iKRt(1,0), a, cp.iKRt(1,1), b, cp.iKRt(1,2), d, cp.iKRt(1,3), 1.0f);
        g.mean[2] = dot4seq(cp.iKRt(2,0), a, cp.iKRt(2,1), b, cp.iKRt(2,2), d, cp.iKRt(2,3), 1.0f);
        const float z = d;
        const float zVariation = (alpha * z * z) / (fB + z * alpha);
        Mat3 J;
        J(0,0) = z;   J(0,1) = 0.f; J(0,2) = (float)c;
        J(1,0) = 0.f; J(1,1) = z;   J(1,2) = (float)r;
        J(2,0) = 0.f; J(2,1) = 0.f; J(2,2) = 1.f;
        J = mat3_mul(iK, J);
        const float dg[3] = { 3.0f, 3.0f, zVariation };
        Mat3 JD;
        for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) JD(i,k) = J(i,k) * dg[k];
        const Mat3 C = mat3_mul(JD, mat3_transpose(J));
        for (int k = 0; k < 9; ++k) { g.cov[k] = C.m[k]; g.info[k] = 0.f; }
        g.infoVec[0] = g.infoVec[1] = g.infoVec[2] = 0.f;
        int flags = kGaussMoments;
        if (cp.hasOffset) gauss_transform(g, flags, cp.offset);
        sb.G[idx] = g;
        sb.Gf[idx] = flags;
      }
    }
    base += tot;
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Cloud::add (cloud.cpp:145-171): dst[k + i] = transformInPlace(T)(src[i]).  The scene cloud keeps per-point normal information
// matrices (9 planes): clouds appended with different transforms have different class matrices.  grid = ceil(n/256), block = 256.
__device__ __forceinline__ void omega_transform(const Mat4& m, float* om /* row-major 3x3, in place */) {
  float t1[9];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) t1[3 * a + b] = dot3seq(m(a,0), om[0 + b], m(a,1), om[3 + b], m(a,2), om[6 + b]);
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) om[3 * a + b] = dot3seq(t1[3 * a], m(b,0), t1[3 * a + 1], m(b,1), t1[3 * a + 2], m(b,2));
}
__global__ void __launch_bounds__(256) k_cloud_append(CloudDev dst, SceneBuffers dsb, CloudDev src, SceneBuffers ssb, int k, int n, int ngauss,
                                                      Mat4 m, int identity) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n || k + i >= dst.capacity) return;
  const int o = k + i;
  float4 P, Nm;                    // old-format view: (x, y, z, curvature), (normal, class word)
  cloud_get(src, i, P, Nm);
  float omP[9], omN[9];
  const int cls = __float_as_int(Nm.w) & kClsMask;
#pragma unroll
  for (int q = 0; q < 9; ++q) omP[q] = src.Om[omp_at(src.capacity, i, q, src.omSym)];
  if (src.OmN) {
#pragma unroll
    for (int q = 0; q < 9; ++q) omN[q] = src.OmN[om_at(src.capacity, i, q)];
  } else {
#pragma unroll
    for (int q = 0; q < 9; ++q) omN[q] = (cls == 1) ? src.omN[0][q] : ((cls == 2) ? src.omN[1][q] : 0.f);
  }
  if (!identity) {       // Cloud::transformInPlace (cloud.cpp:173-186)
    const float px = dot4seq(m(0,0), P.x, m(0,1), P.y, m(0,2), P.z, m(0,3), 1.0f);
    const float py = dot4seq(m(1,0), P.x, m(1,1), P.y, m(1,2), P.z, m(1,3), 1.0f);
    const float pz = dot4seq(m(2,0), P.x, m(2,1), P.y, m(2,2), P.z, m(2,3), 1.0f);
    P.x = px; P.y = py; P.z = pz;
    const float tx = dot4seq(m(0,0), Nm.x, m(0,1), Nm.y, m(0,2), Nm.z, m(0,3), 0.0f);
    const float ty = dot4seq(m(1,0), Nm.x, m(1,1), Nm.y, m(1,2), Nm.z, m(1,3), 0.0f);
    const float tz = dot4seq(m(2,0), Nm.x, m(2,1), Nm.y, m(2,2), Nm.z, m(2,3), 0.0f);
    Nm.x = tx; Nm.y = ty; Nm.z = tz;
    omega_transform(m, omP);
    omega_transform(m, omN);
  }
  cloud_put(dst, o, P, Nm);
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    if (!(dst.omSym && om_is_lower(q))) dst.Om[omp_at(dst.capacity, o, q, dst.omSym)] = omP[q];
    dst.OmN[om_at(dst.capacity, o, q)] = omN[q];
  }
  if (dst.St) {
    float st[16];
    if (src.St) {
#pragma unroll
      for (int q = 0; q < 16; ++q) st[q] = src.St[(size_t)i * 16 + q];
    } else {          // default Stats(): identity, eigenvalues 0, n 0 (stats.h:21-27)
#pragma unroll
      for (int q = 0; q < 16; ++q) st[q] = 0.f;
      st[0] = st[4] = st[8] = 1.f;
    }
    if (!identity) {  // StatsVector::transformInPlace: m * S (stats.h:125-131)
      Mat4 S = mat4_identity();
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) S(a,b) = st[a + 3 * b];
      S(0,3) = st[12]; S(1,3) = st[13]; S(2,3) = st[14];
      const Mat4 R = mat4_mul(m, S);
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) st[a + 3 * b] = R(a,b);
      st[12] = R(0,3); st[13] = R(1,3); st[14] = R(2,3);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) dst.St[(size_t)o * 16 + q] = st[q];
  }
  if (dsb.G && ssb.G && i < ngauss) {
    GaussD g = ssb.G[i]; int flags = ssb.Gf[i];
    if (!identity) gauss_transform(g, flags, m);
    dsb.G[o] = g; dsb.Gf[o] = flags;
  }
}
// Cloud::transformInPlace on the Gaussians and Stats of an existing cloud (k_cloud_transform handles the other arrays)
__global__ void __launch_bounds__(256) k_scene_transform(CloudDev cl, SceneBuffers sb, int n, int ngauss, Mat4 m) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < ngauss && sb.G) { GaussD g = sb.G[i]; int flags = sb.Gf[i]; gauss_transform(g, flags, m); sb.G[i] = g; sb.Gf[i] = flags; }
  if (i < n && cl.St) {
    float* st = cl.St + (size_t)i * 16;
    Mat4 S = mat4_identity();
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) S(a,b) = st[a + 3 * b];
    S(0,3) = st[12]; S(1,3) = st[13]; S(2,3) = st[14];
    const Mat4 R = mat4_mul(m, S);
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) st[a + 3 * b] = R(a,b);
    st[12] = R(0,3); st[13] = R(1,3); st[14] = R(2,3);
  }
}
// class-coded normal information -> 9 explicit planes (a cloud that becomes a scene)
__global__ void __launch_bounds__(256) k_expand_omega_n(CloudDev cl, float* __restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 nc = cl.Nc[i];
  const int cls = normal_class(nc.x, nc.y, nc.z, nc.w, cl.clsThr);
#pragma unroll
  for (int q = 0; q < 9; ++q) out[om_at(cl.capacity, i, q)] = (cls == 1) ? cl.omN[0][q] : ((cls == 2) ? cl.omN[1][q] : 0.f);
}

// ------------------------------------------------------------------------------------------------------------------
// Merger::merge (merger.cpp:15-119).
// 1. k_project_single: z-buffer of the whole cloud (index + depth image of merger.cpp:21-23).
// 2. k_merge_classify: the per-point tests of merger.cpp:42-76 -> _collapsedIndices; a point that merges into target t is pushed on
//    t's list (head/next, order of arrival).
// 3. k_merge_accumulate: one thread per target walks its list in ASCENDING point index -- the order in which the reference's
//    sequential loop calls addInformation, so the fp32 sums carry the same bits -- then mean() moves the point (merger.cpp:91-93).
// 4. exclusive scan of the keep flags + k_merge_compact: stable compaction of every per-point array (merger.cpp:88-104).
__global__ void __launch_bounds__(256) k_merge_classify(CloudDev cl, int n, Mat4 KRt, float minD, float maxD, float maxPointDepth, int rows, int cols,
                                                        const unsigned long long* __restrict__ z, unsigned tag, float distanceThreshold,
                                                        float normalThreshold, int* __restrict__ collapsed, int* __restrict__ head, int* __restrict__ next) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 p = load_xyz(cl.P3, i);
  int res = -1;
  // PinholePointProjector::_project (pinholepointprojector.h:224-233): x, y stay -1 when the depth is out of the projector's range
  const float ix = dot4seq(KRt(0,0), p.x, KRt(0,1), p.y, KRt(0,2), p.z, KRt(0,3), 1.0f);
  const float iy = dot4seq(KRt(1,0), p.x, KRt(1,1), p.y, KRt(1,2), p.z, KRt(1,3), 1.0f);
  const float d  = dot4seq(KRt(2,0), p.x, KRt(2,1), p.y, KRt(2,2), p.z, KRt(2,3), 1.0f);
  bool ok = !(d < minD || d > maxD);
  float fx = -1.f, fy = -1.f;
  if (ok) { const float inv = 1.0f / d; fx = roundf(ix * inv); fy = roundf(iy * inv); }
  // merger.cpp:49-54
  if (ok && !(d < 0 || d > maxPointDepth) && fx >= 0.f && fx < (float)cols && fy >= 0.f && fy < (float)rows) {
    const int x = (int)fx, y = (int)fy;
    const unsigned long long key = z[(size_t)y * cols + x];
    const int targetIndex = zkey_index(key, tag);
    if (targetIndex >= 0) {
      if (targetIndex == i) res = i;
      else {
        const float targetZ = zkey_depth(key, tag);
        const float4 cn = cl.Nc[i], tn = cl.Nc[targetIndex];
        if (fabsf(d - targetZ) < distanceThreshold && dot4seq(cn.x, tn.x, cn.y, tn.y, cn.z, tn.z, 0.f, 0.f) > normalThreshold) {
          res = targetIndex;
          next[i] = atomicExch(&head[targetIndex], i);
        }
      }
    }
  }
  collapsed[i] = res;
}
__global__ void __launch_bounds__(256) k_merge_accumulate(CloudDev cl, SceneBuffers sb, int n, const int* __restrict__ collapsed,
                                                          const int* __restrict__ head, const int* __restrict__ next) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n || collapsed[t] != t) return;
  GaussD g = sb.G[t]; int flags = sb.Gf[t];
  const int h = head[t];
  if (h >= 0) {
    int last = -1;
    for (;;) {                                    // next merged point in ascending index order
      int best = 0x7fffffff;
      for (int j = h; j >= 0; j = next[j]) if (j > last && j < best) best = j;
      if (best == 0x7fffffff) break;
      GaussD o = sb.G[best]; int of = sb.Gf[best];
      gauss_update_info(g, flags);                // Gaussian::addInformation (gaussian.h:47-53)
      if (!(of & kGaussInfo)) {                  // g.informationMatrix() caches the information form in the merged point's own Gaussian;
        gauss_update_info(o, of);                // it survives in the tail of the (never resized) Gaussian vector
        sb.G[best] = o; sb.Gf[best] = of;
      }
      for (int k = 0; k < 9; ++k) g.info[k] = g.info[k] + o.info[k];
      for (int k = 0; k < 3; ++k) g.infoVec[k] = g.infoVec[k] + o.infoVec[k];
      flags &= ~kGaussMoments;
      last = best;
    }
  }
  gauss_update_moments(g, flags);                 // gaussians()[i].mean()  (merger.cpp:92)
  sb.G[t] = g; sb.Gf[t] = flags;
  store_xyz(cl.P3, t, g.mean[0], g.mean[1], g.mean[2]);
}
__global__ void __launch_bounds__(256) k_merge_keep_flags(const int* __restrict__ collapsed, int n, int* __restrict__ keep) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) { const int c = collapsed[i]; keep[i] = (c < 0 || c == i) ? 1 : 0; }
}
// stable compaction: point i with keep[i] moves to offs[i] in the destination arrays
__global__ void __launch_bounds__(256) k_merge_compact(CloudDev src, SceneBuffers ssb, CloudDev dst, SceneBuffers dsb, int n, int ngauss,
                                                       const int* __restrict__ keep, const int* __restrict__ offs) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n || !keep[i]) return;
  const int o = offs[i];
  { const float4 p = load_xyz(src.P3, i); store_xyz(dst.P3, o, p.x, p.y, p.z); dst.Nc[o] = src.Nc[i]; }
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    if (!(dst.omSym && om_is_lower(q))) dst.Om[omp_at(dst.capacity, o, q, dst.omSym)] = src.Om[omp_at(src.capacity, i, q, src.omSym)];
    if (src.OmN && dst.OmN) dst.OmN[om_at(dst.capacity, o, q)] = src.OmN[om_at(src.capacity, i, q)];
  }
  if (src.St && dst.St) {
#pragma unroll
    for (int q = 0; q < 16; ++q) dst.St[(size_t)o * 16 + q] = src.St[(size_t)i * 16 + q];
  }
  if (ssb.G && dsb.G && i < ngauss) { dsb.G[o] = ssb.G[i]; dsb.Gf[o] = ssb.Gf[i]; }
}
// the reference does not resize the Gaussian vector after a merge (merger.cpp:108-112): entries [k, ngauss) keep their old values
__global__ void __launch_bounds__(256) k_gauss_copy_tail(SceneBuffers ssb, SceneBuffers dsb, int from, int to) {
  const int i = from + blockIdx.x * 256 + threadIdx.x;
  if (i < to) { dsb.G[i] = ssb.G[i]; dsb.Gf[i] = ssb.Gf[i]; }
}

// ------------------------------------------------------------------------------------------------------------------
// exclusive scan of n ints: 1024 per block, block totals scanned by one block, then added back
__global__ void __launch_bounds__(1024) k_scan_blocks(const int* __restrict__ in, int* __restrict__ out, int n, int* __restrict__ sums) {
  __shared__ int s[1024];
  const int i = blockIdx.x * 1024 + threadIdx.x;
  const int v = (i < n) ? in[i] : 0;
  s[threadIdx.x] = v;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int add = (threadIdx.x >= (unsigned)off) ? s[threadIdx.x - off] : 0;
    __syncthreads();
    s[threadIdx.x] += add;
    __syncthreads();
  }
  if (i < n) out[i] = s[threadIdx.x] - v;
  if (threadIdx.x == 1023) sums[blockIdx.x] = s[1023];
}
__global__ void __launch_bounds__(1024) k_scan_sums(int* __restrict__ sums, int nb, int* __restrict__ total) {
  __shared__ int s[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += 1024) {
    const int r = base + threadIdx.x;
    const int v = (r < nb) ? sums[r] : 0;
    s[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const int add = (threadIdx.x >= (unsigned)off) ? s[threadIdx.x - off] : 0;
      __syncthreads();
      s[threadIdx.x] += add;
      __syncthreads();
    }
    const int incl = s[threadIdx.x];
    const int c0 = carry;
    if (r < nb) sums[r] = c0 + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = c0 + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}
__global__ void __launch_bounds__(1024) k_scan_add(int* __restrict__ out, int n, const int* __restrict__ sums) {
  const int i = blockIdx.x * 1024 + threadIdx.x;
  if (i < n) out[i] += sums[blockIdx.x];
}
__global__ void k_set_count(int* count, const int* total) { *count = *total; }

// ------------------------------------------------------------------------------------------------------------------
// VoxelCalculator::compute (voxelcalculator.cpp:15-73), canonical semantics (the intended lexicographic order of the voxel
// indices; the reference's IndexComparator, voxelcalculator.h:41-48, is not a strict weak ordering -- DESIGN.md): the FIRST point
// (lowest index) of every voxel survives; survivors come out sorted by (ix, iy, iz).
// Keys: the three truncated indices biased into 21 bits each -> one 63-bit word, so integer order == lexicographic order.
constexpr int kVoxelBits = 21, kVoxelBias = 1 << 20;
__device__ __forceinline__ bool voxel_key(const float4 p, float inverseResolution, unsigned long long& key) {
  const float fx = p.x * inverseResolution, fy = p.y * inverseResolution, fz = p.z * inverseResolution;
  if (!(fabsf(fx) < (float)kVoxelBias && fabsf(fy) < (float)kVoxelBias && fabsf(fz) < (float)kVoxelBias)) return false;
  const unsigned long long ix = (unsigned long long)((int)fx + kVoxelBias), iy = (unsigned long long)((int)fy + kVoxelBias),
                           iz = (unsigned long long)((int)fz + kVoxelBias);
  key = (ix << (2 * kVoxelBits)) | (iy << kVoxelBits) | iz;
  return true;
}
__device__ __forceinline__ unsigned long long voxel_hash(unsigned long long k) {      // splitmix64 finaliser
  k ^= k >> 30; k *= 0xbf58476d1ce4e5b9ull; k ^= k >> 27; k *= 0x94d049bb133111ebull; k ^= k >> 31; return k;
}
// open-addressing table of (key, lowest point index): slot claimed by atomicCAS on the key, index by atomicMin.  tableSize = power of 2.
__global__ void __launch_bounds__(256) k_voxel_insert(CloudDev cl, int n, float inverseResolution, unsigned long long* __restrict__ keys,
                                                      int* __restrict__ first, unsigned tableMask, int* __restrict__ slotOf, int* __restrict__ fault) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  unsigned long long key;
  if (!voxel_key(load_xyz(cl.P3, i), inverseResolution, key)) { atomicExch(fault, 1); slotOf[i] = -1; return; }
  unsigned slot = (unsigned)voxel_hash(key) & tableMask;
  for (unsigned probe = 0; probe <= tableMask; ++probe) {
    const unsigned long long prev = atomicCAS(&keys[slot], ~0ull, key);
    if (prev == ~0ull || prev == key) { atomicMin(&first[slot], i); slotOf[i] = (int)slot; return; }
    slot = (slot + 1) & tableMask;
  }
  atomicExch(fault, 2); slotOf[i] = -1;
}
__global__ void __launch_bounds__(256) k_voxel_survivors(int n, const int* __restrict__ slotOf, const int* __restrict__ first, int* __restrict__ keep) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) { const int s = slotOf[i]; keep[i] = (s >= 0 && first[s] == i) ? 1 : 0; }
}
// survivors -> (key, index) records in index order (stable compaction), sorted afterwards by key
__global__ void __launch_bounds__(256) k_voxel_records(CloudDev cl, int n, float inverseResolution, const int* __restrict__ keep, const int* __restrict__ offs,
                                                       unsigned long long* __restrict__ rkeys, int* __restrict__ ridx) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n || !keep[i]) return;
  unsigned long long key = 0; (void)voxel_key(load_xyz(cl.P3, i), inverseResolution, key);
  rkeys[offs[i]] = key; ridx[offs[i]] = i;
}
// LSD radix sort, 8 bits per pass, of m (key, index) records: histogram per block -> scan -> stable scatter.
// One block handles 2048 consecutive records; inside a block the scatter keeps input order (a single wave walks the chunk), so every
// pass is stable.  The scene sizes this runs on (<= 2^21 survivors) make 8 passes of this simple kernel cheap next to the merge.
constexpr int kSortChunk = 2048;
__global__ void __launch_bounds__(256) k_sort_hist(const unsigned long long* __restrict__ keys, int m, int shift, int* __restrict__ hist /*[256][nblocks]*/, int nblocks) {
  __shared__ int h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int base = blockIdx.x * kSortChunk;
  for (int j = threadIdx.x; j < kSortChunk; j += 256) { const int i = base + j; if (i < m) atomicAdd(&h[(int)((keys[i] >> shift) & 255ull)], 1); }
  __syncthreads();
  hist[threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}
__global__ void __launch_bounds__(64) k_sort_scatter(const unsigned long long* __restrict__ keys, const int* __restrict__ idx, int m, int shift,
                                                     const int* __restrict__ offs /*[256][nblocks] exclusive*/, int nblocks,
                                                     unsigned long long* __restrict__ okeys, int* __restrict__ oidx) {
  __shared__ int pos[256];
  for (int d = threadIdx.x; d < 256; d += 64) pos[d] = offs[d * nblocks + blockIdx.x];
  __syncthreads();
  const int base = blockIdx.x * kSortChunk;
  const int lane = threadIdx.x;
  for (int j0 = 0; j0 < kSortChunk; j0 += 64) {
    const int i = base + j0 + lane;
    const bool in = i < m;
    const unsigned long long key = in ? keys[i] : 0ull;
    const int digit = (int)((key >> shift) & 255ull);
    // rank of this lane among the lanes of the wave with the same digit (lower lanes first): stable
    unsigned long long same = __ballot(in);
#pragma unroll
    for (int b = 0; b < 8; ++b) { const unsigned long long bal = __ballot((digit >> b) & 1); same &= ((digit >> b) & 1) ? bal : ~bal; }
    const int rank = __popcll(same & ((1ull << lane) - 1ull));
    if (in) { const int o = pos[digit] + rank; okeys[o] = key; oidx[o] = idx[i]; }
    __syncthreads();
    if (in && rank == __popcll(same) - 1) pos[digit] += __popcll(same);      // the last lane of each digit group advances the cursor
    __syncthreads();
  }
}
// gather the survivors' arrays in sorted order
__global__ void __launch_bounds__(256) k_voxel_gather(CloudDev src, SceneBuffers ssb, CloudDev dst, SceneBuffers dsb, int m, int withGauss,
                                                      const int* __restrict__ order) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= m) return;
  const int i = order[o];
  { const float4 p = load_xyz(src.P3, i); store_xyz(dst.P3, o, p.x, p.y, p.z); dst.Nc[o] = src.Nc[i]; }
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    if (!(dst.omSym && om_is_lower(q))) dst.Om[omp_at(dst.capacity, o, q, dst.omSym)] = src.Om[omp_at(src.capacity, i, q, src.omSym)];
    if (src.OmN && dst.OmN) dst.OmN[om_at(dst.capacity, o, q)] = src.OmN[om_at(src.capacity, i, q)];
  }
  if (src.St && dst.St) {
#pragma unroll
    for (int q = 0; q < 16; ++q) dst.St[(size_t)o * 16 + q] = src.St[(size_t)i * 16 + q];
  }
  if (withGauss) { dsb.G[o] = ssb.G[i]; dsb.Gf[o] = ssb.Gf[i]; }
}

}  // namespace pwnhip
