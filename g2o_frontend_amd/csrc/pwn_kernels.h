// pwn_kernels.h -- hand-written gfx950 kernels of the PWN dense-registration path.
//
// One kernel per stage of the reference's CPU path (g2o_frontend/pwn_core/, cited per kernel).
// All kernels are batched: blockIdx.y selects the frame / pair, whose buffers are described by a
// FrameDesc / PairDesc record in device memory.  Layout in HBM is SoA:
//   cloud:  P3[3*i .. 3*i+2] = x, y, z                     (Point: 12 bytes -- the projection reads nothing else)
//           Nc[i] = float4(nx, ny, nz, curvature)          (Normal + Stats::curvature(); the class of the normal information matrix --
//                   0 = zero, 1 = flat, 2 = non-flat -- is not stored: normal_class() derives it from this record and the threshold the
//                   cloud was converted with)
//           Om[((r*cap + i)*3 + c]                         (point information matrix: three planes of 12-byte rows, om_at(); nine dword
//                   planes until round 2: k_stats stores three 12-byte rows per point instead of nine dwords, -9 % of its time)
//           64 bytes per point; until round 2 the point carried the curvature and the normal a class word (68 bytes): the fused pass
//           fetches 8 bytes less per candidate (-7.5 % of its time), the projection 4 of 16 bytes less per point
//   images: row-major int32 / float32, lanes along image x (row-coalesced loads)
//   integral image: 10 planes [ch][rows][cols] (x y z n xx xy xz yy yz zz)
//   z-buffer of the aligner: uint32 per pixel = epoch tag (11 b) | point index (21 b), empty = ~0 (kZ32Tag0);
//   z-buffer of the stand-alone projection / Merger: uint64 = epoch tag (8 b) | float_bits(depth) (31 b) | point index (25 b)
// Arithmetic follows the reference's evaluation order (left-to-right inner products, no FMA:
// compiled with -ffp-contract=off) so integer outputs are bit-exact and fp32 outputs differ from
// the CPU path only through libm-vs-ocml trig and summation order of the H/b reduction.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "pwn_math.h"

namespace pwnhip {

constexpr int kMaxIter = 64;
constexpr int kIntegralChannels = 10;
constexpr int kAccN = 37;              // Htt9 Htr9 Hrr9 bt3 br3 chi2 inliers C K
constexpr int kPixPerThread = 8;
constexpr int kAlignBlock = 256;
constexpr unsigned long long kZEmpty = ~0ull;
// 64-bit z-buffer word (stand-alone projection, Merger::merge: the scene stage) = epoch tag (8 bits) | depth bits (31, depth >= 0) |
// point index (25 bits).  Call number j uses tag kZTag0 - j: a smaller tag wins atomicMin, so words left by earlier projections behave
// as "empty" and the buffer is cleared only when the tag space is used up (take_tags).  The reference's clouds have no size limit; a scene
// grows by a frame per step (pwn_aligner.cpp:205-208: 1.2 M points per 1280x960 frame), so the index field is 25 bits wide here -- until
// round 4 it was 21 (tag 12), the width the ALIGNER's 32-bit word still has (kZIndexBits below).
constexpr unsigned kZTag0 = 0xFEu;
constexpr int kZ64IndexBits = 25, kZ64TagShift = 56;
constexpr int kMaxCloudPoints = 1 << kZ64IndexBits;      // 33 554 432 points per cloud
constexpr int kZIndexBits = 21;                           // index field of the aligner's 32-bit word (kZ32Tag0)
constexpr int kMaxAlignerPoints = 1 << kZIndexBits;       // 2 097 152: what a cloud handed to Aligner::align may hold
__host__ __device__ __forceinline__ unsigned long long zkey(unsigned tag, float d, int i) {
  return ((unsigned long long)tag << kZ64TagShift) | ((unsigned long long)(__builtin_bit_cast(unsigned, d) & 0x7fffffffu) << kZ64IndexBits) | (unsigned)i;
}
__host__ __device__ __forceinline__ int zkey_index(unsigned long long k, unsigned tag) {
  return ((unsigned)(k >> kZ64TagShift) == tag) ? (int)(k & ((1u << kZ64IndexBits) - 1u)) : -1;
}
__host__ __device__ __forceinline__ float zkey_depth(unsigned long long k, unsigned tag) {
  return ((unsigned)(k >> kZ64TagShift) == tag) ? __builtin_bit_cast(float, (unsigned)((k >> kZ64IndexBits) & 0x7fffffffu)) : FLT_MAX;
}

// Z-buffer of the ALIGNER (round 2): 32 bits per pixel = epoch tag (11 b) | point index (21 b), empty = ~0.  The depth is not stored:
// a pixel's first point under the current tag arrives by one atomicMin (a smaller tag beats whatever older projections left, like
// in the 64-bit form); only when two points of the SAME projection meet in a pixel -- a few per cent -- does the later one compare
// depths (it recomputes the other's from the cloud) and install the nearer one, ties to the lower index, with a compare-and-swap loop
// (project_point32).  Same winner as the reference's sequential `>` test (pinholepointprojector.cpp:61), half the atomic bytes (the
// projection runs at the memory-side atomic rate: tools/micro/zbuf_atomics.hip, 78 us against 122 us for the bare scatter), half the
// z-buffer bytes in the fused pass.  Depth images (matchClouds score, CorrespondenceFinder::*DepthImage()) are recomputed from the
// winning point with the projection's own matrix: the same expression, the same bits.
constexpr unsigned kZ32Tag0 = 0x7FEu;       // 0x7FF is the tag of the empty word
constexpr unsigned kZ32IndexMask = (1u << kZIndexBits) - 1u;
__host__ __device__ __forceinline__ unsigned z32key(unsigned tag, int i) { return (tag << kZIndexBits) | (unsigned)i; }
__host__ __device__ __forceinline__ int z32_index(unsigned w, unsigned tag) { return ((w >> kZIndexBits) == tag) ? (int)(w & kZ32IndexMask) : -1; }

constexpr int kClsMask = 3;      // class of the normal information matrix (0 zero, 1 flat, 2 non-flat) as a word in the old-format view (cloud_get)

struct CloudDev {
  float*  P3;        // [capacity][3] x y z
  float4* Nc;        // [capacity] (nx, ny, nz, curvature)
  float*  Om;        // [3][capacity][3]: row r of point i at (r * capacity + i) * 3 (om_at)
  float*  OmN;       // optional, same layout: full normal information matrices (uploaded clouds, scenes), else nullptr
  float*  St;        // optional [capacity][16] stats: U(9, column-major) eigenvalues(3) mean(3) n(1)
  int*    count;
  int     capacity;
  float   omN[2][9]; // class matrices (row-major 3x3): [0] flat, [1] non-flat
  float   clsThr;    // NormalInformationMatrixCalculator::_curvatureThreshold the cloud was converted with (informationmatrixcalculator.cpp:47-52)
  int     omSym;     // storage of Om: 0 = exact9 (three planes of 12-byte rows, all nine entries as computed), 1 = sym6 (two planes:
                     // (xx xy xz) and (yy yz zz), the upper triangle as computed; readers mirror it).  OmN is always nine entries.
};
// NormalInformationMatrixCalculator::compute (informationmatrixcalculator.cpp:38-58): zero normal -> zero matrix, else flat / non-flat by
// the curvature.  The converter decides on the normal before the sensor offset is applied; a rotation does not turn a non-zero normal
// into the zero vector, and a candidate with a zero normal is rejected before its class is looked at (correspondencefinder.cpp:69).
// information matrices (Om, OmN): entry k = 3 r + c of point i
__host__ __device__ __forceinline__ size_t om_at(size_t cap, size_t i, int k) { return ((size_t)(k / 3) * cap + i) * 3 + (size_t)(k % 3); }
// sym6 storage of the POINT information matrix (PWN_HIP_OMEGA_SYM6): U diag U^t is symmetric up to the rounding of its nine separately
// evaluated entries (informationmatrixcalculator.cpp:26-30), and SURVEY.md 8(d) counts it as 24 bytes.  Entry k = 3 r + c lives in slot
// sym_slot(k) of the six stored values (row-major upper triangle), i.e. at om_at(cap, i, sym_slot(k)): slots 0-2 are plane 0, 3-5 plane 1.
// Writers store the upper triangle only (om_is_lower entries are dropped), readers get the mirrored value.
__host__ __device__ constexpr int sym_slot(int k) { return k == 0 ? 0 : (k == 1 || k == 3) ? 1 : (k == 2 || k == 6) ? 2 : k == 4 ? 3 : (k == 5 || k == 7) ? 4 : 5; }
__host__ __device__ constexpr bool om_is_lower(int k) { return k == 3 || k == 6 || k == 7; }
__host__ __device__ __forceinline__ size_t omp_at(size_t cap, size_t i, int k, int sym) { return om_at(cap, i, sym ? sym_slot(k) : k); }
__host__ __device__ __forceinline__ int om_planes(int sym) { return sym ? 2 : 3; }
__host__ __device__ __forceinline__ int normal_class(float nx, float ny, float nz, float curvature, float thr) {
  return (nx != 0.f || ny != 0.f || nz != 0.f) ? ((curvature < thr) ? 1 : 2) : 0;
}

struct FrameDesc {
  const float* depth;        // float32 metres, or nullptr when `raw` is used
  const uint16_t* raw;       // optional uint16 source (DepthImage_convert_16UC1_to_32FC1 fused in: d = raw ? scale*raw : 0)
  float raw_scale;
  int* index;
  int* interval;
  float* integral;   // [10][rows*cols]
  int* rowoff;       // [rows] (stand-alone unProject) or [rows][strips] (converter fast path)
  unsigned long long* carry; // [strips][bands][160] strip-to-strip hand-over words of k_unproject_integral
  CloudDev cloud;
  int* count_out;    // optional: page-locked host word that receives the point count as well (single-frame calls: no gather kernel, no copy back)
};

struct ConvertParams {
  int rows, cols;
  Mat4 iKRt;
  float ivx, ivy;            // K*(R,R,0): pixels-per-metre of the world radius at unit depth
  float minD, maxD;
  int minRadius, maxRadius, minPoints;
  float statsCurvThr, pointInfoCurvThr, normalInfoCurvThr;
  float pFlat[3], pNonFlat[3];
  int hasOffset;
  Mat4 offset;               // sensor offset with last row forced
  int keepStats;
  int spinLimit;             // polls of a strip hand-over word before a waiting lane gives up and raises the fault flag (kSpinLimit)
  int dbgWithhold;           // test hook (pwn_hip_debug_withhold_carry): index of one hand-over word that is NOT written, -1 = none
  int omSym;                 // storage of the clouds' point information matrices (CloudDev::omSym): 1 = k_stats stores the upper triangle as two 12-byte rows
  int lean;                  // 1 = the front end (k_unproject_integral*) stores neither the points nor the interval image and k_stats recomputes
                             // both from the depth (the same expressions, the same bits): 20 bytes per pixel less written and 18 less read back
};

struct PairState {
  Mat4 T;          // Aligner::_T
  Mat4 invTcorr;   // _T.inverse() handed to CorrespondenceFinder::compute
  Mat4 invTcorrPrev; // the one of the last executed outer iteration (the finder's correspondences that _computeStatistics reuses)
  Mat4 invT;       // Linearizer::_T
  Mat4 KRt;        // projector matrix of the next reference projection
  Mat4 KRtLast;    // projector matrix of the reference projection executed last (its depth image is recomputed with it)
  Mat4 KRtCur;     // projector matrix of the current-cloud projection
  int   it;
  int   pad[3];
  float chi2[kMaxIter];
  int   inliers[kMaxIter];
  int   ncorr[kMaxIter];
  int   ncand[kMaxIter];
};

struct PairDesc {
  CloudDev ref, cur;
  unsigned* zref;          // 32-bit z-buffers (tag | index), one word per pixel
  unsigned* zcur;
  int* curidx;             // index image of the current cloud (resolved once from zcur, or the converter's own image)
  const int* refidx0;      // optional: index image of the reference cloud for the FIRST outer iteration (the converter's own image when the
                           // initial guess is the identity: that projection returns it), else nullptr
  double* partials;        // [nblocks][kAccN]
  PairState* state;
  PairState* state_out;    // optional: page-locked host copy that k_solve_update keeps up to date in what the host reads back (T, it, the traces)
  int* fault;              // page-locked host word of the call: k_project stores 1 when a pixel's settle loop gave up (z32_settle)
  unsigned* zdepth;        // depth image of the two-pass projection (k_project_robust), nullptr unless the call runs in that mode
};

struct AlignParams {
  int rows, cols;
  Mat3 K;
  Mat4 refOffset;
  float minD, maxD;
  float sqDist, normalThr, flatThr, minRatio, maxRatio;
  float maxChi2;
  int robust;
  int settleGuard;         // rounds of z32_settle's compare-and-swap loop before a thread gives up (kSettleGuard)
};

// ------------------------------------------------------------------------------------------------------------------
// small device helpers
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// Butterfly sum of the 64 lanes in the order xor 1, 2, 4, 8, 16, 32 with DPP operands instead of ds_bpermute (no LDS crossbar, no
// lgkmcnt waits): after each step all lanes of a group hold the group's sum, so quad_perm / row_half_mirror / row_mirror bring
// "the other half's sum" exactly like an xor shuffle would, and row_bcast15 / row_bcast31 add the neighbouring rows' sums
// (a + b == b + a bitwise): the fixed tree ((..(l0+l1)+(l2+l3)..)) of a wave, so results stay bitwise reproducible.
// The total is valid in lane 63 only.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum_dpp_lane63(float v) {
  v += dpp_f32<0xB1, 0xF>(v);       // quad_perm [1,0,3,2]  (xor 1)
  v += dpp_f32<0x4E, 0xF>(v);       // quad_perm [2,3,0,1]  (xor 2)
  v += dpp_f32<0x141, 0xF>(v);      // row_half_mirror      (the other quad of the 8)
  v += dpp_f32<0x140, 0xF>(v);      // row_mirror           (the other 8 of the row)
  v += dpp_f32<0x142, 0xA>(v);      // row_bcast15 -> rows 1 and 3
  v += dpp_f32<0x143, 0xC>(v);      // row_bcast31 -> rows 2 and 3
  return v;
}
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// Pointers that reach a kernel inside a descriptor struct are generic to the compiler (flat_* instructions, 64-bit VALU address
// arithmetic, and flat traffic ties the LDS counter to the vector-memory one).  They all point to hipMalloc'ed memory: as_global()
// says so, which gives global_* instructions with a scalar base + 32-bit lane offset.
template <typename T> using gptr = T __attribute__((address_space(1)))*;
template <typename T> __device__ __forceinline__ gptr<T> as_global(T* p) { return (gptr<T>)(uintptr_t)p; }
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load4(gptr<const v4f> p) { const v4f v = *p; return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void store4(gptr<v4f> p, const float4 v) { v4f t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; *p = t; }
// three packed floats: a clang ext_vector_type(3) has sizeof = alignof = 16, but the xyz records and the information-matrix rows sit at
// 12-byte strides -- the access type therefore carries alignment 4 (still one global_load/store_dwordx3; an `align 16` access at a 4-byte
// aligned address would be undefined behaviour and licenses the compiler to widen it to 16 bytes, over the next record;
// tests/test_capi_cpu.py disassembles the library and checks that k_stats / k_unproject_integral contain no dwordx4 store)
typedef float v3f_raw __attribute__((ext_vector_type(3)));
typedef v3f_raw v3f __attribute__((aligned(4)));
// point i of a packed xyz array as (x, y, z, 0): one 12-byte load
__device__ __forceinline__ float4 load_xyz(gptr<const float> p3, unsigned i) { const v3f_raw v = *(gptr<const v3f>)(p3 + 3u * i); return make_float4(v.x, v.y, v.z, 0.f); }
__device__ __forceinline__ float4 load_xyz(const float* p3, int i) { return load_xyz(as_global(p3), (unsigned)i); }
__device__ __forceinline__ void store_xyz(float* p3, int i, float x, float y, float z) { v3f_raw v; v.x = x; v.y = y; v.z = z; *(gptr<v3f>)(as_global(p3) + 3u * (unsigned)i) = v; }
// Old-format view of point i for the code that is not hot (scene stage, stand-alone stages): P = (x, y, z, curvature),
// Nm = (nx, ny, nz, class word); cloud_put stores the same view back (the class word is derived data and is dropped).
__device__ __forceinline__ void cloud_get(const CloudDev& c, int i, float4& P, float4& Nm) {
  P = load_xyz(c.P3, i);
  const float4 nc = c.Nc[i];
  P.w = nc.w;
  Nm = make_float4(nc.x, nc.y, nc.z, __int_as_float(normal_class(nc.x, nc.y, nc.z, nc.w, c.clsThr)));
}
__device__ __forceinline__ void cloud_put(const CloudDev& c, int i, const float4 P, const float4 Nm) {
  store_xyz(c.P3, i, P.x, P.y, P.z);
  c.Nc[i] = make_float4(Nm.x, Nm.y, Nm.z, P.w);
}
// force a wave-uniform value into a scalar register
__device__ __forceinline__ float uniform(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ Mat4 uniform_iso(const Mat4& T) {
  Mat4 r;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    r(0,c) = uniform(T(0,c)); r(1,c) = uniform(T(1,c)); r(2,c) = uniform(T(2,c));
  }
  r(3,0) = 0.f; r(3,1) = 0.f; r(3,2) = 0.f; r(3,3) = 1.f;
  return r;
}

__device__ __forceinline__ Mat4 uniform_iso_global(gptr<const float> m) {      // column-major 4x4 in global memory -> SGPRs
  Mat4 r;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    r(0,c) = uniform(m[4 * c]); r(1,c) = uniform(m[4 * c + 1]); r(2,c) = uniform(m[4 * c + 2]);
  }
  r(3,0) = 0.f; r(3,1) = 0.f; r(3,2) = 0.f; r(3,3) = 1.f;
  return r;
}

// ------------------------------------------------------------------------------------------------------------------
// DepthImage_convert_16UC1_to_32FC1 (pwn_core/pwn_static.cpp:54-68).  grid.y = frame
struct RawDesc { const uint16_t* src; float* dst; };
__global__ void k_u16_to_f32(const RawDesc* __restrict__ d, int n, float scale) {
  const RawDesc rd = d[blockIdx.y];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint16_t s = rd.src[i];
    rd.dst[i] = s ? scale * (float)s : 0.0f;
  }
}
// DepthImage_convert_32FC1_to_16UC1 (pwn_core/pwn_static.cpp:38-52)
__global__ void k_f32_to_u16(const float* __restrict__ src, uint16_t* __restrict__ dst, int n, float scale) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float f = src[i];
    dst[i] = (f < FLT_MAX) ? (uint16_t)(scale * f) : (uint16_t)0;
  }
}
// DepthImage_scale (pwn_core/pwn_static.cpp:5-36): one thread per destination pixel
__global__ void k_depth_scale(const float* __restrict__ src, int srows, int scols, int step, float maxCov, float* __restrict__ dst) {
  const int rows = srows / step, cols = scols / step;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const int r = i / cols, c = i % cols;
  float acc = 0, acc2 = 0; int np = 0;
  const int sr = r * step, sc = c * step;
  for (int a = 0; a < step; ++a)
    for (int b = 0; b < step; ++b)
      if (sr + a < srows && sc + b < scols) {
        const float f = src[(size_t)(sr + a) * scols + sc + b];
        acc += f; acc2 += f * f; np += f > 0;
      }
  float out = 0.f;
  if (np) {
    const float mu = acc / np;
    const float sigma = acc2 / np - mu * mu;
    if (!(sigma > maxCov)) out = mu;
  }
  dst[i] = out;
}

// depth of pixel i of a frame: the float image, or the raw uint16 image converted on the fly (pwn_static.cpp:54-68)
__device__ __forceinline__ float frame_depth(const FrameDesc& f, size_t i) {
  if (f.raw) { const uint16_t s = f.raw[i]; return s ? f.raw_scale * (float)s : 0.0f; }
  return f.depth[i];
}
// ------------------------------------------------------------------------------------------------------------------
// Ordered compaction, step 1: valid pixels per image row.  grid = (rows, frames), block = 256.
// validity test = PinholePointProjector::_unProject (pwn_core/pinholepointprojector.h:246-248)
__global__ void __launch_bounds__(256) k_row_count(const FrameDesc* __restrict__ frames, ConvertParams cp) {
  const FrameDesc& f = frames[blockIdx.y];
  const int r = blockIdx.x;
  int cnt = 0;
  for (int c = threadIdx.x; c < cp.cols; c += 256) {
    const float d = frame_depth(f, (size_t)r * cp.cols + c);
    cnt += !(d < cp.minD || d > cp.maxD);
  }
  __shared__ int s[4];
  float t = wave_sum((float)cnt);           // <= 64*N small ints: exact in fp32
  if (lane_id() == 0) s[threadIdx.x >> 6] = (int)t;
  __syncthreads();
  if (threadIdx.x == 0) f.rowoff[r] = s[0] + s[1] + s[2] + s[3];
}
// step 2: exclusive scan of the row counts (in place) + total.  grid = frames, block = 1024
__global__ void __launch_bounds__(1024) k_row_offsets(const FrameDesc* __restrict__ frames, int rows) {
  const FrameDesc& f = frames[blockIdx.x];
  __shared__ int s[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < rows; base += 1024) {
    const int r = base + threadIdx.x;
    const int v = (r < rows) ? f.rowoff[r] : 0;
    s[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const int add = (threadIdx.x >= off) ? s[threadIdx.x - off] : 0;
      __syncthreads();
      s[threadIdx.x] += add;
      __syncthreads();
    }
    const int incl = s[threadIdx.x];
    const int c0 = carry;
    if (r < rows) f.rowoff[r] = c0 + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = c0 + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) { *f.cloud.count = carry; if (f.count_out) *f.count_out = carry; }
}
// per-frame point counts -> one contiguous array (single D2H copy per batch)
__global__ void k_gather_counts(const FrameDesc* __restrict__ frames, int n, int* __restrict__ out, const int* __restrict__ fault) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = *frames[i].cloud.count;
  if (i == n) out[n] = *fault;      // time-out flag of k_unproject_integral rides along with the counts
}
// step 3: PinholePointProjector::unProject (pwn_core/pinholepointprojector.cpp:93-133) + projectIntervals (:135-147).
// Point index = row-major rank of the valid pixel.  grid = (rows, frames), block = 256.
__global__ void __launch_bounds__(256) k_unproject(const FrameDesc* __restrict__ frames, ConvertParams cp) {
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
    if (in) {
      const size_t pix = (size_t)r * cp.cols + c;
      int idx = -1, itv = -1;
      if (valid) {
        idx = base + woff + rank;
        // _unProject: p = iKRt * (x*d, y*d, d, 1)
        const float a = (float)c * d, b = (float)r * d;
        float4 p;
        p.x = dot4seq(cp.iKRt(0,0), a, cp.iKRt(0,1), b, cp.iKRt(0,2), d, cp.iKRt(0,3), 1.0f);
        p.y = dot4seq(cp.iKRt(1,0), a, cp.iKRt(1,1), b, cp.iKRt(1,2), d, cp.iKRt(1,3), 1.0f);
        p.z = dot4seq(cp.iKRt(2,0), a, cp.iKRt(2,1), b, cp.iKRt(2,2), d, cp.iKRt(2,3), 1.0f);
        p.w = 0.f;
        if (f.cloud.P3 && idx < f.cloud.capacity) store_xyz(f.cloud.P3, idx, p.x, p.y, p.z);
        // _projectInterval: int(max(fx*R/d, fy*R/d))
        const float inv = 1.0f / d;
        const float px = cp.ivx * inv, py = cp.ivy * inv;
        itv = (px > py) ? (int)px : (int)py;
      }
      f.index[pix] = idx;
      f.interval[pix] = itv;
    }
    base += tot;
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------------
// PointIntegralImage::compute, passes 1+2 (pwn_core/pointintegralimage.cpp:16-35): scatter the points into the
// accumulators and prefix-sum along image x inside each image row.  fp32 addition is not associative and
// the covariance downstream differences these sums, so every chain keeps the reference's strictly sequential
// left-to-right order: one thread per (row, channel) chain, tiles staged through LDS so that all global
// traffic is row-coalesced.  grid = (ceil(rows/16), frames), block = 256.
// band height of the integral-image kernels (a multiple of 4: one row per compute wave and step); measured on MI355X: the hand-over
// wavefront of the strip kernel fills and drains in half the time with 8-row bands: -7 % against 16
constexpr int kIR_Rows = 8, kIR_Cols = 64, kIR_Stride = kIR_Cols + 1;
static_assert(kIR_Rows % 4 == 0 && (kIR_Rows & (kIR_Rows - 1)) == 0 && kIR_Rows <= 64, "band height");
__global__ void __launch_bounds__(256) k_integral_rows(const FrameDesc* __restrict__ frames, int rows, int cols) {
  const FrameDesc& f = frames[blockIdx.y];
  const int r0 = blockIdx.x * kIR_Rows;
  __shared__ float tile[kIntegralChannels * kIR_Rows * kIR_Stride];   // 41.6 KB
  const int tid = threadIdx.x;
  const size_t N = (size_t)rows * cols;
  // scan role: thread t < 160 owns chain (row = t % 16, ch = t / 16)
  const int srow = tid % kIR_Rows, sch = tid / kIR_Rows;
  float carry = 0.f;
  for (int x0 = 0; x0 < cols; x0 += kIR_Cols) {
    // phase 1: 1024 pixels, 4 per thread, lanes along x
#pragma unroll
    for (int j = 0; j < kIR_Rows / 4; ++j) {
      const int q = tid + 256 * j;
      const int lr = q / kIR_Cols, lc = q % kIR_Cols;
      const int r = r0 + lr, c = x0 + lc;
      float v[kIntegralChannels];
#pragma unroll
      for (int k = 0; k < kIntegralChannels; ++k) v[k] = 0.f;
      if (r < rows && c < cols) {
        const int idx = f.index[(size_t)r * cols + c];
        if (idx >= 0 && idx < f.cloud.capacity) {
          const float4 p = load_xyz(f.cloud.P3, idx);
          v[0] = p.x; v[1] = p.y; v[2] = p.z; v[3] = 1.0f;
          v[4] = p.x * p.x; v[5] = p.x * p.y; v[6] = p.x * p.z;
          v[7] = p.y * p.y; v[8] = p.y * p.z; v[9] = p.z * p.z;
        }
      }
#pragma unroll
      for (int k = 0; k < kIntegralChannels; ++k) tile[(k * kIR_Rows + lr) * kIR_Stride + lc] = v[k];
    }
    __syncthreads();
    // phase 2: sequential scan of 64 columns per chain
    if (tid < kIntegralChannels * kIR_Rows) {
      float* t = &tile[(sch * kIR_Rows + srow) * kIR_Stride];
      float vals[kIR_Cols];
#pragma unroll
      for (int c = 0; c < kIR_Cols; ++c) vals[c] = t[c];
#pragma unroll
      for (int c = 0; c < kIR_Cols; ++c) { carry = vals[c] + carry; vals[c] = carry; }
#pragma unroll
      for (int c = 0; c < kIR_Cols; ++c) t[c] = vals[c];
    }
    __syncthreads();
    // phase 3: coalesced write-back, plane by plane
#pragma unroll
    for (int j = 0; j < kIR_Rows / 4; ++j) {
      const int q = tid + 256 * j;
      const int lr = q / kIR_Cols, lc = q % kIR_Cols;
      const int r = r0 + lr, c = x0 + lc;
      if (r < rows && c < cols) {
#pragma unroll
        for (int k = 0; k < kIntegralChannels; ++k)
          f.integral[k * N + (size_t)r * cols + c] = tile[(k * kIR_Rows + lr) * kIR_Stride + lc];
      }
    }
    __syncthreads();
  }
}
// Converter latency path (a few frames: tracker, makeCloud): unProject + projectIntervals fused into the row scan.  Depth is read once;
// the kernel writes the points, the index and interval images AND the row-prefixed integral planes (k_integral_cols finishes them).
// One workgroup per 16-row x 64-column tile, so a VGA frame is 300 workgroups instead of 30 walking ten tiles each: unprojecting and
// storing a tile does not depend on its left neighbour, only the 160 (channel, row) chains do -- they continue from the neighbour's
// carry, handed over as in k_unproject_integral (one 64-bit word per chain = launch epoch << 32 | float bits, written once per launch;
// a tile waits only for a lower-numbered workgroup; bounded poll that raises *fault instead of hanging).  All tiles of a band sit on
// one XCD (workgroup ids are dealt round-robin over the 8 XCDs), so the hand-over words stay in that XCD's L2.  Wave w owns rows
// r0 + w + 4j (j = 0..3): one wave instruction covers the 64 columns of one row, the point index of a valid pixel = row offset +
// valid pixels of the row left of the tile (counted here) + popcount(ballot below the lane).
// grid = (8 * ceil(bands/8) * strips, frames), block = 256.
constexpr int kII_ChainsRows = kIntegralChannels * kIR_Rows;     // 160
__global__ void __launch_bounds__(256) k_unproject_integral_rows(const FrameDesc* __restrict__ frames, ConvertParams cp, unsigned epoch, int* __restrict__ fault) {
  const FrameDesc& f = frames[blockIdx.y];
  const int rows = cp.rows, cols = cp.cols;
  const int S = (cols + kIR_Cols - 1) / kIR_Cols, NB = (rows + kIR_Rows - 1) / kIR_Rows;
  const unsigned kk = blockIdx.x >> 3;
  const int band = 8 * (int)(kk / (unsigned)S) + (int)(blockIdx.x & 7u), s = (int)(kk % (unsigned)S);
  if (band >= NB) return;
  const int r0 = band * kIR_Rows, x0 = s * kIR_Cols;
  __shared__ float tile[kIntegralChannels * kIR_Rows * kIR_Stride];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t N = (size_t)rows * cols;
  // 1. unproject the tile
#pragma unroll
  for (int j = 0; j < kIR_Rows / 4; ++j) {
    const int lr = wave + 4 * j, lc = lane;
    const int r = r0 + lr, c = x0 + lc;
    const bool in = r < rows && c < cols;
    int base = 0;
    if (r < rows) {                                           // wave-uniform
      base = f.rowoff[r];
      for (int t = 0; t < s; ++t) {                           // full strips left of this one
        const float dl = frame_depth(f, (size_t)r * cols + t * kIR_Cols + lane);
        base += __popcll(__ballot(!(dl < cp.minD || dl > cp.maxD)));
      }
    }
    const float d = in ? frame_depth(f, (size_t)r * cols + c) : 0.f;
    const bool valid = in && !(d < cp.minD || d > cp.maxD);
    const unsigned long long bal = __ballot(valid);
    float v[kIntegralChannels];
#pragma unroll
    for (int k = 0; k < kIntegralChannels; ++k) v[k] = 0.f;
    if (in) {
      int idx = -1, itv = -1;
      if (valid) {
        idx = base + __popcll(bal & ((1ull << lane) - 1ull));
        const float a = (float)c * d, b = (float)r * d;
        float4 p;
        p.x = dot4seq(cp.iKRt(0,0), a, cp.iKRt(0,1), b, cp.iKRt(0,2), d, cp.iKRt(0,3), 1.0f);
        p.y = dot4seq(cp.iKRt(1,0), a, cp.iKRt(1,1), b, cp.iKRt(1,2), d, cp.iKRt(1,3), 1.0f);
        p.z = dot4seq(cp.iKRt(2,0), a, cp.iKRt(2,1), b, cp.iKRt(2,2), d, cp.iKRt(2,3), 1.0f);
        p.w = 0.f;
        if (idx < f.cloud.capacity) {
          if (!cp.lean) store_xyz(f.cloud.P3, idx, p.x, p.y, p.z);
          v[0] = p.x; v[1] = p.y; v[2] = p.z; v[3] = 1.0f;
          v[4] = p.x * p.x; v[5] = p.x * p.y; v[6] = p.x * p.z;
          v[7] = p.y * p.y; v[8] = p.y * p.z; v[9] = p.z * p.z;
        }
        if (!cp.lean) {
          const float inv = 1.0f / d;
          const float px = cp.ivx * inv, py = cp.ivy * inv;
          itv = (px > py) ? (int)px : (int)py;
        }
      }
      f.index[(size_t)r * cols + c] = idx;
      if (!cp.lean) f.interval[(size_t)r * cols + c] = itv;
    }
#pragma unroll
    for (int k = 0; k < kIntegralChannels; ++k) tile[(k * kIR_Rows + lr) * kIR_Stride + lc] = v[k];
  }
  __syncthreads();
  // 2. the 160 (channel, row) chains of the tile, continued from the tile to the left
  if (tid < kII_ChainsRows) {
    float carry = 0.f;
    if (s > 0) {
      const unsigned long long* src = f.carry + ((size_t)(s - 1) * NB + band) * kII_ChainsRows + tid;
      unsigned long long w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while ((unsigned)(w >> 32) != epoch) {
        if (++spins >= cp.spinLimit) { atomicOr(fault, 1); break; }    // a starved chain finishes with garbage instead of hanging the device
        __builtin_amdgcn_s_sleep(1);
        w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      carry = __uint_as_float((unsigned)w);
    }
    const int srow = tid % kIR_Rows, sch = tid / kIR_Rows;
    float* t = &tile[(sch * kIR_Rows + srow) * kIR_Stride];
#pragma unroll 1
    for (int c0 = 0; c0 < kIR_Cols; c0 += 16) {          // 16 LDS reads in flight, then the sequential adds
      float vals[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) vals[c] = t[c0 + c];
#pragma unroll
      for (int c = 0; c < 16; ++c) { carry = vals[c] + carry; vals[c] = carry; }
#pragma unroll
      for (int c = 0; c < 16; ++c) t[c0 + c] = vals[c];
    }
    if (s + 1 < S && (s * NB + band) * kII_ChainsRows + tid != cp.dbgWithhold)
      __hip_atomic_store(f.carry + ((size_t)s * NB + band) * kII_ChainsRows + tid,
                         ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(carry), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  // 3. the row-prefixed planes
#pragma unroll
  for (int j = 0; j < kIR_Rows / 4; ++j) {
    const int lr = wave + 4 * j, lc = lane;
    const int r = r0 + lr, c = x0 + lc;
    if (r < rows && c < cols) {
#pragma unroll
      for (int k = 0; k < kIntegralChannels; ++k)
        f.integral[k * N + (size_t)r * cols + c] = tile[(k * kIR_Rows + lr) * kIR_Stride + lc];
    }
  }
}
// ------------------------------------------------------------------------------------------------------------------
// Converter fast path, single pass: unProject + projectIntervals + PointIntegralImage::compute (all three passes of
// pwn_core/pointintegralimage.cpp:16-43) with ONE write of the integral planes and no read of them.
//
// A workgroup owns one 64-column strip of one frame and walks it top to bottom in bands of 16 rows.  Per band:
//   1. unproject the 16x64 pixels (as k_unproject; the point index comes from per-(row, strip) offsets, k_strip_count),
//      10 channel values per pixel into LDS;
//   2. x pass: 160 threads own the (channel, row) chains, continue them from the carry of the strip to the LEFT and hand the
//      carry on to the strip to the right;
//   3. y pass: 640 (channel, column) chains whose running sums live in registers from band to band; coalesced store.
// Every chain is the reference's strictly sequential fp32 sum, so the planes are bit-identical to the three-kernel path.
//
// The strip-to-strip hand-over is one 64-bit word per (strip, band, chain) = launch epoch << 32 | float bits: value and
// "ready" flag travel in the same single-copy-atomic store, so no fence is needed; a word is written once per launch (no
// slot reuse inside a launch, hence no back-pressure), stale words carry an older epoch.  Forward progress: strip s waits
// only for strip s-1 of the same frame, which has a smaller workgroup id and is therefore dispatched before it; the poll is
// bounded (kSpinLimit), a starved chain raises *fault and finishes with garbage instead of hanging the device.
// Workgroup -> (frame, strip) puts all strips of a frame on one XCD (ids are dealt round-robin over the 8 XCDs), so the
// hand-over words stay in that XCD's L2.  grid = 8 * ceil(frames/8) * strips, block = 256.
constexpr int kII_Chains = kII_ChainsRows;                   // 10 planes x band rows: the same hand-over word layout in both kernels that hand over
constexpr int kSpinLimit = 1 << 20;      // polls of >= 64 cycles + one L2 round trip each: ~1 s, against hand-over waits of microseconds
__host__ __device__ __forceinline__ int strips_of(int cols) { return (cols + kIR_Cols - 1) / kIR_Cols; }
__host__ __device__ __forceinline__ int bands_of(int rows) { return (rows + kIR_Rows - 1) / kIR_Rows; }

// valid pixels per (row, 64-column strip); k_row_offsets over rows*strips entries turns them into point-index offsets.
// One wave per image row, 16 bytes per lane and load (8 uint16 or 4 float pixels), so a 640-pixel row is two (five) wave loads instead
// of ten dependent 128-byte ones; the per-lane counts are summed over the 8 (16) lanes of a strip with xor shuffles (small integers:
// exact).  Needs rows that start 16-byte aligned (cols % 8 == 0 resp. cols % 4 == 0), else k_strip_count_any.
// grid = (ceil(rows / 4), frames), block = 256
template <bool RAW>
__global__ void __launch_bounds__(256) k_strip_count(const FrameDesc* __restrict__ frames, ConvertParams cp) {
  const FrameDesc& f = frames[blockIdx.y];
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const int r = blockIdx.x * 4 + wave, S = strips_of(cp.cols);
  if (r >= cp.rows) return;
  constexpr int PX = RAW ? 8 : 4;                      // pixels per lane and load
  constexpr int LPS = kIR_Cols / PX;                   // lanes per strip
  const float scale = f.raw_scale;
  for (int c0 = 0; c0 < cp.cols; c0 += 64 * PX) {
    const int c = c0 + lane * PX;
    int cnt = 0;
    if (c < cp.cols) {                                 // cols % PX == 0: a lane's pixels are all inside or all outside the row
      float d[PX];
      if (RAW) {
        const v4u w = *(gptr<const v4u>)(as_global(f.raw) + ((size_t)r * cp.cols + c));
        const unsigned u[4] = { w.x, w.y, w.z, w.w };
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned lo = u[k] & 0xFFFFu, hi = u[k] >> 16;
          d[2 * k] = lo ? scale * (float)lo : 0.0f; d[2 * k + 1] = hi ? scale * (float)hi : 0.0f;      // pwn_static.cpp:54-68
        }
      } else {
        const v4f w = *(gptr<const v4f>)(as_global(f.depth) + ((size_t)r * cp.cols + c));
        d[0] = w.x; d[1] = w.y; d[2] = w.z; d[3] = w.w;
      }
#pragma unroll
      for (int k = 0; k < PX; ++k) cnt += !(d[k] < cp.minD || d[k] > cp.maxD);
    }
#pragma unroll
    for (int off = 1; off < LPS; off <<= 1) cnt += __shfl_xor(cnt, off, 64);
    const int s = c / kIR_Cols;
    if ((lane & (LPS - 1)) == 0 && c < cp.cols) f.rowoff[r * S + s] = cnt;
  }
}
// any width / alignment.  grid = (rows, frames), block = 256
__global__ void __launch_bounds__(256) k_strip_count_any(const FrameDesc* __restrict__ frames, ConvertParams cp) {
  const FrameDesc& f = frames[blockIdx.y];
  const int r = blockIdx.x, S = strips_of(cp.cols);
  const int wave = threadIdx.x >> 6, lane = lane_id();
  for (int s = wave; s < S; s += 4) {
    const int c = s * kIR_Cols + lane;
    const float d = (c < cp.cols) ? frame_depth(f, (size_t)r * cp.cols + c) : 0.f;
    const bool valid = c < cp.cols && !(d < cp.minD || d > cp.maxD);
    const unsigned long long bal = __ballot(valid);
    if (lane == 0) f.rowoff[r * S + s] = __popcll(bal);
  }
}

// workgroup barrier that orders LDS traffic only: global stores stay in flight across it (a __syncthreads() would drain them)
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// Block = 4 compute waves + 1 loader wave.  gfx950 counts vector loads and stores in ONE in-order counter (vmcnt), so a wave that
// waits for a load also waits for every store it issued before it; the compute waves issue ~60 KB of stores per band and must not
// wait for memory at all.  The loader wave therefore does every global read of the block (depth and strip offsets of the NEXT
// band, hand-over words of the current one) and passes the values on through LDS; it never stores to global memory.
// Plane, index and interval stores are non-temporal (measured: strip kernel -5 %, k_stats after it -4 %).
constexpr int kII_Threads = 320;
// Body of the single-pass front end for strip s of frame f (planes [10][rows][cols], read back by k_stats).
__device__ __forceinline__ void unproject_integral_body(const FrameDesc& f, const ConvertParams& cp, const int s, const unsigned epoch, int* __restrict__ fault) {
  const int rows = cp.rows, cols = cp.cols;
  const int S = strips_of(cols), NB = bands_of(rows);
  __shared__ float tile[kIntegralChannels * kIR_Rows * kIR_Stride];
  __shared__ float stage[2][kIR_Rows * kIR_Cols];     // depth of the band (metres; 0 outside the image), double-buffered
  __shared__ int sbase[2][kIR_Rows];                  // point-index offset of (row, strip)
  __shared__ float cin[kII_Chains];                   // x-pass carries handed over by the strip to the left
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the role branches below are wave-uniform
  const size_t N = (size_t)rows * cols;
  const int x0 = s * kIR_Cols, c = x0 + lane;
  const bool loader = wave == 4;
  const bool chain = tid < kII_Chains;
  const gptr<const float> gdepth = as_global(f.depth);
  const gptr<const uint16_t> graw = as_global(f.raw);
  const bool is_raw = f.raw != nullptr;
  const float raw_scale = f.raw_scale;
  const gptr<const int> growoff = as_global((const int*)f.rowoff);
  const gptr<int> gindex = as_global(f.index), ginterval = as_global(f.interval);
  const gptr<float> gintegral = as_global(f.integral);
  float* const gP3 = f.cloud.P3;
  const gptr<unsigned long long> gcarry = as_global(f.carry);
  const int capacity = f.cloud.capacity;
  const bool lean = cp.lean != 0;
  float vcarry[3] = { 0.f, 0.f, 0.f };
  bool starved = false;

  // loader: request depth + offsets of band b (all loads unconditional and in flight together; pixels outside the image read
  // element 0 and are zeroed when staged), later convert and put them into stage[b & 1]
  unsigned bits[kIR_Rows]; int off = 0;
  auto request_band = [&](int b) {
    const int r0 = b * kIR_Rows;
    if (is_raw) {
#pragma unroll
      for (int i = 0; i < kIR_Rows; ++i) {
        const int r = r0 + i;
        const unsigned pix = (r < rows && c < cols) ? (unsigned)(r * cols + c) : 0u;
        bits[i] = graw[pix];
      }
    } else {
#pragma unroll
      for (int i = 0; i < kIR_Rows; ++i) {
        const int r = r0 + i;
        const unsigned pix = (r < rows && c < cols) ? (unsigned)(r * cols + c) : 0u;
        bits[i] = __float_as_uint(gdepth[pix]);
      }
    }
    const int rr = r0 + (lane & (kIR_Rows - 1));
    off = growoff[(rr < rows ? rr : 0) * S + s];
  };
  auto stage_band = [&](int b) {
    const int r0 = b * kIR_Rows;
#pragma unroll
    for (int i = 0; i < kIR_Rows; ++i) {
      float d;
      if (is_raw) d = bits[i] ? raw_scale * (float)bits[i] : 0.0f;      // DepthImage_convert_16UC1_to_32FC1 (pwn_static.cpp:54-68)
      else d = __uint_as_float(bits[i]);
      stage[b & 1][i * kIR_Cols + lane] = (r0 + i < rows && c < cols) ? d : 0.f;
    }
    sbase[b & 1][lane & (kIR_Rows - 1)] = off;       // 4 lanes write the same value
  };
  if (loader) { request_band(0); stage_band(0); }
  lds_barrier();

  for (int band = 0; band < NB; ++band) {
    const int r0 = band * kIR_Rows;
    if (loader) {
      const int nb = (band + 1 < NB) ? band + 1 : band;        // the last band re-reads itself (unconditional loads, result unused)
      request_band(nb);
      if (s > 0) {
        const gptr<unsigned long long> src = gcarry + ((size_t)(s - 1) * NB + band) * kII_Chains;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          const int k = lane + 64 * m;
          if (k < kII_Chains) {
            unsigned long long w = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while ((unsigned)(w >> 32) != epoch && !starved) {    // after one time-out this lane stops waiting: the launch is lost anyway
              if (++spins >= cp.spinLimit) { atomicOr(fault, 1); starved = true; break; }
              __builtin_amdgcn_s_sleep(1);
              w = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            cin[k] = __uint_as_float((unsigned)w);
          }
        }
      }
      stage_band(band + 1);
    } else {
      // 1. unproject: wave w owns rows r0 + w + 4j
#pragma unroll
      for (int jj = 0; jj < kIR_Rows / 4; ++jj) {
        const int lr = wave + 4 * jj;
        const int r = r0 + lr;
        const bool in = r < rows && c < cols;
        const float dd = stage[band & 1][lr * kIR_Cols + lane];
        const bool valid = in && !(dd < cp.minD || dd > cp.maxD);
        const unsigned long long bal = __ballot(valid);
        float v[kIntegralChannels];
#pragma unroll
        for (int k = 0; k < kIntegralChannels; ++k) v[k] = 0.f;
        if (in) {
          int idx = -1, itv = -1;
          if (valid) {
            idx = sbase[band & 1][lr] + __popcll(bal & ((1ull << lane) - 1ull));
            const float a = (float)c * dd, b = (float)r * dd;
            float4 p;
            p.x = dot4seq(cp.iKRt(0,0), a, cp.iKRt(0,1), b, cp.iKRt(0,2), dd, cp.iKRt(0,3), 1.0f);
            p.y = dot4seq(cp.iKRt(1,0), a, cp.iKRt(1,1), b, cp.iKRt(1,2), dd, cp.iKRt(1,3), 1.0f);
            p.z = dot4seq(cp.iKRt(2,0), a, cp.iKRt(2,1), b, cp.iKRt(2,2), dd, cp.iKRt(2,3), 1.0f);
            p.w = 0.f;
            if (idx < capacity) {
              if (!lean) store_xyz(gP3, idx, p.x, p.y, p.z);
              v[0] = p.x; v[1] = p.y; v[2] = p.z; v[3] = 1.0f;
              v[4] = p.x * p.x; v[5] = p.x * p.y; v[6] = p.x * p.z;
              v[7] = p.y * p.y; v[8] = p.y * p.z; v[9] = p.z * p.z;
            }
            if (!lean) {
              const float inv = 1.0f / dd;
              const float px = cp.ivx * inv, py = cp.ivy * inv;
              itv = (px > py) ? (int)px : (int)py;
            }
          }
          __builtin_nontemporal_store(idx, gindex + (unsigned)(r * cols + c));
          if (!lean) __builtin_nontemporal_store(itv, ginterval + (unsigned)(r * cols + c));
        }
#pragma unroll
        for (int k = 0; k < kIntegralChannels; ++k) tile[(k * kIR_Rows + lr) * kIR_Stride + lane] = v[k];
      }
    }
    lds_barrier();
    // 2. x pass, chain (channel = tid / 16, row = tid % 16)
    if (chain) {
      float carry = (s > 0) ? cin[tid] : 0.f;
      float* t = &tile[tid * kIR_Stride];
#pragma unroll 1
      for (int c0 = 0; c0 < kIR_Cols; c0 += 16) {
        float vals[16];
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) vals[cc] = t[c0 + cc];
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) { carry = vals[cc] + carry; vals[cc] = carry; }
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) t[c0 + cc] = vals[cc];
      }
      if (s + 1 < S && (s * NB + band) * kII_Chains + tid != cp.dbgWithhold)
        __hip_atomic_store(gcarry + ((size_t)s * NB + band) * kII_Chains + tid,
                           ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(carry),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    lds_barrier();
    // 3. y pass, chain q = (channel = q / 64, column = q % 64)
    if (!loader) {
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) {
        const int q = tid + 256 * jj;
        if (q < kIntegralChannels * kIR_Cols) {
          const int ch = q >> 6;
          float vals[kIR_Rows];
#pragma unroll
          for (int r = 0; r < kIR_Rows; ++r) vals[r] = tile[(ch * kIR_Rows + r) * kIR_Stride + lane];
          float vc = vcarry[jj];
#pragma unroll
          for (int r = 0; r < kIR_Rows; ++r) { vc = vals[r] + vc; vals[r] = vc; }
          vcarry[jj] = vc;
          if (c < cols) {
            const gptr<float> dst = gintegral + ((size_t)ch * N + (size_t)r0 * cols + c);
#pragma unroll
            for (int r = 0; r < kIR_Rows; ++r) if (r0 + r < rows) __builtin_nontemporal_store(vals[r], dst + (unsigned)(r * cols));
          }
        }
      }
    }
    lds_barrier();
  }
}
__global__ void __launch_bounds__(kII_Threads) k_unproject_integral(const FrameDesc* __restrict__ frames, ConvertParams cp, int nframes,
                                                                    unsigned epoch, int* __restrict__ fault) {
  const int S = strips_of(cp.cols);
  const unsigned j = blockIdx.x >> 3;
  const int fi = 8 * (int)(j / (unsigned)S) + (int)(blockIdx.x & 7u), s = (int)(j % (unsigned)S);
  if (fi >= nframes) return;
  unproject_integral_body(frames[fi], cp, s, epoch, fault);
}
// pass 3 (pwn_core/pointintegralimage.cpp:38-43): prefix-sum along image y inside each image column, sequential.
// one thread per (column, channel) chain, lanes along x.  grid = (ceil(cols/64), 10, frames), block = 64: this kernel only runs on the
// latency path (a few frames), where one-wave workgroups spread a frame's 6400 chains over 100 CUs instead of 30.
constexpr int kIC_Block = 64;
// fault / fault_out: the time-out flag of k_unproject_integral_rows (the launch before this one, complete by now) forwarded to a page-locked
// host word, so that the host reads it without a copy of its own (fault_out may be nullptr).
__global__ void __launch_bounds__(kIC_Block) k_integral_cols(const FrameDesc* __restrict__ frames, int rows, int cols, const int* __restrict__ fault,
                                                             int* __restrict__ fault_out) {
  if (fault_out && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *fault_out = *fault;
  const FrameDesc& f = frames[blockIdx.z];
  const int c = blockIdx.x * kIC_Block + threadIdx.x;
  if (c >= cols) return;
  float* p = f.integral + (size_t)blockIdx.y * rows * cols + c;
  float carry = 0.f;
  int r = 0;
  constexpr int U = 16;
  for (; r + U <= rows; r += U) {
    float v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = p[(size_t)(r + k) * cols];
#pragma unroll
    for (int k = 0; k < U; ++k) { carry = v[k] + carry; v[k] = carry; }
#pragma unroll
    for (int k = 0; k < U; ++k) p[(size_t)(r + k) * cols] = v[k];
  }
  for (; r < rows; ++r) { carry = p[(size_t)r * cols] + carry; p[(size_t)r * cols] = carry; }
}

// ------------------------------------------------------------------------------------------------------------------
// StatsCalculatorIntegralImage::compute per-pixel loop (pwn_core/statscalculatorintegralimage.cpp:33-80) fused with
// PointInformationMatrixCalculator / NormalInformationMatrixCalculator::compute (informationmatrixcalculator.cpp:9-58)
// and Cloud::transformInPlace (cloud.cpp:173-186).  One thread per pixel, lanes along x.
// XCD-aware 1-D grid: workgroups are dealt round-robin over the 8 XCDs (block b and b+8 share an XCD and its 4 MiB L2), so
// block b works on frame 8*(k / blocksPerFrame) + (b % 8), k = b / 8, walking that frame's rows in order: every frame lives on
// one XCD and the four corner reads of neighbouring rows (2*radius rows apart, ~1.6 MB of planes) are L2 hits instead of one
// L2 fill per XCD.  Placement only affects speed.  grid = 8 * ceil(frames/8) * rows * ceil(cols/256) (frames >= 8; else frames * rows *
// ceil(cols/256), frame-major), block = 256.
__device__ __forceinline__ int clampi(int v, int lo, int hi) { v = (v < lo) ? lo : v; v = (v > hi) ? hi : v; return v; }
// streamed-once data of k_stats (index / interval / point in, cloud out) uses non-temporal accesses so that it does not evict the
// integral-image lines, which are the only data with reuse (each value is read by ~4 pixels)
template <typename PTR> __device__ __forceinline__ auto stream_load(PTR p) -> std::remove_cv_t<std::remove_reference_t<decltype(*p)>> {
  return __builtin_nontemporal_load(p);
}
template <typename PTR, typename T> __device__ __forceinline__ void stream_store(PTR p, T v) { __builtin_nontemporal_store(v, p); }
// the 12-byte records: not through the templates above (template argument deduction drops the typedef's alignment attribute and the access
// would be emitted with the vector type's natural `align 16`)
__device__ __forceinline__ v3f_raw stream_load3(gptr<const float> p) { return __builtin_nontemporal_load((gptr<const v3f>)p); }
__device__ __forceinline__ void stream_store3(gptr<float> p, v3f_raw v) { __builtin_nontemporal_store(v, (gptr<v3f>)p); }

// One pixel of the stats pass (everything after the thread has found its pixel): planes [10][rows][cols], point index from the index image.
__device__ __forceinline__ void stats_pixel(const FrameDesc& f, const ConvertParams& cp, const int r, const int c) {
  const int rows = cp.rows, cols = cp.cols;
  const size_t N = (size_t)rows * cols;
  // descriptor pointers are generic to the compiler; they all point to hipMalloc'ed memory: global_* instructions with a scalar
  // base and a 32-bit lane offset instead of flat_* with 64-bit VALU address arithmetic (the kernel is VALU-bound)
  const gptr<const int> gindex = as_global((const int*)f.index), ginterval = as_global((const int*)f.interval);
  const gptr<const float> gintegral = as_global((const float*)f.integral);
  const gptr<float> gP = as_global(f.cloud.P3), gN = as_global((float*)f.cloud.Nc), gOm = as_global(f.cloud.Om);
  const int cap = f.cloud.capacity;
  const unsigned upix = (unsigned)(r * cols + c);
  const int idx = stream_load(gindex + upix);
  if (idx < 0 || idx >= cap) return;
  int itv;
  float4 P;
  if (cp.lean) {
    // the front end kept the point and the interval to itself: the same expressions on the same depth (pinholepointprojector.h:246-251,264-274)
    float d;
    if (f.raw) { const unsigned sv = stream_load(as_global(f.raw) + upix); d = sv ? f.raw_scale * (float)sv : 0.0f; }
    else d = stream_load(as_global(f.depth) + upix);
    const float a = (float)c * d, b = (float)r * d;
    P.x = dot4seq(cp.iKRt(0,0), a, cp.iKRt(0,1), b, cp.iKRt(0,2), d, cp.iKRt(0,3), 1.0f);
    P.y = dot4seq(cp.iKRt(1,0), a, cp.iKRt(1,1), b, cp.iKRt(1,2), d, cp.iKRt(1,3), 1.0f);
    P.z = dot4seq(cp.iKRt(2,0), a, cp.iKRt(2,1), b, cp.iKRt(2,2), d, cp.iKRt(2,3), 1.0f);
    P.w = 0.f;
    const float inv = 1.0f / d;
    const float px = cp.ivx * inv, py = cp.ivy * inv;
    itv = (px > py) ? (int)px : (int)py;
  } else {
    itv = stream_load(ginterval + upix);
    const v3f_raw pv = stream_load3((gptr<const float>)gP + 3u * (unsigned)idx); P.x = pv.x; P.y = pv.y; P.z = pv.z; P.w = 0.f;      // one 12-byte load
  }
  float nx = 0.f, ny = 0.f, nz = 0.f;
  float curvature = 0.f;          // Stats() default: eigenvalues 0 -> curvature() = 0/(0+1e-9) = 0  (stats.h:21-27,98-103)
  int cls = 0;
  float om[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) om[k] = 0.f;
  float U[9] = { 1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f };   // column-major
  float ev[3] = { 0.f, 0.f, 0.f };
  float mean[3] = { 0.f, 0.f, 0.f };
  int npts = 0;
  if (itv >= 0) {
    int rad = itv;
    if (rad < cp.minRadius) rad = cp.minRadius;
    if (rad > cp.maxRadius) rad = cp.maxRadius;
    // PointIntegralImage::getRegion (pointintegralimage.cpp:53-66)
    const int xmin = clampi(c - rad - 1, 0, cols - 1), xmax = clampi(c + rad - 1, 0, cols - 1);
    const int ymin = clampi(r - rad - 1, 0, rows - 1), ymax = clampi(r + rad - 1, 0, rows - 1);
    const unsigned oA = (unsigned)(ymax * cols + xmax), oB = (unsigned)(ymin * cols + xmin);
    const unsigned oC = (unsigned)(ymax * cols + xmin), oD = (unsigned)(ymin * cols + xmax);
    float a[kIntegralChannels];
#pragma unroll
    for (int k = 0; k < kIntegralChannels; ++k) {
      const gptr<const char> pl = (gptr<const char>)(gintegral + (size_t)k * N);      // plane base: scalar; lane offsets: 32-bit bytes
      float v;
      v = *(gptr<const float>)(pl + 4u * oA);
      v = v + *(gptr<const float>)(pl + 4u * oB);
      v = v - *(gptr<const float>)(pl + 4u * oC);
      v = v - *(gptr<const float>)(pl + 4u * oD);
      a[k] = v;
    }
    const int n = (int)a[3];
    if (n >= cp.minPoints) {
      npts = n;
      // PointAccumulator::mean / covariance (pointaccumulator.h:66-86)
      float d = a[3];
      float c00 = 0, c10 = 0, c20 = 0, c11 = 0, c21 = 0, c22 = 0;
      if (d != 0.f) {
        d = 1.0f / d;
        mean[0] = a[0] * d; mean[1] = a[1] * d; mean[2] = a[2] * d;
        c00 = a[4] * d - mean[0] * mean[0];
        c10 = a[5] * d - mean[1] * mean[0];
        c20 = a[6] * d - mean[2] * mean[0];
        c11 = a[7] * d - mean[1] * mean[1];
        c21 = a[8] * d - mean[2] * mean[1];
        c22 = a[9] * d - mean[2] * mean[2];
      }
      Vec3 v0, v1, v2;
      eig3_direct(c00, c10, c20, c11, c21, c22, ev, v0, v1, v2);
      if (ev[0] < 0.0f) ev[0] = 0.0f;
      U[0] = v0.x; U[1] = v0.y; U[2] = v0.z; U[3] = v1.x; U[4] = v1.y; U[5] = v1.z; U[6] = v2.x; U[7] = v2.y; U[8] = v2.z;
      // Stats::curvature (stats.h:98-103): fp32 sum, double +1e-9 and divide
      curvature = (float)((double)ev[0] / ((double)(ev[0] + ev[1] + ev[2]) + 1e-9));
      if (curvature < cp.statsCurvThr) {
        nx = v0.x; ny = v0.y; nz = v0.z;
        // normal.dot(point) > 0 -> flip (4-vector dot, w term 0*1)
        const float dp = dot4seq(nx, P.x, ny, P.y, nz, P.z, 0.f, 1.f);
        if (dp > 0) { nx = -nx; ny = -ny; nz = -nz; }
      }
    }
  }
  // information matrices
  const float sq = dot4seq(nx, nx, ny, ny, nz, nz, 0.f, 0.f);
  if (sq > 0) {
    float dg[3];
    if (curvature < cp.pointInfoCurvThr) { dg[0] = cp.pFlat[0]; dg[1] = cp.pFlat[1]; dg[2] = cp.pFlat[2]; }
    else { dg[0] = 1.0f / ev[0]; dg[1] = 1.0f / ev[1]; dg[2] = 1.0f / ev[2]; }
    // (U * D) * U^T, U(i,k) = U[i + 3k]
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        om[3 * i + j] = dot3seq(U[i] * dg[0], U[j], U[i + 3] * dg[1], U[j + 3], U[i + 6] * dg[2], U[j + 6]);
    cls = (curvature < cp.normalInfoCurvThr) ? 1 : 2;
  }
  if (cp.keepStats && f.cloud.St) {
    float* st = f.cloud.St + (size_t)idx * 16;
    // Stats 4x4 after the optional sensor-offset left-multiplication (stats.h:125-131)
    if (cp.hasOffset) {
      Mat4 S = mat4_identity();
      if (npts > 0) {
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) S(i,j) = U[i + 3 * j];
        S(0,3) = mean[0]; S(1,3) = mean[1]; S(2,3) = mean[2]; S(3,3) = 1.f;
      }
      const Mat4 R = mat4_mul(cp.offset, S);
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) st[i + 3 * j] = R(i,j);
      st[12] = R(0,3); st[13] = R(1,3); st[14] = R(2,3);
    } else {
      for (int k = 0; k < 9; ++k) st[k] = U[k];
      st[12] = mean[0]; st[13] = mean[1]; st[14] = mean[2];
    }
    st[9] = ev[0]; st[10] = ev[1]; st[11] = ev[2];
    st[15] = (float)npts;
  }
  if (cp.hasOffset) {
    // Cloud::transformInPlace: p = m*p, n = m*n, Omega = T*Omega*T^t with T = m without last row/col
    const Mat4& m = cp.offset;
    const float px = dot4seq(m(0,0), P.x, m(0,1), P.y, m(0,2), P.z, m(0,3), 1.0f);
    const float py = dot4seq(m(1,0), P.x, m(1,1), P.y, m(1,2), P.z, m(1,3), 1.0f);
    const float pz = dot4seq(m(2,0), P.x, m(2,1), P.y, m(2,2), P.z, m(2,3), 1.0f);
    P.x = px; P.y = py; P.z = pz;
    const float tx = dot4seq(m(0,0), nx, m(0,1), ny, m(0,2), nz, m(0,3), 0.0f);
    const float ty = dot4seq(m(1,0), nx, m(1,1), ny, m(1,2), nz, m(1,3), 0.0f);
    const float tz = dot4seq(m(2,0), nx, m(2,1), ny, m(2,2), nz, m(2,3), 0.0f);
    nx = tx; ny = ty; nz = tz;
    float t1[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) t1[3 * i + j] = dot3seq(m(i,0), om[0 + j], m(i,1), om[3 + j], m(i,2), om[6 + j]);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) om[3 * i + j] = dot3seq(t1[3 * i], m(j,0), t1[3 * i + 1], m(j,1), t1[3 * i + 2], m(j,2));
  }
  (void)cls;      // not stored: normal_class(normal, curvature, normalInfoCurvThr) gives it back
  {
    // one 12-byte and one 16-byte store (dword stores per lane would write every 1 KiB segment of the wave several times at a fraction of the density)
    v3f_raw pv; pv.x = P.x; pv.y = P.y; pv.z = P.z;
    v4f nv; nv.x = nx; nv.y = ny; nv.z = nz; nv.w = curvature;
    stream_store3(gP + 3u * (unsigned)idx, pv);
    stream_store((gptr<v4f>)(gN + 4u * (unsigned)idx), nv);
  }
  if (cp.omSym) {                        // sym6: (xx xy xz) (yy yz zz), the upper triangle as computed -- 24 bytes per point instead of 36
    v3f_raw r0, r1; r0.x = om[0]; r0.y = om[1]; r0.z = om[2]; r1.x = om[4]; r1.y = om[5]; r1.z = om[8];
    stream_store3(gOm + 3u * (unsigned)idx, r0);
    stream_store3(gOm + 3u * (size_t)cap + 3u * (unsigned)idx, r1);
    return;
  }
#pragma unroll
  for (int r3 = 0; r3 < 3; ++r3) {      // three 12-byte rows (nine dword stores per point cost k_stats 9 % more time)
    v3f_raw rw; rw.x = om[3 * r3]; rw.y = om[3 * r3 + 1]; rw.z = om[3 * r3 + 2];
    stream_store3(gOm + (size_t)r3 * 3u * (size_t)cap + 3u * (unsigned)idx, rw);
  }
}

__global__ void __launch_bounds__(256) k_stats(const FrameDesc* __restrict__ frames, ConvertParams cp, int nframes) {
  const int nxb = (cp.cols + 255) / 256;
  const int perFrame = nxb * cp.rows;
  int frame, rem;
  if (nframes >= 8) {
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    frame = (k / perFrame) * 8 + xcd;
    rem = k % perFrame;
  } else {                                   // fewer frames than XCDs (tracker, makeCloud): a frame per XCD would idle the others; grid = nframes * perFrame
    frame = blockIdx.x / perFrame;
    rem = blockIdx.x % perFrame;
  }
  if (frame >= nframes) return;
  const FrameDesc& f = frames[frame];
  const int r = rem / nxb;
  const int c = (rem % nxb) * 256 + threadIdx.x;
  if (c >= cp.cols) return;
  stats_pixel(f, cp, r, c);
}

// Cloud::transformInPlace on an existing device cloud (cloud.cpp:173-186); grid = ceil(cap/256)
__global__ void __launch_bounds__(256) k_cloud_transform(CloudDev cl, Mat4 m) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= *cl.count || i >= cl.capacity) return;
  float4 P, Nm;
  cloud_get(cl, i, P, Nm);
  const float px = dot4seq(m(0,0), P.x, m(0,1), P.y, m(0,2), P.z, m(0,3), 1.0f);
  const float py = dot4seq(m(1,0), P.x, m(1,1), P.y, m(1,2), P.z, m(1,3), 1.0f);
  const float pz = dot4seq(m(2,0), P.x, m(2,1), P.y, m(2,2), P.z, m(2,3), 1.0f);
  P.x = px; P.y = py; P.z = pz;
  const float tx = dot4seq(m(0,0), Nm.x, m(0,1), Nm.y, m(0,2), Nm.z, m(0,3), 0.0f);
  const float ty = dot4seq(m(1,0), Nm.x, m(1,1), Nm.y, m(1,2), Nm.z, m(1,3), 0.0f);
  const float tz = dot4seq(m(2,0), Nm.x, m(2,1), Nm.y, m(2,2), Nm.z, m(2,3), 0.0f);
  Nm.x = tx; Nm.y = ty; Nm.z = tz;
  for (int pass = 0; pass < 2; ++pass) {
    float* base = pass == 0 ? cl.Om : cl.OmN;
    if (!base) continue;
    float om[9], t1[9];
    const int sym = pass == 0 ? cl.omSym : 0;
    for (int k = 0; k < 9; ++k) om[k] = base[omp_at(cl.capacity, i, k, sym)];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) t1[3 * a + b] = dot3seq(m(a,0), om[0 + b], m(a,1), om[3 + b], m(a,2), om[6 + b]);
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) om[3 * a + b] = dot3seq(t1[3 * a], m(b,0), t1[3 * a + 1], m(b,1), t1[3 * a + 2], m(b,2));
    for (int k = 0; k < 9; ++k) if (!(sym && om_is_lower(k))) base[omp_at(cl.capacity, i, k, sym)] = om[k];
  }
  cloud_put(cl, i, P, Nm);
}

// ------------------------------------------------------------------------------------------------------------------
// PinholePointProjector::project (pwn_core/pinholepointprojector.cpp:33-66): z-buffer by 64-bit atomicMin on
// (depth bits, point index): nearest point wins, ties keep the lowest index (the reference's strict '>' in a
// sequential loop).  grid = (ceil(capacity/256), pairs), block = 256.  which: 0 = reference cloud, 1 = current.
__device__ __forceinline__ void project_point(const Mat4& KRt, float minD, float maxD, int rows, int cols,
                                              const float4 p, int i, unsigned long long* z, unsigned tag) {
  const float ix = dot4seq(KRt(0,0), p.x, KRt(0,1), p.y, KRt(0,2), p.z, KRt(0,3), 1.0f);
  const float iy = dot4seq(KRt(1,0), p.x, KRt(1,1), p.y, KRt(1,2), p.z, KRt(1,3), 1.0f);
  const float d  = dot4seq(KRt(2,0), p.x, KRt(2,1), p.y, KRt(2,2), p.z, KRt(2,3), 1.0f);
  if (d < minD || d > maxD) return;
  const float inv = 1.0f / d;
  const float fx = roundf(ix * inv), fy = roundf(iy * inv);
  // int conversion of out-of-range floats is undefined on the CPU; such points are rejected by the bounds test
  if (!(fx >= 0.f && fx < (float)cols && fy >= 0.f && fy < (float)rows)) return;
  const int x = (int)fx, y = (int)fy;
  atomicMin(&z[(size_t)y * cols + x], zkey(tag, d, i));
}
// depth of a point under a projector matrix: the third row of _project (pinholepointprojector.h:224-233), same expression as project_point
__device__ __forceinline__ float point_depth(const Mat4& KRt, const float4 p) {
  return dot4seq(KRt(2,0), p.x, KRt(2,1), p.y, KRt(2,2), p.z, KRt(2,3), 1.0f);
}
// 32-bit z-buffer insert (see kZ32Tag0) in two steps, so that a thread with several points has all its atomics in flight before it
// looks at the first returned word.  z32_insert: the projection and the atomicMin; w = nullptr when the point is rejected.
struct Z32Pending { unsigned* w; unsigned key, old; float d; };
__device__ __forceinline__ Z32Pending z32_insert(const Mat4& KRt, float minD, float maxD, int rows, int cols, const float4 p, int i, unsigned* z, unsigned tag) {
  Z32Pending r; r.w = nullptr; r.key = 0u; r.old = ~0u; r.d = 0.f;
  const float ix = dot4seq(KRt(0,0), p.x, KRt(0,1), p.y, KRt(0,2), p.z, KRt(0,3), 1.0f);
  const float iy = dot4seq(KRt(1,0), p.x, KRt(1,1), p.y, KRt(1,2), p.z, KRt(1,3), 1.0f);
  const float d  = dot4seq(KRt(2,0), p.x, KRt(2,1), p.y, KRt(2,2), p.z, KRt(2,3), 1.0f);
  if (d < minD || d > maxD) return r;
  const float inv = 1.0f / d;
  const float fx = roundf(ix * inv), fy = roundf(iy * inv);
  // int conversion of out-of-range floats is undefined on the CPU; such points are rejected by the bounds test
  if (!(fx >= 0.f && fx < (float)cols && fy >= 0.f && fy < (float)rows)) return r;
  r.w = &z[(size_t)(int)fy * cols + (int)fx];
  r.key = z32key(tag, i); r.d = d;
  r.old = atomicMin(r.w, r.key);
  return r;
}
// z32_settle: nothing to do for the first point of this projection in its pixel (older tags and the empty word are larger than any key of
// the current tag); otherwise make sure the word ends up with the nearest of every point this thread gets to see, ties to the lower index.
// P: the cloud's points (to evaluate the depth of a point met in the pixel).
// The loop is lock-free (a failed compare-and-swap means another thread's write went through, and the word only changes a bounded number of
// times: one insert per point, and every successful swap installs a strictly nearer point), but its length has no useful bound when tens of
// thousands of points of one projection share a pixel.  A thread that has not settled after guardLimit rounds gives up and raises the
// call's fault word (page-locked host memory, PairDesc::fault): the host then repeats the call with the two-pass projection below, so a
// pixel is never silently left with the wrong point (pinholepointprojector.cpp:54-63).
constexpr int kSettleGuard = 4096;
__device__ __forceinline__ bool z32_settle(const Z32Pending& q, const Mat4& KRt, const float* __restrict__ P3, unsigned tag, int guardLimit) {
  if (!q.w || (q.old >> kZIndexBits) != tag) return true;
  int best = (int)(q.key & kZ32IndexMask); float dbest = q.d;
  unsigned cur = q.old < q.key ? q.old : q.key;       // what the word holds after the atomicMin, unless someone changed it since
  unsigned seen = q.old;
  for (int guard = 0; guard < guardLimit; ++guard) {
    const int j = (int)(seen & kZ32IndexMask);
    if (j != best) {
      const float dj = point_depth(KRt, load_xyz(P3, j));
      if (dj < dbest || (dj == dbest && j < best)) { best = j; dbest = dj; }
    }
    const unsigned want = z32key(tag, best);
    if (cur == want) return true;
    const unsigned r = atomicCAS(q.w, cur, want);
    if (r == cur) return true;                        // installed
    seen = r; cur = r;                                // somebody else wrote (same tag): weigh its point too and try again
  }
  return false;
}
// PPT points per thread: all loads first, then all atomicMin's, then the (rare) collisions -- with one point per thread the wave waits
// for every returned word before it does anything else (measured per 64-pair launch: 134 us with 1 point per thread, 104 with 2, 101
// with 4; the 64-bit no-return atomicMin it replaces: 131).  Small launches (single alignments) keep 1: they need the workgroups.
// grid.x = ceil(capacity / (256 * PPT))
template <int PPT>
__global__ void __launch_bounds__(256) k_project(const PairDesc* __restrict__ pairs, AlignParams ap, int which, unsigned tag) {
  constexpr int kProjectPointsPerThread = PPT;
  const PairDesc& pd = pairs[blockIdx.y];
  const CloudDev& cl = which ? pd.cur : pd.ref;
  const int n = min(*cl.count, cl.capacity);
  const int i0 = blockIdx.x * 256 * kProjectPointsPerThread + threadIdx.x;
  if (i0 >= n) return;
  const Mat4 KRt = uniform_iso(which ? pd.state->KRtCur : pd.state->KRt);
  unsigned* z = which ? pd.zcur : pd.zref;
  float4 p[kProjectPointsPerThread];
#pragma unroll
  for (int j = 0; j < kProjectPointsPerThread; ++j) { const int i = i0 + 256 * j; if (i < n) p[j] = load_xyz(cl.P3, i); }      // loads first (12 bytes per point)
  Z32Pending q[kProjectPointsPerThread];
#pragma unroll
  for (int j = 0; j < kProjectPointsPerThread; ++j) {
    const int i = i0 + 256 * j;
    if (i < n) q[j] = z32_insert(KRt, ap.minD, ap.maxD, ap.rows, ap.cols, p[j], i, z, tag); else q[j].w = nullptr;
  }
  bool settled = true;
#pragma unroll
  for (int j = 0; j < kProjectPointsPerThread; ++j) settled = z32_settle(q[j], KRt, cl.P3, tag, ap.settleGuard) && settled;
  if (!settled) __hip_atomic_store(pd.fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// The same projection without a retry loop, for calls whose k_project gave up (and for nothing else: it reads the points twice and clears a
// depth image per projection).  Pass 1: the nearest depth per pixel by atomicMin on the depth's bit pattern (depths are >= 0: min_distance >= 0
// is checked by the host, so the patterns order like the values; -0 counts as 0) in the pair's own depth image PairDesc::zdepth, cleared to ~0
// by the host before the launch.  Pass 2: every point that has its pixel's nearest depth enters the 32-bit word by atomicMin on tag | index:
// the nearest point, ties to the lowest index -- the reference's sequential strict '>' (pinholepointprojector.cpp:61), the word k_project leaves.
__device__ __forceinline__ bool project_pixel(const Mat4& KRt, const AlignParams& ap, const float4 p, int& pix, unsigned& dbits) {
  const float ix = dot4seq(KRt(0,0), p.x, KRt(0,1), p.y, KRt(0,2), p.z, KRt(0,3), 1.0f);
  const float iy = dot4seq(KRt(1,0), p.x, KRt(1,1), p.y, KRt(1,2), p.z, KRt(1,3), 1.0f);
  const float d  = dot4seq(KRt(2,0), p.x, KRt(2,1), p.y, KRt(2,2), p.z, KRt(2,3), 1.0f);
  if (d < ap.minD || d > ap.maxD) return false;
  const float inv = 1.0f / d;
  const float fx = roundf(ix * inv), fy = roundf(iy * inv);
  if (!(fx >= 0.f && fx < (float)ap.cols && fy >= 0.f && fy < (float)ap.rows)) return false;
  pix = (int)fy * ap.cols + (int)fx;
  dbits = __float_as_uint(d) & 0x7fffffffu;
  return true;
}
__global__ void __launch_bounds__(256) k_project_robust(const PairDesc* __restrict__ pairs, AlignParams ap, int which, unsigned tag, int pass) {
  const PairDesc& pd = pairs[blockIdx.y];
  const CloudDev& cl = which ? pd.cur : pd.ref;
  const int n = min(*cl.count, cl.capacity);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const Mat4 KRt = uniform_iso(which ? pd.state->KRtCur : pd.state->KRt);
  int pix; unsigned dbits;
  if (!project_pixel(KRt, ap, load_xyz(cl.P3, i), pix, dbits)) return;
  if (pass == 0) atomicMin(&pd.zdepth[pix], dbits);
  else if (pd.zdepth[pix] == dbits) atomicMin(&(which ? pd.zcur : pd.zref)[pix], z32key(tag, i));
}
// current-cloud z-buffer -> int index image, once per alignment.  grid = (blocks, pairs)
__global__ void k_resolve_cur(const PairDesc* __restrict__ pairs, int n, unsigned tag) {
  const PairDesc& pd = pairs[blockIdx.y];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    pd.curidx[i] = z32_index(pd.zcur[i], tag);
}
// CorrespondenceFinder::{reference,current}{Index,Depth}Image() of a pair after its alignment: index from the 32-bit z-buffer, depth
// recomputed from the winning point with the matrix its projection used; empty pixels -1 / FLT_MAX (pinholepointprojector.cpp:41-42)
__global__ void k_pair_images(const PairDesc* __restrict__ pairs, int which, unsigned tag, int n, int* __restrict__ index, float* __restrict__ depth) {
  const PairDesc& pd = pairs[0];
  const CloudDev& cl = which ? pd.cur : pd.ref;
  const unsigned* z = which ? pd.zcur : pd.zref;
  const Mat4 KRt = which ? pd.state->KRtCur : pd.state->KRtLast;
  const int np = min(*cl.count, cl.capacity);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int k = z32_index(z[i], tag);
    if (index) index[i] = k;
    if (depth) depth[i] = (k >= 0 && k < np) ? point_depth(KRt, load_xyz(cl.P3, k)) : FLT_MAX;
  }
}
// stand-alone projection with an explicit matrix (pwn_hip_project)
__global__ void __launch_bounds__(256) k_project_single(CloudDev cl, Mat4 KRt, float minD, float maxD, int rows, int cols,
                                                        unsigned long long* z, unsigned tag) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int n = min(*cl.count, cl.capacity);
  if (i >= n) return;
  project_point(KRt, minD, maxD, rows, cols, load_xyz(cl.P3, i), i, z, tag);
}
// z-buffer -> (index image, depth image): empty pixels -1 / FLT_MAX (pinholepointprojector.cpp:41-42)
__global__ void k_zbuf_resolve(const unsigned long long* __restrict__ z, int n, int* __restrict__ index, float* __restrict__ depth, unsigned tag) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const unsigned long long k = z[i];
    if (index) index[i] = zkey_index(k, tag);
    if (depth) depth[i] = zkey_depth(k, tag);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// One correspondence's contribution to the normal equations: Linearizer::update loop body
// (pwn_core/linearizer.cpp:57-88).  oP / oN: row-major 3x3 information matrices of the CURRENT point.
// The 4x4 products of the reference are written out for the non-zero 3x3 part; dropped terms are exact zeros.
__device__ __forceinline__ void skewT_mul(float tx, float ty, float tz, const float* m /*row-major 3x3*/, float* out) {
  // out = S^T * m with S = skew(v) = -2[v]x, (tx,ty,tz) = 2v   (bm_se3.h:54-66)
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    out[0 + j] = (-tz) * m[3 + j] + ty * m[6 + j];
    out[3 + j] = tz * m[0 + j] + (-tx) * m[6 + j];
    out[6 + j] = (-ty) * m[0 + j] + tx * m[3 + j];
  }
}
__device__ __forceinline__ void mul_skew(const float* a /*row-major 3x3*/, float tx, float ty, float tz, float* out) {
  // out = a * S
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    out[3 * i + 0] = a[3 * i + 1] * (-tz) + a[3 * i + 2] * ty;
    out[3 * i + 1] = a[3 * i + 0] * tz + a[3 * i + 2] * (-tx);
    out[3 * i + 2] = a[3 * i + 0] * (-ty) + a[3 * i + 1] * tx;
  }
}
// Accumulator policies of linearize_term: registers (RegAcc) or one private LDS column per thread (LdsAcc; frees ~34 VGPRs,
// which is what limits the occupancy of the fused kernel).  LdsAcc::add is a plain read-add-write of the thread's own slot.
struct RegAcc {
  float* a;
  __device__ __forceinline__ void add(int k, float v) const { a[k] += v; }
};
// The Gauss-Newton step solves with Matrix6f::ldlt(), which reads the LOWER triangle of H only (aligner.cpp:112; Eigen's LDLT
// default): the strictly upper entries of the Htt and Hrr blocks (sums 3, 6, 7 and 21, 24, 25) are never looked at, so the
// iteration kernels do not accumulate them (FULL = false: 28 sums).  Aligner::_computeStatistics inverts the full H, the pass
// that feeds it runs with FULL = true (34 sums).
__host__ __device__ constexpr bool acc_is_upper(int k) { return k == 3 || k == 6 || k == 7 || k == 21 || k == 24 || k == 25; }
__host__ __device__ constexpr int acc_slot_lower(int k) { return k - (k > 3) - (k > 6) - (k > 7) - (k > 21) - (k > 24) - (k > 25); }
template <bool FULL> struct LdsAcc {
  static constexpr int kSlots = FULL ? 34 : 28;
  float* base;      // &lds[threadIdx.x], stride kAlignBlock
  __device__ __forceinline__ void add(int k, float v) const {
    if (!FULL && acc_is_upper(k)) return;
    base[(FULL ? k : acc_slot_lower(k)) * kAlignBlock] += v;
  }
};
// a*b + c*d of the H / b products: two roundings of the products and one of the sum, as the CPU path evaluates them (no FMA)
#define MAD2(a, b, c, d) ((a) * (b) + (c) * (d))
// returns false if the term is rejected (non-robust kernel and chi2 above threshold)
template <typename ACC>
__device__ __forceinline__ bool linearize_term(const float3 rp, const float3 rn, const float3 cp, const float3 cn,
                                               const float* oP, const float* oN, float maxChi2, int robust, const ACC acc) {
  const float pe0 = rp.x - cp.x, pe1 = rp.y - cp.y, pe2 = rp.z - cp.z;
  const float ne0 = rn.x - cn.x, ne1 = rn.y - cn.y, ne2 = rn.z - cn.z;
  float ep[3], en[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    ep[i] = dot3seq(oP[3 * i], pe0, oP[3 * i + 1], pe1, oP[3 * i + 2], pe2);
    en[i] = dot3seq(oN[3 * i], ne0, oN[3 * i + 1], ne1, oN[3 * i + 2], ne2);
  }
  const float localError = dot3seq(pe0, ep[0], pe1, ep[1], pe2, ep[2]) + dot3seq(ne0, en[0], ne1, en[1], ne2, en[2]);
  float kscale = 1.f;
  if (localError > maxChi2) {
    if (robust) kscale = sqrtf(maxChi2 / localError);
    else return false;
  }
  const float ptx = 2 * rp.x, pty = 2 * rp.y, ptz = 2 * rp.z;
  const float ntx = 2 * rn.x, nty = 2 * rn.y, ntz = 2 * rn.z;
  // Sp^T ep + Sn^T en
  const float s0 = (MAD2((-ptz), ep[1], pty, ep[2])) + (MAD2((-ntz), en[1], nty, en[2]));
  const float s1 = (MAD2(ptz, ep[0], (-ptx), ep[2])) + (MAD2(ntz, en[0], (-ntx), en[2]));
  const float s2 = (MAD2((-pty), ep[0], ptx, ep[1])) + (MAD2((-nty), en[0], ntx, en[1]));
  // Row by row (keeps few values live): Htt += omegaP ; Htr += omegaP*Sp ; Hrr += (Sp^T omegaP Sp + Sn^T omegaN Sn).
  // accumulators are column-major 3x3 blocks: acc[base + i + 3*j]
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float a0 = oP[3 * i], a1 = oP[3 * i + 1], a2 = oP[3 * i + 2];
    acc.add(0 + i, a0); acc.add(0 + i + 3, a1); acc.add(0 + i + 6, a2);
    acc.add(9 + i, MAD2(a1, (-ptz), a2, pty));                    // (omegaP * Sp)(i, 0..2)
    acc.add(9 + i + 3, MAD2(a0, ptz, a2, (-ptx)));
    acc.add(9 + i + 6, MAD2(a0, (-pty), a1, ptx));
    // row i of Sp^T*omegaP and of Sn^T*omegaN
    float p0, p1, p2, q0, q1, q2;
    if (i == 0) {
      p0 = MAD2((-ptz), oP[3], pty, oP[6]); p1 = MAD2((-ptz), oP[4], pty, oP[7]); p2 = MAD2((-ptz), oP[5], pty, oP[8]);
      q0 = MAD2((-ntz), oN[3], nty, oN[6]); q1 = MAD2((-ntz), oN[4], nty, oN[7]); q2 = MAD2((-ntz), oN[5], nty, oN[8]);
    } else if (i == 1) {
      p0 = MAD2(ptz, oP[0], (-ptx), oP[6]); p1 = MAD2(ptz, oP[1], (-ptx), oP[7]); p2 = MAD2(ptz, oP[2], (-ptx), oP[8]);
      q0 = MAD2(ntz, oN[0], (-ntx), oN[6]); q1 = MAD2(ntz, oN[1], (-ntx), oN[7]); q2 = MAD2(ntz, oN[2], (-ntx), oN[8]);
    } else {
      p0 = MAD2((-pty), oP[0], ptx, oP[3]); p1 = MAD2((-pty), oP[1], ptx, oP[4]); p2 = MAD2((-pty), oP[2], ptx, oP[5]);
      q0 = MAD2((-nty), oN[0], ntx, oN[3]); q1 = MAD2((-nty), oN[1], ntx, oN[4]); q2 = MAD2((-nty), oN[2], ntx, oN[5]);
    }
    acc.add(18 + i, (MAD2(p1, (-ptz), p2, pty)) + (MAD2(q1, (-ntz), q2, nty)));
    acc.add(18 + i + 3, (MAD2(p0, ptz, p2, (-ptx))) + (MAD2(q0, ntz, q2, (-ntx))));
    acc.add(18 + i + 6, (MAD2(p0, (-pty), p1, ptx)) + (MAD2(q0, (-nty), q1, ntx)));
  }
  acc.add(27, kscale * ep[0]); acc.add(28, kscale * ep[1]); acc.add(29, kscale * ep[2]);
  acc.add(30, kscale * s0); acc.add(31, kscale * s1); acc.add(32, kscale * s2);
  acc.add(33, kscale * localError);
  return true;
}
__device__ __forceinline__ void load_omegas(const CloudDev& cur, int ci, int cls, float* oP, float* oN) {
  const size_t cap = (size_t)cur.capacity;
#pragma unroll
  for (int k = 0; k < 9; ++k) oP[k] = cur.Om[omp_at(cap, ci, k, cur.omSym)];
  if (cur.OmN) {
#pragma unroll
    for (int k = 0; k < 9; ++k) oN[k] = cur.OmN[om_at(cap, ci, k)];
  } else {
#pragma unroll
    for (int k = 0; k < 9; ++k) oN[k] = (cls == 1) ? cur.omN[0][k] : ((cls == 2) ? cur.omN[1][k] : 0.f);
  }
}
__device__ __forceinline__ float3 iso_point(const Mat4& T, const float4 p) {
  float3 r;
  r.x = dot4seq(T(0,0), p.x, T(0,1), p.y, T(0,2), p.z, T(0,3), 1.0f);
  r.y = dot4seq(T(1,0), p.x, T(1,1), p.y, T(1,2), p.z, T(1,3), 1.0f);
  r.z = dot4seq(T(2,0), p.x, T(2,1), p.y, T(2,2), p.z, T(2,3), 1.0f);
  return r;
}
__device__ __forceinline__ float3 iso_normal(const Mat4& T, const float4 n) {
  float3 r;
  r.x = dot4seq(T(0,0), n.x, T(0,1), n.y, T(0,2), n.z, T(0,3), 0.0f);
  r.y = dot4seq(T(1,0), n.x, T(1,1), n.y, T(1,2), n.z, T(1,3), 0.0f);
  r.z = dot4seq(T(2,0), n.x, T(2,1), n.y, T(2,2), n.z, T(2,3), 0.0f);
  return r;
}
// CorrespondenceFinder::compute acceptance tests (pwn_core/correspondencefinder.cpp:60-99) for one pixel.
// rp / rn return the reference point / normal remapped by Tc (reused by the linearizer when its transform is the same).
__device__ __forceinline__ bool correspondence_test(const AlignParams& ap, const Mat4& Tc, const float4 rP, const float4 rN,
                                                    const float4 cP, const float4 cN, float3& rp, float3& rn) {
  if (dot4seq(cN.x, cN.x, cN.y, cN.y, cN.z, cN.z, 0.f, 0.f) == 0.0f || dot4seq(rN.x, rN.x, rN.y, rN.y, rN.z, rN.z, 0.f, 0.f) == 0.0f)
    return false;
  rp = iso_point(Tc, rP);
  rn = iso_normal(Tc, rN);
  if (dot4seq(cN.x, rn.x, cN.y, rn.y, cN.z, rn.z, 0.f, 0.f) < ap.normalThr) return false;
  const float dx = cP.x - rp.x, dy = cP.y - rp.y, dz = cP.z - rp.z;
  if (dot4seq(dx, dx, dy, dy, dz, dz, 0.f, 0.f) > ap.sqDist) return false;
  float rc = rP.w, cc = cP.w;
  if (rc < ap.flatThr) rc = ap.flatThr;
  if (cc < ap.flatThr) cc = ap.flatThr;
  // The reference evaluates (rc + 1e-5) / (cc + 1e-5) in double and compares the float-rounded ratio with the bounds
  // (correspondencefinder.cpp:96-99).  An fp32 estimate (relative error < 1e-6) decides every case that is not within
  // 1e-5 of a bound; only those run the exact double computation, so the decision is always the reference's.
  const float est = (rc + 1e-5f) / (cc + 1e-5f);
  if (est < ap.minRatio * (1.0f - 1e-5f) || est > ap.maxRatio * (1.0f + 1e-5f)) return false;
  if (est > ap.minRatio * (1.0f + 1e-5f) && est < ap.maxRatio * (1.0f - 1e-5f)) return true;
  const float ratio = (float)(((double)rc + 1e-5) / ((double)cc + 1e-5));
  if (ratio < ap.minRatio || ratio > ap.maxRatio) return false;
  return true;
}

// block-level reduction of kAccN fp32 accumulators -> fp64 partial record of the block
template <bool FULL = true>
__device__ __forceinline__ void block_reduce_store(float* acc, double* out_generic) {
  const gptr<double> out = as_global(out_generic);
  __shared__ float red[kAlignBlock / 64][kAccN];
#pragma unroll
  for (int k = 0; k < kAccN; ++k) {
    if (!FULL && k < 34 && acc_is_upper(k)) {            // sums the lower-triangle passes do not accumulate: exact zeros, no reduction
      if (lane_id() == 63) red[threadIdx.x >> 6][k] = 0.f;
      continue;
    }
    const float v = wave_sum_dpp_lane63(acc[k]);
    if (lane_id() == 63) red[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < kAccN) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kAlignBlock / 64; ++w) s += (double)red[w][threadIdx.x];
    out[threadIdx.x] = s;
  }
}

// Fused CorrespondenceFinder::compute + Linearizer::update (correspondencefinder.cpp:45-106, linearizer.cpp:33-90):
// one pass over the pixels, no correspondence list is materialised; the z-buffer is epoch-tagged, nothing is written back.
// The kernel is latency-bound (index -> point/normal gathers -> information-matrix gathers are dependent loads), so it
// is software-pipelined by hand as a rolling three-stage loop: indices of pixel j+2, point/normal gathers of pixel j+1,
// tests + arithmetic of pixel j.  Measured on MI355X: prefetching the 9 information-matrix planes as well costs more
// in occupancy (174 VGPRs -> 2 waves/SIMD) than it hides (13.2 ms vs 9.8 ms per 128x10 pair-iterations; docs/experiments.md).
// grid = (ceil(N / (256*kPixPerThread)), pairs), block = 256.
struct Candidate {
  float4 rP, rN, cP, cN;
  int ci;
  bool valid;
};
// The descriptor's pointers are generic to the compiler: flat_* loads, which count on BOTH the vector-memory and the LDS counter, so
// every wait for an LDS accumulator update would also wait for the gathers of the next pixel that are meant to stay in flight.
// They all point to hipMalloc'ed memory: as_global() -> global_* loads (vmcnt only).
struct PairPtrs {
  gptr<const float> refP, curP;      // packed xyz
  gptr<const v4f> refN, curN;        // (normal, curvature)
  gptr<const float> curOm, curOmN;
  gptr<const unsigned> zref;
  gptr<const int> curidx, refidx0;
  unsigned cap;      // capacity of the current cloud (plane stride)
};
__device__ __forceinline__ PairPtrs pair_ptrs(const PairDesc& pd) {
  PairPtrs q;
  q.refP = as_global((const float*)pd.ref.P3); q.refN = as_global((const v4f*)pd.ref.Nc);
  q.curP = as_global((const float*)pd.cur.P3); q.curN = as_global((const v4f*)pd.cur.Nc);
  q.curOm = as_global((const float*)pd.cur.Om); q.curOmN = as_global((const float*)pd.cur.OmN);
  q.zref = as_global((const unsigned*)pd.zref); q.curidx = as_global((const int*)pd.curidx);
  q.refidx0 = as_global(pd.refidx0);
  q.cap = (unsigned)pd.cur.capacity;
  return q;
}
__device__ __forceinline__ void candidate_load(const PairPtrs& q, int ri, int ci, int nref, int ncur, Candidate& c) {
  c.valid = !(ri < 0 || ci < 0 || ri >= nref || ci >= ncur);
  c.ci = ci;
  if (c.valid) {
    // 12 + 16 bytes per side; the rest of the pass keeps the (point, curvature) / (normal, -) records it was written for
    c.rP = load_xyz(q.refP, (unsigned)ri); c.cP = load_xyz(q.curP, (unsigned)ci); c.cN = load4(q.curN + (unsigned)ci);
    c.rN = load4(q.refN + (unsigned)ri);
    c.rP.w = c.rN.w; c.cP.w = c.cN.w;
  }
}
template <bool SAME_T, bool SYM, typename ACC>
__device__ __forceinline__ void candidate_consume(const PairDesc& pd, const PairPtrs& q, const AlignParams& ap, const Mat4& Tc, const Mat4& Tl,
                                                  const Candidate& c, const ACC acc, float* cnt /* K, C, inliers */, const float* omNtab /* LDS [3][9] */) {
  if (!c.valid) return;
  cnt[0] += 1.f;
  float3 rp, rn;
  if (!correspondence_test(ap, Tc, c.rP, c.rN, c.cP, c.cN, rp, rn)) return;
  cnt[1] += 1.f;
  float oN[9];
  const int cls = (c.cP.w < pd.cur.clsThr) ? 1 : 2;      // normal_class(): the normal is not zero here (correspondence_test)
  if (pd.cur.OmN) {
#pragma unroll
    for (int k = 0; k < 9; ++k) oN[k] = q.curOmN[om_at(q.cap, (unsigned)c.ci, k)];
  } else {
    // class -> matrix through a 27-float LDS table (row 0 = zero matrix): one broadcast read per entry.  Selecting from the descriptor
    // compiled into a tree of exec-masked branches with a scalar load and a wait inside each, 18 of them per correspondence.
    const float* t = omNtab + 9 * (cls > 2 ? 0 : cls);
#pragma unroll
    for (int k = 0; k < 9; ++k) oN[k] = t[k];
  }
  if (!SAME_T) { rp = iso_point(Tl, c.rP); rn = iso_normal(Tl, c.rN); }   // inner iterations > 0: the linearizer's transform moved on
  float oP[9];
  {
    const size_t cap = (size_t)q.cap;
    const unsigned ci = (unsigned)c.ci;
    if (SYM) {                            // sym6 storage: two 12-byte loads, the lower triangle mirrored in registers (no extra VALU)
      const v3f_raw r0 = *(gptr<const v3f>)(q.curOm + 3u * ci);
      const v3f_raw r1 = *(gptr<const v3f>)(q.curOm + 3u * cap + 3u * ci);
      oP[0] = r0.x; oP[1] = r0.y; oP[2] = r0.z; oP[3] = r0.y; oP[4] = r1.x; oP[5] = r1.y; oP[6] = r0.z; oP[7] = r1.y; oP[8] = r1.z;
    } else {
#pragma unroll
      for (int r3 = 0; r3 < 3; ++r3) {      // three 12-byte loads
        const v3f_raw rw = *(gptr<const v3f>)(q.curOm + (size_t)r3 * 3u * cap + 3u * ci);
        oP[3 * r3] = rw.x; oP[3 * r3 + 1] = rw.y; oP[3 * r3 + 2] = rw.z;
      }
    }
  }
  if (linearize_term(rp, rn, make_float3(c.cP.x, c.cP.y, c.cP.z), make_float3(c.cN.x, c.cN.y, c.cN.z), oP, oN, ap.maxChi2, ap.robust, acc))
    cnt[2] += 1.f;
}
// usePrevTc: the acceptance tests run with the previous outer iteration's transform (Aligner::_computeStatistics re-linearizes
// the finder's existing correspondences at the final transform, aligner.cpp:165-170).
template <bool SAME_T, bool FULL_H, bool SYM = false, int PPT = kPixPerThread>
__global__ void __launch_bounds__(kAlignBlock, 1) k_corr_linearize(const PairDesc* __restrict__ pairs, AlignParams ap, unsigned tag, int usePrevTc, int ownRefIndex) {
  const PairDesc& pd = pairs[blockIdx.y];
  const int N = ap.rows * ap.cols;
  const PairState* stp = pd.state;
  const Mat4 Tc = uniform_iso_global(as_global((const float*)(usePrevTc ? stp->invTcorrPrev.m : stp->invTcorr.m)));
  const Mat4 Tl = uniform_iso_global(as_global((const float*)stp->invT.m));
  constexpr int kSlots = LdsAcc<FULL_H>::kSlots;
  __shared__ float lacc[kSlots * kAlignBlock];        // float sums of the thread in column threadIdx.x (bank-conflict free)
#pragma unroll
  for (int k = 0; k < kSlots; ++k) lacc[k * kAlignBlock + threadIdx.x] = 0.f;
  const LdsAcc<FULL_H> sums = { &lacc[threadIdx.x] };
  __shared__ float omNtab[27];
  if (threadIdx.x < 27) omNtab[threadIdx.x] = threadIdx.x < 9 ? 0.f : pd.cur.omN[(threadIdx.x - 9) / 9][(threadIdx.x - 9) % 9];
  __syncthreads();
  float cnt[3] = { 0.f, 0.f, 0.f };
  const int nref = min(*as_global((const int*)pd.ref.count), pd.ref.capacity), ncur = min(*as_global((const int*)pd.cur.count), pd.cur.capacity);
  const int pix0 = blockIdx.x * PPT * kAlignBlock + threadIdx.x;
  const PairPtrs q = pair_ptrs(pd);
  auto load_indices = [&](int j, int& ri, int& ci) {
    const int pix = pix0 + j * kAlignBlock;
    ri = -1; ci = -1;
    if (j < PPT && pix < N) {
      ri = ownRefIndex ? q.refidx0[(unsigned)pix] : z32_index(q.zref[(unsigned)pix], tag);      // wave-uniform choice
      ci = q.curidx[(unsigned)pix];
    }
  };
  int ri1, ci1, ri2, ci2;
  load_indices(0, ri1, ci1);
  load_indices(1, ri2, ci2);
  Candidate nxt;
  candidate_load(q, ri1, ci1, nref, ncur, nxt);
#pragma unroll 1
  for (int j = 0; j < PPT; ++j) {
    const Candidate cur = nxt;
    candidate_load(q, ri2, ci2, nref, ncur, nxt);        // gathers of pixel j+1: in flight during the arithmetic below
    load_indices(j + 2, ri2, ci2);                        // indices of pixel j+2
    candidate_consume<SAME_T, SYM>(pd, q, ap, Tc, Tl, cur, sums, cnt, omNtab);
  }
  float acc[kAccN];
#pragma unroll
  for (int k = 0; k < 34; ++k)
    acc[k] = (FULL_H || !acc_is_upper(k)) ? lacc[(FULL_H ? k : acc_slot_lower(k)) * kAlignBlock + threadIdx.x] : 0.f;
  acc[36] = cnt[0]; acc[35] = cnt[1]; acc[34] = cnt[2];
  block_reduce_store<FULL_H>(acc, pd.partials + (size_t)blockIdx.x * kAccN);
}

// Latency shape of the fused pass for small launches (sequential odometry: PwnTracker::processFrame aligns one pair at a time).  A pair is
// only 150 workgroups of k_corr_linearize, one wave per SIMD, so the launch lasts as long as one thread's chain of 8 dependent pixels.  Here
// a workgroup has 1024 threads for the same 2048-pixel tile: thread (q, col) = (tid / 256, tid % 256) takes pixels q and q + 4 of column col, so
// the gathers and the arithmetic of all eight pixels of a column run side by side.  The sums stay bit-identical to k_corr_linearize: the
// per-pixel terms are added to the column's LDS accumulator in the same order, pixel 0, 1, ... 7 -- eight ordered turns, a workgroup barrier
// between turns -- and the block reduction is the same tree on the same 256 columns.  (Putting the four threads of a column into one wave
// needs no barriers but makes every wave walk all eight turns and quarters the gathers' coalescing: measured 19 vs 16 us per launch.)
// grid = (ceil(N / 2048), pairs), block = 1024.
constexpr int kLatBlock = 4 * kAlignBlock;
template <bool SAME_T, bool FULL_H, bool SYM = false>
__global__ void __launch_bounds__(kLatBlock) k_corr_linearize_lat(const PairDesc* __restrict__ pairs, AlignParams ap, unsigned tag, int usePrevTc, int ownRefIndex) {
  static_assert(kPixPerThread == 8, "two rounds of four turns");
  const PairDesc& pd = pairs[blockIdx.y];
  const int N = ap.rows * ap.cols;
  const PairState* stp = pd.state;
  const Mat4 Tc = uniform_iso_global(as_global((const float*)(usePrevTc ? stp->invTcorrPrev.m : stp->invTcorr.m)));
  const Mat4 Tl = uniform_iso_global(as_global((const float*)stp->invT.m));
  constexpr int kSlots = LdsAcc<FULL_H>::kSlots;
  __shared__ float lacc[kSlots * kAlignBlock];
  __shared__ float lcnt[3 * kAlignBlock];
  __shared__ float omNtab[27];
  const int tid = threadIdx.x, col = tid & (kAlignBlock - 1), q = tid >> 8;
  for (int i = tid; i < kSlots * kAlignBlock; i += kLatBlock) lacc[i] = 0.f;
  if (tid < 3 * kAlignBlock) lcnt[tid] = 0.f;
  if (tid < 27) omNtab[tid] = tid < 9 ? 0.f : pd.cur.omN[(tid - 9) / 9][(tid - 9) % 9];
  const int nref = min(*as_global((const int*)pd.ref.count), pd.ref.capacity), ncur = min(*as_global((const int*)pd.cur.count), pd.cur.capacity);
  const int pix0 = blockIdx.x * kPixPerThread * kAlignBlock + col;
  const PairPtrs pp = pair_ptrs(pd);
  // both pixels' indices, then both pixels' gathers, then both pixels' terms: every load is in flight before the first ordered turn
  int ri[2], ci[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int pix = pix0 + (q + 4 * r) * kAlignBlock;
    ri[r] = -1; ci[r] = -1;
    if (pix < N) {
      ri[r] = ownRefIndex ? pp.refidx0[(unsigned)pix] : z32_index(pp.zref[(unsigned)pix], tag);
      ci[r] = pp.curidx[(unsigned)pix];
    }
  }
  Candidate cand[2];
  candidate_load(pp, ri[0], ci[0], nref, ncur, cand[0]);
  candidate_load(pp, ri[1], ci[1], nref, ncur, cand[1]);
  __syncthreads();                                   // accumulators zeroed, class table written
  float t[2][kAccN], c3[2][3];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
#pragma unroll
    for (int k = 0; k < kAccN; ++k) t[r][k] = 0.f;
    c3[r][0] = c3[r][1] = c3[r][2] = 0.f;
    candidate_consume<SAME_T, SYM>(pd, pp, ap, Tc, Tl, cand[r], RegAcc{ t[r] }, c3[r], omNtab);
  }
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const bool has = c3[r][2] > 0.f;              // linearize_term added this pixel's terms
#pragma unroll 1
    for (int turn = 0; turn < 4; ++turn) {
      if (q == turn) {
        if (has) {
#pragma unroll
          for (int k = 0; k < 34; ++k)
            if (FULL_H || !acc_is_upper(k)) lacc[(FULL_H ? k : acc_slot_lower(k)) * kAlignBlock + col] += t[r][k];
        }
        lcnt[col] += c3[r][0]; lcnt[kAlignBlock + col] += c3[r][1]; lcnt[2 * kAlignBlock + col] += c3[r][2];     // counts: integers, exact in any order
      }
      __syncthreads();
    }
  }
  // block reduction: the tree of block_reduce_store on the 256 columns (waves 0..3), the other waves only meet the barrier
  __shared__ float red[kAlignBlock / 64][kAccN];
  if (tid < kAlignBlock) {
    float acc[kAccN];
#pragma unroll
    for (int k = 0; k < 34; ++k)
      acc[k] = (FULL_H || !acc_is_upper(k)) ? lacc[(FULL_H ? k : acc_slot_lower(k)) * kAlignBlock + tid] : 0.f;
    acc[36] = lcnt[tid]; acc[35] = lcnt[kAlignBlock + tid]; acc[34] = lcnt[2 * kAlignBlock + tid];
#pragma unroll
    for (int k = 0; k < kAccN; ++k) {
      if (!FULL_H && k < 34 && acc_is_upper(k)) { if (lane_id() == 63) red[tid >> 6][k] = 0.f; continue; }
      const float v = wave_sum_dpp_lane63(acc[k]);
      if (lane_id() == 63) red[tid >> 6][k] = v;
    }
  }
  __syncthreads();
  if (tid < kAccN) {
    const gptr<double> out = as_global(pd.partials + (size_t)blockIdx.x * kAccN);
    double sum = 0.0;
#pragma unroll
    for (int w = 0; w < kAlignBlock / 64; ++w) sum += (double)red[w][tid];
    out[tid] = sum;
  }
}

// CorrespondenceFinder::compute as a per-pixel pair image (compacted on the host in row-major order).
__global__ void __launch_bounds__(256) k_correspondence_image(CloudDev ref, CloudDev cur, const int* __restrict__ refIndex,
                                                              const int* __restrict__ curIndex, AlignParams ap, Mat4 Tc,
                                                              int2* __restrict__ out, int* __restrict__ counters) {
  const int N = ap.rows * ap.cols;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= N) return;
  int2 res = make_int2(-1, -1);
  const int ri = refIndex[pix], ci = curIndex[pix];
  const int nref = min(*ref.count, ref.capacity), ncur = min(*cur.count, cur.capacity);
  if (ri >= 0 && ci >= 0 && ri < nref && ci < ncur) {
    atomicAdd(&counters[0], 1);
    float3 rp, rn;
    float4 rP, rN, cP, cN;
    cloud_get(ref, ri, rP, rN); cloud_get(cur, ci, cP, cN);
    if (correspondence_test(ap, Tc, rP, rN, cP, cN, rp, rn)) res = make_int2(ri, ci);
  }
  out[pix] = res;
}
// Linearizer::update on an explicit list.  grid = ceil(C/(256*4)), block 256.
__global__ void __launch_bounds__(kAlignBlock) k_linearize_list(CloudDev ref, CloudDev cur, const int2* __restrict__ corr, int C,
                                                                AlignParams ap, Mat4 Tl, double* __restrict__ partials) {
  float acc[kAccN];
#pragma unroll
  for (int k = 0; k < kAccN; ++k) acc[k] = 0.f;
  const int nref = min(*ref.count, ref.capacity), ncur = min(*cur.count, cur.capacity);
#pragma unroll
  for (int j = 0; j < kPixPerThread; ++j) {
    const int i = (blockIdx.x * kPixPerThread + j) * kAlignBlock + threadIdx.x;
    if (i >= C) continue;
    const int2 c = corr[i];
    if (c.x < 0 || c.y < 0 || c.x >= nref || c.y >= ncur) continue;
    acc[35] += 1.f;
    float4 rP, rN, cP, cN;
    cloud_get(ref, c.x, rP, rN); cloud_get(cur, c.y, cP, cN);
    float oP[9], oN[9];
    load_omegas(cur, c.y, __float_as_int(cN.w) & kClsMask, oP, oN);
    const float3 rp = iso_point(Tl, rP), rn = iso_normal(Tl, rN);
    if (linearize_term(rp, rn, make_float3(cP.x, cP.y, cP.z), make_float3(cN.x, cN.y, cN.z), oP, oN, ap.maxChi2, ap.robust, RegAcc{ acc })) acc[34] += 1.f;
  }
  block_reduce_store(acc, partials + (size_t)blockIdx.x * kAccN);
}

// ------------------------------------------------------------------------------------------------------------------
// PwnMatcherBase::matchClouds post-alignment score (pwn_tracker/pwn_matcher_base.cpp:153-182): both finder depth images
// -> uint16 millimetres (DepthImage_convert_32FC1_to_16UC1, FLT_MAX -> 0), mask = both > 0, diff = |cur - ref| as float
// BITWISE-ANDed with the float mask 255.0f (that is what cv::Mat `abs(a-b) & mask` does on CV_32F data), inliers =
// mask && diff < threshold, sum of diff.  Every surviving diff is a multiple of 1/64 (or < 2^-120), so the sum is kept
// exactly in 1/64 fixed point with integer atomics -> bitwise reproducible.  out[pair] = {nonZeros, inliers, sum64, tiny}.
struct MatchAcc { unsigned long long nonZeros, inliers; long long sum64; unsigned long long tiny; };
// sum of the masked differences / nonZeros (pwn_matcher_base.cpp:179; 0/0 = NaN like the reference): the sum is exact in 1/64 units; values
// below 2^-120 only matter when nothing else was added.  One expression for the host (pwn_hip_match_result) and the device (result records).
__host__ __device__ __forceinline__ float match_reprojection_distance(const MatchAcc& a) {
  const double sum = a.sum64 ? (double)a.sum64 / 64.0 : (double)a.tiny * 1e-37;
  return (float)sum / (float)(int)a.nonZeros;
}
__device__ __forceinline__ unsigned short depth_to_u16(float d, float scale) { return (d < FLT_MAX) ? (unsigned short)(scale * d) : (unsigned short)0; }
// curOwn: the current cloud was not projected (batch path, identity current offset): its depth image is read off the cloud through its
// own index image -- pixel -> point -> z.  With an identity offset the projector's depth is the point's z bit for bit (row 2 of K*R
// is (0, 0, 1), K*t is 0: d = ((0*x + 0*y) + 1*z) + 0*1), and every point lands on its own pixel (see pwn_hip_cloud::idximg).
__global__ void __launch_bounds__(256) k_match_score(const PairDesc* __restrict__ pairs, int n, unsigned refTag, unsigned curTag, float scale,
                                                     float threshold, MatchAcc* __restrict__ out, int curOwn) {
  const PairDesc& pd = pairs[blockIdx.y];
  int nz = 0, inl = 0; long long sum = 0; int tiny = 0;
  const int ncur = min(*pd.cur.count, pd.cur.capacity), nref = min(*pd.ref.count, pd.ref.capacity);
  const Mat4 KRtRef = uniform_iso(pd.state->KRtLast), KRtCur = uniform_iso(pd.state->KRtCur);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float dc = FLT_MAX, dr = FLT_MAX;
    if (curOwn) { const int ci = pd.curidx[i]; if (ci >= 0 && ci < ncur) dc = load_xyz(pd.cur.P3, ci).z; }
    else { const int ci = z32_index(pd.zcur[i], curTag); if (ci >= 0 && ci < ncur) dc = point_depth(KRtCur, load_xyz(pd.cur.P3, ci)); }
    { const int ri = z32_index(pd.zref[i], refTag); if (ri >= 0 && ri < nref) dr = point_depth(KRtRef, load_xyz(pd.ref.P3, ri)); }
    const unsigned short c = depth_to_u16(dc, scale);
    const unsigned short r = depth_to_u16(dr, scale);
    if (c > 0 && r > 0) {
      ++nz;
      const float ad = fabsf((float)c - (float)r);
      const unsigned bits = __float_as_uint(ad) & 0x437F0000u;            // & bits(255.0f)
      const float d = __uint_as_float(bits);
      if (d < threshold) ++inl;
      const unsigned e = bits >> 23;
      if (e >= 128u) sum += (long long)(128u + ((bits >> 16) & 0x7Fu)) << ((e - 128u));   // d * 64 = (128+m7) * 2^(e-128)
      else if (bits) ++tiny;                                               // d < 2^-120: cannot change a float sum >= 1/64
    }
  }
  // wave reduction of the integers, the four waves of the block combined through LDS, then ONE set of atomics per block: the record of a pair is
  // a single 32-byte target, and memory-side atomics on one address serialise (one set per wave from 256 blocks per pair cost a 64-pair launch
  // 0.68 ms -- 5 x what its reads take; integer sums: any order gives the same bits)
  for (int off = 32; off > 0; off >>= 1) {
    nz += __shfl_xor(nz, off, 64); inl += __shfl_xor(inl, off, 64); tiny += __shfl_xor(tiny, off, 64);
    sum += __shfl_xor(sum, off, 64);
  }
  __shared__ long long part[4][4];
  if (lane_id() == 0) { long long* q = part[threadIdx.x >> 6]; q[0] = nz; q[1] = inl; q[2] = sum; q[3] = tiny; }
  __syncthreads();
  if (threadIdx.x == 0) {
    long long t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = part[0][k] + part[1][k] + part[2][k] + part[3][k];
    MatchAcc* o = out + blockIdx.y;
    if (t[0]) atomicAdd(&o->nonZeros, (unsigned long long)t[0]);
    if (t[1]) atomicAdd(&o->inliers, (unsigned long long)t[1]);
    if (t[2]) atomicAdd((unsigned long long*)&o->sum64, (unsigned long long)t[2]);
    if (t[3]) atomicAdd(&o->tiny, (unsigned long long)t[3]);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Deterministic final reduction (fixed block order, fp64) + the Gauss-Newton step of Aligner::align
// (pwn_core/aligner.cpp:86-117): H = H_lin + I + 1000 I, dx = ldlt(H) \ (-b), invT = v2t(dx) * invT, and at the
// end of an outer iteration _T = v2t(t2v(invT^-1)) plus the projector matrix of the next reference projection.
// grid = pairs, block = 256.
struct SolveOut { float H[36]; float b[6]; float chi2; int inliers, ncorr, ncand; };
// fixed partition: lane c < kAccN of wave s sums blocks s, s+4, s+8, ... in order; the four wave sums are
// combined in a fixed order -> bitwise reproducible.  blockDim.x must be 256.
__device__ __forceinline__ void reduce_partials(const double* partials, int nblocks, double* sums /*shared[kAccN]*/) {
  __shared__ double part[4][kAccN];
  const int c = threadIdx.x & 63, s = threadIdx.x >> 6;
  if (c < kAccN) {
    double acc = 0.0;
    int b = s;
    // up to kRW independent loads in flight, then the adds in block order (one pair at VGA: 150 records, 38 per wave: ONE memory round trip;
    // with 32 + a remainder loop it was three).  Records past the end are neither loaded nor added: the sum is the same chain as before.
    constexpr int kRW = 40;
    for (; b < nblocks; b += 4 * kRW) {
      double v[kRW];
#pragma unroll
      for (int i = 0; i < kRW; ++i) { const int bb = b + 4 * i; v[i] = (bb < nblocks) ? partials[(size_t)bb * kAccN + c] : 0.0; }
#pragma unroll
      for (int i = 0; i < kRW; ++i) if (b + 4 * i < nblocks) acc += v[i];
    }
    part[s][c] = acc;
  }
  __syncthreads();
  if (threadIdx.x < kAccN) sums[threadIdx.x] = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
  __syncthreads();
}
__device__ __forceinline__ void assemble_Hb(const double* s, float* H, float* b) {
  // Linearizer.cpp:109-114
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      H[i + 6 * j] = (float)s[0 + i + 3 * j];
      H[i + 6 * (j + 3)] = (float)s[9 + i + 3 * j];
      H[(i + 3) + 6 * (j + 3)] = (float)s[18 + i + 3 * j];
    }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) H[(i + 3) + 6 * j] = H[j + 6 * (i + 3)];
#pragma unroll
  for (int i = 0; i < 3; ++i) { b[i] = (float)s[27 + i]; b[i + 3] = (float)s[30 + i]; }
}
// The step is written out in the kernel, on registers with constant indices only.  (Round 3 had it as a pure function returning the new state in a
// struct, for two experiments that evaluated it inside other kernels: the struct cost the kernel 29 VGPRs, 68 bytes of scratch and a 16 KB
// LDS-promoted array, and a 64-pair launch 18 us instead of 10.5 -- found in the v14 kernel trace and undone.)
// last: this is the alignment's final step -- with pd.state_out set (a few pairs: latency path), the pose goes to the page-locked host copy of
// the state as well (the traces go there step by step), and the host needs no copy back of the state.
__global__ void __launch_bounds__(256) k_solve_update(const PairDesc* __restrict__ pairs, AlignParams ap, int nblocks, int outerEnd, int last) {
  const PairDesc& pd = pairs[blockIdx.x];
  __shared__ double sums[kAccN];
  reduce_partials(pd.partials, nblocks, sums);
  if (threadIdx.x != 0) return;
  PairState& st = *pd.state;
  PairState* const so = pd.state_out;
  float H[36], b[6];                                                     // registers: every index below is a constant
  assemble_Hb(sums, H, b);
  const int it = st.it;
  if (it < kMaxIter) {
    st.chi2[it] = (float)sums[33];
    st.inliers[it] = (int)sums[34];
    st.ncorr[it] = (int)sums[35];
    st.ncand[it] = (int)sums[36];
    if (so) { so->chi2[it] = (float)sums[33]; so->inliers[it] = (int)sums[34]; so->ncorr[it] = (int)sums[35]; so->ncand[it] = (int)sums[36]; }
  }
  st.it = it + 1;
#pragma unroll
  for (int d = 0; d < 6; ++d) H[d + 6 * d] = H[d + 6 * d] + 1.0f;        // aligner.cpp:92
#pragma unroll
  for (int d = 0; d < 6; ++d) H[d + 6 * d] = H[d + 6 * d] + 1000.0f;     // aligner.cpp:94
  float nb[6], dx[6];
#pragma unroll
  for (int d = 0; d < 6; ++d) nb[d] = -b[d];
  ldlt_solve6(H, nb, dx);
  Mat4 invT = st.invT;
  set_last_row(invT);
  invT = iso_mul(v2t(dx), invT);
  if (outerEnd) {
    Mat4 T = iso_inverse(invT);
    float v[6];
    t2v(T, v);
    T = v2t(v);
    set_last_row(T);
    st.T = T;
    if (so && last) { so->T = T; so->it = it + 1; }
    const Mat4 Tinv = iso_inverse(T);
    st.invTcorrPrev = st.invTcorr;
    st.invTcorr = Tinv;
    invT = Tinv;
    Mat4 KRt, iKRt; Mat3 iK;
    projector_matrices(ap.K, iso_mul(T, ap.refOffset), KRt, iKRt, iK);
    st.KRtLast = st.KRt;        // the reference projection of this outer iteration (what the finder's depth image belongs to)
    st.KRt = KRt;
  }
  set_last_row(invT);
  st.invT = invT;
}
// The result record of a pair as it travels between ranks (include/pwn_hip.h: PWN_HIP_RECORD_FLOATS): 64 floats =
//   [0:16] T column-major  [16] chi2 of the last iteration  [17] its inliers  [18] iterations  [19] the caller's pair id
//   [20:30] chi2_i  [30:40] inliers_i  [40:50] C_i  [50:60] K_i (first 10 iterations)  [60] M_ref  [61] M_cur  [62] iterations in the traces
//   [63] 0, or 1 when a projection of the call gave up on a pixel (z32_settle): the library repeats such a call and the records it returns are the
//        repeat's, but a reader that takes the records off the device BEFORE the call has returned (pwn_hip_ctx_set_enqueued_callback) sees the
//        first attempt's and must drop them
// written on the device from the pair's state, so that a gather of records needs no trip through the host.  Counts travel as float
// (exact below 2^24).  grid = pairs, block = 64
// match != nullptr: a record has kMatchRecordFloats words, [64:68] = PwnMatcherBase::MatcherResult's image fields of the pair (nonZeros, outliers,
// inliers, reprojection distance: pwn_matcher_base.cpp:175-181) from its MatchAcc, [68:72] = 0.  block = 64, or 128 for the long records.
constexpr int kRecordFloats = 64, kRecordTrace = 10, kMatchRecordFloats = 72;
__global__ void __launch_bounds__(128) k_pack_records(const PairDesc* __restrict__ pairs, const int* __restrict__ pair_ids, int first_id, float* __restrict__ out,
                                                      const MatchAcc* __restrict__ match) {
  const PairDesc& pd = pairs[blockIdx.x];
  const PairState& st = *pd.state;
  const int t = threadIdx.x, it = st.it, m = it < kRecordTrace ? it : kRecordTrace;
  const int len = match ? kMatchRecordFloats : kRecordFloats;
  if (t >= len) return;
  float v = 0.f;
  if (t >= kRecordFloats) {
    const MatchAcc a = match[blockIdx.x];
    if (t == 64) v = (float)(int)a.nonZeros;
    else if (t == 65) v = (float)((int)a.nonZeros - (int)a.inliers);
    else if (t == 66) v = (float)(int)a.inliers;
    else if (t == 67) v = match_reprojection_distance(a);
  }
  else if (t < 16) v = st.T.m[t];
  else if (t == 16) v = it > 0 ? st.chi2[it - 1] : 0.f;
  else if (t == 17) v = it > 0 ? (float)st.inliers[it - 1] : 0.f;
  else if (t == 18) v = (float)it;
  else if (t == 19) v = (float)(pair_ids ? pair_ids[blockIdx.x] : first_id + (int)blockIdx.x);
  else if (t < 30) v = (t - 20 < m) ? st.chi2[t - 20] : 0.f;
  else if (t < 40) v = (t - 30 < m) ? (float)st.inliers[t - 30] : 0.f;
  else if (t < 50) v = (t - 40 < m) ? (float)st.ncorr[t - 40] : 0.f;
  else if (t < 60) v = (t - 50 < m) ? (float)st.ncand[t - 50] : 0.f;
  else if (t == 60) v = (float)*pd.ref.count;
  else if (t == 61) v = (float)*pd.cur.count;
  else if (t == 62) v = (float)m;
  else if (t == 63) v = (float)__hip_atomic_load(pd.fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // 1 = the call's fault word is up (see the layout above)
  out[(size_t)blockIdx.x * len + t] = v;
}
// reduction only, one record per pair (Aligner::_computeStatistics' 11th update).  grid = pairs, block = 256
__global__ void __launch_bounds__(256) k_reduce_pairs(const PairDesc* __restrict__ pairs, int nblocks, SolveOut* __restrict__ out) {
  __shared__ double sums[kAccN];
  reduce_partials(pairs[blockIdx.x].partials, nblocks, sums);
  if (threadIdx.x != 0) return;
  SolveOut* o = out + blockIdx.x;
  assemble_Hb(sums, o->H, o->b);
  o->chi2 = (float)sums[33]; o->inliers = (int)sums[34]; o->ncorr = (int)sums[35]; o->ncand = (int)sums[36];
}
// reduction only (pwn_hip_linearize)
__global__ void __launch_bounds__(256) k_reduce_only(const double* __restrict__ partials, int nblocks, SolveOut* __restrict__ out) {
  __shared__ double sums[kAccN];
  reduce_partials(partials, nblocks, sums);
  if (threadIdx.x != 0) return;
  assemble_Hb(sums, out->H, out->b);
  out->chi2 = (float)sums[33]; out->inliers = (int)sums[34]; out->ncorr = (int)sums[35]; out->ncand = (int)sums[36];
}

}  // namespace pwnhip
