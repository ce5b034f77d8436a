// What would a one-launch alignment (Aligner::align's ten Gauss-Newton iterations in ONE cooperative kernel, aligner.cpp:66-118) gain over the
// chain of launches the library issues today?  Two measurements on one VGA pair's shape (150 workgroups x 1024 threads, one per CU):
//
//  1. the cost of a grid-wide barrier alone (agent-scope release, one atomic arrive, poll of a generation word, agent-scope acquire) -- hand-written,
//     and hip's cooperative_groups grid.sync() next to it;
//  2. a SKELETON of one alignment with the real dependency structure and comparable memory behaviour, in both forms:
//       per iteration:  "project"  300 k atomicMin's into a 307 200-word z-buffer (2 per thread)
//                       "fused"    per thread 2 x (z-buffer word -> dependent 16-byte gather -> dependent 12-byte gather), 37 partial sums per workgroup
//                       "solve"    one thread reduces the 150 partial records in fixed order and runs a dependent chain of ~1 300 fp32 operations
//                                  (the 6x6 LDLt + SE(3) update of k_solve_update), writes 16 floats every workgroup reads in the next iteration
//     (a) 30 launches (project, fused, solve) x 10 in one stream -- what the library does;
//     (b) one cooperative launch, 3 grid barriers per iteration, workgroup 0 solves;
//     (c) one cooperative launch, 2 grid barriers per iteration, EVERY workgroup reduces and solves redundantly (no third barrier).
//   If (b) / (c) are not well ahead of (a) here, the real kernel (more registers, the bit-exact summation order, scratch for the 6x6 step) will not be.
//
// hipcc --offload-arch=gfx950 -O3 tools/micro/grid_barrier.hip -o /tmp/grid_barrier && /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <chrono>
#include <cstdio>
#include <vector>
namespace cg = cooperative_groups;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int kWG = 150, kThreads = 1024, kPix = 307200, kPts = 300000, kAcc = 37, kIters = 10, kChain = 1300;

struct Sync { unsigned count, gen; };
__device__ __forceinline__ void grid_barrier(Sync* s, unsigned nwg) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned g = __hip_atomic_load(&s->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned prev = __hip_atomic_fetch_add(&s->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == nwg - 1) {
      __hip_atomic_store(&s->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&s->gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(&s->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g) __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}
// ---- 1. barriers alone
__global__ void __launch_bounds__(kThreads) k_barriers(Sync* s, int n, int* sink) {
  for (int i = 0; i < n; ++i) grid_barrier(s, gridDim.x);
  if (threadIdx.x == 12345) *sink = 1;
}
__global__ void __launch_bounds__(kThreads) k_barriers_cg(int n, int* sink) {
  cg::grid_group g = cg::this_grid();
  for (int i = 0; i < n; ++i) g.sync();
  if (threadIdx.x == 12345) *sink = 1;
}
// ---- 2. the skeleton's phases
struct Bufs { unsigned* z; const float4* A; const float* B; double* partials; float* state; };
__device__ __forceinline__ void phase_project(const Bufs& b, unsigned tag, int wg, int nwg) {
  const int per = (kPts + nwg * kThreads - 1) / (nwg * kThreads);
  const float s0 = b.state[0];
  for (int j = 0; j < per; ++j) {
    const int i = (wg * per + j) * kThreads + threadIdx.x;
    if (i < kPts) {
      int pix = i + (i >> 5) + 3 + (int)(s0 * 0.f); pix = pix < kPix ? pix : pix - kPix;
      atomicMin(&b.z[pix], (tag << 21) | (unsigned)(i & 0x1FFFFF));
    }
  }
}
__device__ __forceinline__ void phase_fused(const Bufs& b, unsigned tag, int wg) {
  __shared__ float red[kThreads / 64][kAcc];
  float acc[kAcc];
#pragma unroll
  for (int k = 0; k < kAcc; ++k) acc[k] = 0.f;
  const float s1 = b.state[1];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int pix = (wg * 2 + j) * kThreads + threadIdx.x;
    if (pix < kPix) {
      const unsigned w = b.z[pix];
      if ((w >> 21) == tag) {
        const unsigned i = w & 0x1FFFFF;
        const float4 a = b.A[i];                                          // dependent gather 1 (normal + curvature)
        const unsigned i2 = (i + (unsigned)(a.w * 0.f)) % kPts;
        const float x = b.B[3 * i2], y = b.B[3 * i2 + 1], zz = b.B[3 * i2 + 2];   // dependent gather 2 (information-matrix row)
#pragma unroll
        for (int k = 0; k < kAcc; ++k) acc[k] += (a.x + x) * (a.y + y) + (a.z + zz) * (float)(k + 1) + s1;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < kAcc; ++k) {
    float v = acc[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < kAcc) {
    double s = 0.0;
    for (int w2 = 0; w2 < kThreads / 64; ++w2) s += (double)red[w2][threadIdx.x];
    b.partials[(size_t)wg * kAcc + threadIdx.x] = s;
  }
}
__device__ __forceinline__ void phase_solve(const Bufs& b, int nwg, bool write) {
  __shared__ double sums[kAcc];
  if (threadIdx.x < kAcc) {
    double s = 0.0;
    for (int w = 0; w < nwg; ++w) s += b.partials[(size_t)w * kAcc + threadIdx.x];
    sums[threadIdx.x] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float x = (float)sums[0] * 1e-9f + 1.0f;
    for (int i = 0; i < kChain; ++i) x = x * 1.0000001f + (float)sums[1 + (i % 36)] * 1e-12f;      // dependent fp32 chain
    if (write) for (int k = 0; k < 16; ++k) b.state[k] = x * 1e-9f * (float)k;
  }
  __syncthreads();
}
__global__ void __launch_bounds__(256) k_project(Bufs b, unsigned tag) { phase_project(b, tag, blockIdx.x * 256 / kThreads, kWG); }
// (a) separate launches: project with 256-thread blocks like the library's k_project<1>, fused with 1024, solve with one block
__global__ void __launch_bounds__(256) k_project_a(Bufs b, unsigned tag) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= kPts) return;
  int pix = i + (i >> 5) + 3 + (int)(b.state[0] * 0.f); pix = pix < kPix ? pix : pix - kPix;
  atomicMin(&b.z[pix], (tag << 21) | (unsigned)(i & 0x1FFFFF));
}
__global__ void __launch_bounds__(kThreads) k_fused_a(Bufs b, unsigned tag) { phase_fused(b, tag, blockIdx.x); }
__global__ void __launch_bounds__(256) k_solve_a(Bufs b) { phase_solve(b, kWG, true); }
// (b), (c) one cooperative launch
template <int MODE>
__global__ void __launch_bounds__(kThreads) k_persistent(Bufs b, Sync* s, unsigned tag0) {
  const int wg = blockIdx.x, nwg = gridDim.x;
  for (int it = 0; it < kIters; ++it) {
    const unsigned tag = tag0 - (unsigned)it;
    phase_project(b, tag, wg, nwg);
    grid_barrier(s, nwg);
    phase_fused(b, tag, wg);
    grid_barrier(s, nwg);
    if (MODE == 0) { if (wg == 0) phase_solve(b, nwg, true); grid_barrier(s, nwg); }
    else phase_solve(b, nwg, wg == 0);       // everybody computes the step; only workgroup 0 publishes it (the others keep it in registers in the real kernel)
  }
}
int main() {
  Bufs b; Sync* s; int* sink;
  CK(hipMalloc(&b.z, 4ull * kPix)); CK(hipMemset(b.z, 0xFF, 4ull * kPix));
  CK(hipMalloc((void**)&b.A, 16ull * kPts)); CK(hipMemset((void*)b.A, 0, 16ull * kPts));
  CK(hipMalloc((void**)&b.B, 12ull * kPts)); CK(hipMemset((void*)b.B, 0, 12ull * kPts));
  CK(hipMalloc(&b.partials, 8ull * kWG * kAcc)); CK(hipMemset(b.partials, 0, 8ull * kWG * kAcc));
  CK(hipMalloc(&b.state, 64)); CK(hipMemset(b.state, 0, 64));
  CK(hipMalloc(&s, sizeof(Sync))); CK(hipMemset(s, 0, sizeof(Sync)));
  CK(hipMalloc(&sink, 4));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  int coop = 0; CK(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, 0));
  int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_persistent<0>, kThreads, 0));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  std::printf("cooperative launch supported: %d; resident 1024-thread workgroups of the skeleton per CU: %d x %d CUs\n", coop, occ, prop.multiProcessorCount);
  auto timeit = [&](const char* name, int reps, auto body, double per) {
    for (int i = 0; i < 5; ++i) body();
    (void)hipStreamSynchronize(st);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) { body(); (void)hipStreamSynchronize(st); }
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    std::printf("%-78s %9.1f us  (%.2f us %s)\n", name, us, us / per, per == 1 ? "per call" : "per unit");
    return us;
  };
  // 1. barriers
  for (int n : { 0, 30, 300 }) {
    char nm[128]; std::snprintf(nm, sizeof nm, "cooperative launch + wait, %d hand-written grid barriers, 150 x 1024", n);
    void* args[] = { &s, &n, &sink };
    timeit(nm, 200, [&] { (void)hipLaunchCooperativeKernel((void*)k_barriers, dim3(kWG), dim3(kThreads), args, 0, st); }, n ? n : 1);
  }
  for (int n : { 30, 300 }) {
    char nm[128]; std::snprintf(nm, sizeof nm, "cooperative launch + wait, %d cooperative_groups grid.sync()", n);
    void* args[] = { &n, &sink };
    timeit(nm, 200, [&] { (void)hipLaunchCooperativeKernel((void*)k_barriers_cg, dim3(kWG), dim3(kThreads), args, 0, st); }, n);
  }
  // 2. skeleton
  unsigned tag0 = 0x7FE;
  auto next_tags = [&]() { if (tag0 < 64) { (void)hipMemsetAsync(b.z, 0xFF, 4ull * kPix, st); tag0 = 0x7FE; } const unsigned t = tag0; tag0 -= kIters; return t; };
  const double a = timeit("(a) 30 launches: (project 1172 x 256, fused 150 x 1024, solve 1 x 256) x 10", 300, [&] {
    const unsigned t = next_tags();
    for (int it = 0; it < kIters; ++it) {
      hipLaunchKernelGGL(k_project_a, dim3((kPts + 255) / 256), dim3(256), 0, st, b, t - it);
      hipLaunchKernelGGL(k_fused_a, dim3(kWG), dim3(kThreads), 0, st, b, t - it);
      hipLaunchKernelGGL(k_solve_a, dim3(1), dim3(256), 0, st, b);
    } }, kIters);
  const double bb = timeit("(b) one cooperative launch, 3 grid barriers per iteration, workgroup 0 solves", 300, [&] {
    unsigned t = next_tags(); void* args[] = { &b, &s, &t };
    (void)hipLaunchCooperativeKernel((void*)k_persistent<0>, dim3(kWG), dim3(kThreads), args, 0, st); }, kIters);
  const double c = timeit("(c) one cooperative launch, 2 grid barriers per iteration, every workgroup solves", 300, [&] {
    unsigned t = next_tags(); void* args[] = { &b, &s, &t };
    (void)hipLaunchCooperativeKernel((void*)k_persistent<1>, dim3(kWG), dim3(kThreads), args, 0, st); }, kIters);
  std::printf("one-launch form against the launch chain: (b) %+.1f %%, (c) %+.1f %%\n", 100.0 * (bb - a) / a, 100.0 * (c - a) / a);
  CK(hipGetLastError());
  return 0;
}
