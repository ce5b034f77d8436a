// Micro-benchmark: four "corner" reads of the 10 integral-image sums per pixel (k_stats' access pattern: lanes along x, the four corners
// at +-R rows / columns) from (a) 10 separate float planes (40 dword loads per pixel) and (b) three channel groups stored interleaved per
// pixel: float4, float4, float2 (8 x dwordx4 + 4 x dwordx2 loads per pixel).  Same bytes, wider accesses.
// Round 4: (c) the 40 dword loads with a per-pixel radius 10..30 (the real kernel's range, VGA configuration) and (d) north_star's "LDS-staged window
// tiles": a 1024-thread workgroup per 32-row x 64-column tile stages, channel by channel (double-buffered), the (32 + 62) x 128 plane values its
// pixels' windows can reach (radius <= 30 + the -1 of getRegion) into LDS with coalesced 512-byte rows and takes the four corners from there.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ROWS = 480, COLS = 640, N = ROWS * COLS, FRAMES = 64, R = 12;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__global__ void __launch_bounds__(256) k_planes(const float* __restrict__ base, float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y, f = blockIdx.z;
  if (c >= COLS) return;
  const float* fr = base + (size_t)f * 10 * N;
  const int x0 = clampi(c - R - 1, 0, COLS - 1), x1 = clampi(c + R - 1, 0, COLS - 1), y0 = clampi(r - R - 1, 0, ROWS - 1), y1 = clampi(r + R - 1, 0, ROWS - 1);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 10; ++k) { const float* p = fr + (size_t)k * N; s += ((p[y1 * COLS + x1] + p[y0 * COLS + x0]) - p[y1 * COLS + x0]) - p[y0 * COLS + x1]; }
  out[((size_t)f * ROWS + r) * COLS + c] = s;
}
__global__ void __launch_bounds__(256) k_groups(const float* __restrict__ base, float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y, f = blockIdx.z;
  if (c >= COLS) return;
  const float* fr = base + (size_t)f * 10 * N;
  const v4f* A = (const v4f*)fr; const v4f* B = (const v4f*)(fr + 4 * (size_t)N); const v2f* C = (const v2f*)(fr + 8 * (size_t)N);
  const int x0 = clampi(c - R - 1, 0, COLS - 1), x1 = clampi(c + R - 1, 0, COLS - 1), y0 = clampi(r - R - 1, 0, ROWS - 1), y1 = clampi(r + R - 1, 0, ROWS - 1);
  const int i11 = y1 * COLS + x1, i00 = y0 * COLS + x0, i10 = y1 * COLS + x0, i01 = y0 * COLS + x1;
  const v4f a = ((A[i11] + A[i00]) - A[i10]) - A[i01];
  const v4f b = ((B[i11] + B[i00]) - B[i10]) - B[i01];
  const v2f d = ((C[i11] + C[i00]) - C[i10]) - C[i01];
  out[((size_t)f * ROWS + r) * COLS + c] = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + d.x + d.y;
}
#ifndef RADIUS_PER_LANE
#define RADIUS_PER_LANE 0
#endif
// 10..30: the depth-dependent radius is smooth in the image (a wave's 64 neighbouring pixels share it, or differ by one, except across a depth edge);
// RADIUS_PER_LANE=1 makes it jump from lane to lane instead (every lane's corners in different rows: the worst case, 25 x slower for the gathers)
__device__ __forceinline__ int radius_of(int r, int c) { return RADIUS_PER_LANE ? 10 + (3 * r + 5 * c) % 21 : 10 + ((r >> 4) + (c >> 6) + ((c & 63) > 40 ? 1 : 0)) % 21; }
__global__ void __launch_bounds__(256) k_planes_var(const float* __restrict__ base, float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y, f = blockIdx.z;
  if (c >= COLS) return;
  const float* fr = base + (size_t)f * 10 * N;
  const int R2 = radius_of(r, c);
  const int x0 = clampi(c - R2 - 1, 0, COLS - 1), x1 = clampi(c + R2 - 1, 0, COLS - 1), y0 = clampi(r - R2 - 1, 0, ROWS - 1), y1 = clampi(r + R2 - 1, 0, ROWS - 1);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 10; ++k) { const float* p = fr + (size_t)k * N; s += ((p[y1 * COLS + x1] + p[y0 * COLS + x0]) - p[y1 * COLS + x0]) - p[y0 * COLS + x1]; }
  out[((size_t)f * ROWS + r) * COLS + c] = s;
}
constexpr int TR = 32, TC = 64, HALO_LO = 31, HALO_HI = 29, TROWS = TR + HALO_LO + HALO_HI + 2 /* 94 */, TCOLS = 128;
__global__ void __launch_bounds__(1024) k_tiles(const float* __restrict__ base, float* __restrict__ out) {
  extern __shared__ float tile[];                                   // [2][TROWS][TCOLS]
  const int c0 = blockIdx.x * TC, r0 = blockIdx.y * TR, f = blockIdx.z;
  const float* fr = base + (size_t)f * 10 * N;
  const int tid = threadIdx.x;
  const int ty0 = r0 - HALO_LO - 1, tx0 = c0 - HALO_LO - 1;         // image coordinates of tile cell (0, 0)
  // the thread's two pixels: (r0 + tid / 64, c0 + tid % 64) and 16 rows below
  const int pc = c0 + (tid & 63), pr[2] = { r0 + (tid >> 6), r0 + (tid >> 6) + 16 };
  int o11[2], o00[2], o10[2], o01[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int R2 = radius_of(pr[q], pc);
    const int x0 = clampi(pc - R2 - 1, 0, COLS - 1) - tx0, x1 = clampi(pc + R2 - 1, 0, COLS - 1) - tx0;
    const int y0 = clampi(pr[q] - R2 - 1, 0, ROWS - 1) - ty0, y1 = clampi(pr[q] + R2 - 1, 0, ROWS - 1) - ty0;
    o11[q] = y1 * TCOLS + x1; o00[q] = y0 * TCOLS + x0; o10[q] = y1 * TCOLS + x0; o01[q] = y0 * TCOLS + x1;
  }
  auto stage = [&](int k, int buf) {
    const float* p = fr + (size_t)k * N;
    float* t = tile + buf * TROWS * TCOLS;
    for (int i = tid; i < TROWS * TCOLS; i += 1024) {             // 12 coalesced loads per thread
      const int ty = i / TCOLS, tx = i % TCOLS;
      const int y = clampi(ty0 + ty, 0, ROWS - 1), x = clampi(tx0 + tx, 0, COLS - 1);
      t[i] = p[y * COLS + x];
    }
  };
  float s[2] = { 0.f, 0.f };
  stage(0, 0);
  __syncthreads();
  for (int k = 0; k < 10; ++k) {
    if (k + 1 < 10) stage(k + 1, (k + 1) & 1);
    const float* t = tile + (k & 1) * TROWS * TCOLS;
#pragma unroll
    for (int q = 0; q < 2; ++q) s[q] += ((t[o11[q]] + t[o00[q]]) - t[o10[q]]) - t[o01[q]];
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) if (pr[q] < ROWS && pc < COLS) out[((size_t)f * ROWS + pr[q]) * COLS + pc] = s[q];
}
int main() {
  float* buf; float* out;
  CK(hipMalloc(&buf, sizeof(float) * (size_t)N * 10 * FRAMES)); CK(hipMalloc(&out, sizeof(float) * (size_t)N * FRAMES));
  CK(hipMemset(buf, 0, sizeof(float) * (size_t)N * 10 * FRAMES));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const dim3 g((COLS + 255) / 256, ROWS, FRAMES);
  CK(hipFuncSetAttribute((const void*)k_tiles, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TROWS * TCOLS * 4));
  for (int v = 0; v < 4; ++v) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipEventRecord(a));
      if (v == 0) hipLaunchKernelGGL(k_planes, g, dim3(256), 0, 0, (const float*)buf, out);
      else if (v == 1) hipLaunchKernelGGL(k_groups, g, dim3(256), 0, 0, (const float*)buf, out);
      else if (v == 2) hipLaunchKernelGGL(k_planes_var, g, dim3(256), 0, 0, (const float*)buf, out);
      else hipLaunchKernelGGL(k_tiles, dim3(COLS / TC, ROWS / TR, FRAMES), dim3(1024), 2 * TROWS * TCOLS * 4, 0, (const float*)buf, out);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (rep > 0 && ms < best) best = ms;
    }
    const char* names[4] = { "10 planes, 40 dword loads / pixel, R = 12   ", "3 interleaved groups, 12 wide loads / px   ", "10 planes, 40 dword loads / px, R = 10..30 ", "LDS-staged 32x64 window tiles, R = 10..30  " };
    printf("%s: %.1f us per 64 frames (compulsory 12.3 MB planes + 1.2 MB out per frame: %.2f TB/s)\n", names[v], best * 1e3,
           (44.0 * N * FRAMES) / 1e9 / best);
  }
  return 0;
}
