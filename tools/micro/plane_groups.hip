// Micro-benchmark: four "corner" reads of the 10 integral-image sums per pixel (k_stats' access pattern: lanes along x, the four corners
// at +-R rows / columns) from (a) 10 separate float planes (40 dword loads per pixel) and (b) three channel groups stored interleaved per
// pixel: float4, float4, float2 (8 x dwordx4 + 4 x dwordx2 loads per pixel).  Same bytes, wider accesses.
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
int main() {
  float* buf; float* out;
  CK(hipMalloc(&buf, sizeof(float) * (size_t)N * 10 * FRAMES)); CK(hipMalloc(&out, sizeof(float) * (size_t)N * FRAMES));
  CK(hipMemset(buf, 0, sizeof(float) * (size_t)N * 10 * FRAMES));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const dim3 g((COLS + 255) / 256, ROWS, FRAMES);
  for (int v = 0; v < 2; ++v) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipEventRecord(a));
      if (v == 0) hipLaunchKernelGGL(k_planes, g, dim3(256), 0, 0, (const float*)buf, out); else hipLaunchKernelGGL(k_groups, g, dim3(256), 0, 0, (const float*)buf, out);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (rep > 0 && ms < best) best = ms;
    }
    printf("%s: %.1f us per 64 frames (compulsory 12.3 MB planes + 1.2 MB out per frame: %.2f TB/s)\n", v == 0 ? "10 planes, 40 dword loads / pixel      " : "3 interleaved groups, 12 wide loads / px", best * 1e3,
           (44.0 * N * FRAMES) / 1e9 / best);
  }
  return 0;
}
