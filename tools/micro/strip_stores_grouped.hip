// Micro-benchmark: the integral-image writer's store pattern with the ten channels as ten dword planes (k_unproject_integral today) against
// three interleaved groups per pixel (float4, float4, float2).  Same bytes, 8-row bands, 64-column strips, 64 VGA frames.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ROWS = 480, COLS = 640, N = ROWS * COLS, K = 10, FRAMES = 64, W = 64, BR = 8;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE, bool NT> __global__ void __launch_bounds__(256) k(float* base) {
  constexpr int S = COLS / W;
  const int f = blockIdx.x / S, s = blockIdx.x % S;
  float* fr = base + (size_t)f * K * N;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int band = 0; band < ROWS / BR; ++band) {
    const int r0 = band * BR, c = s * W + lane;
    if (MODE == 0) {            // (channel, column) chains: 640 of them over 256 threads, 8 dword stores each
      for (int q = threadIdx.x; q < K * W; q += 256) {
        const int ch = q >> 6;
        float* dst = fr + (size_t)ch * N + (size_t)r0 * COLS + c;
        for (int r = 0; r < BR; ++r) { if (NT) __builtin_nontemporal_store((float)(q + r), dst + r * COLS); else dst[r * COLS] = (float)(q + r); }
      }
    } else {                    // (group, column) tasks: 192 of them, 8 wide stores each
      if (wv < 3) {
        for (int r = 0; r < BR; ++r) {
          const size_t pix = (size_t)(r0 + r) * COLS + c;
          if (wv < 2) { v4f v; v.x = v.y = v.z = v.w = (float)(lane + r); v4f* dst = (v4f*)(fr + (size_t)wv * 4 * N) + pix; if (NT) __builtin_nontemporal_store(v, dst); else *dst = v; }
          else { v2f v; v.x = v.y = (float)(lane + r); v2f* dst = (v2f*)(fr + (size_t)8 * N) + pix; if (NT) __builtin_nontemporal_store(v, dst); else *dst = v; }
        }
      }
    }
  }
}
int main() {
  float* buf; CK(hipMalloc(&buf, sizeof(float) * (size_t)N * K * FRAMES));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const double gb = 4.0 * N * K * FRAMES / 1e9;
  const char* names[4] = { "ten dword planes", "ten dword planes, non-temporal", "float4 + float4 + float2 groups", "groups, non-temporal" };
  for (int v = 0; v < 4; ++v) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipEventRecord(a));
      const dim3 g(FRAMES * COLS / W), bl(256);
      if (v == 0) hipLaunchKernelGGL((k<0, false>), g, bl, 0, 0, buf);
      if (v == 1) hipLaunchKernelGGL((k<0, true>), g, bl, 0, 0, buf);
      if (v == 2) hipLaunchKernelGGL((k<1, false>), g, bl, 0, 0, buf);
      if (v == 3) hipLaunchKernelGGL((k<1, true>), g, bl, 0, 0, buf);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (rep > 0 && ms < best) best = ms;
    }
    printf("%-36s %.1f us  %.2f TB/s\n", names[v], best * 1e3, gb / best);
  }
  return 0;
}
