// Micro-benchmark: store efficiency of the integral-image writer's pattern.  A workgroup owns a W-column strip of a frame and walks it top
// to bottom in 16-row bands, writing 10 planes: per band 10 x 16 row pieces of W floats (row stride = cols floats, plane stride = N).
// W = 64 is k_unproject_integral's shape (256-byte pieces); wider strips write longer contiguous pieces.  Same bytes in every variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ROWS = 480, COLS = 640, N = ROWS * COLS, K = 10, FRAMES = 64;
template <int W> __global__ void __launch_bounds__(256) k(float* base) {
  constexpr int S = COLS / W;                       // strips per frame
  const int f = blockIdx.x / S, s = blockIdx.x % S;
  float* fr = base + (size_t)f * K * N;
  for (int band = 0; band < ROWS / 16; ++band) {
    // 256 threads: (row-in-band, column) pieces; each thread stores 10 planes x (16 * W / 256) elements
    for (int e = threadIdx.x; e < 16 * W; e += 256) {
      const int r = band * 16 + e / W, c = s * W + e % W;
#pragma unroll
      for (int k2 = 0; k2 < K; ++k2) fr[(size_t)k2 * N + (size_t)r * COLS + c] = (float)(e + k2);
    }
  }
}
int main() {
  float* buf; CK(hipMalloc(&buf, sizeof(float) * (size_t)N * K * FRAMES));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const double gb = 4.0 * N * K * FRAMES / 1e9;
  for (int v = 0; v < 4; ++v) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipEventRecord(a));
      if (v == 0) hipLaunchKernelGGL(k<64>, dim3(FRAMES * COLS / 64), dim3(256), 0, 0, buf);
      if (v == 1) hipLaunchKernelGGL(k<128>, dim3(FRAMES * COLS / 128), dim3(256), 0, 0, buf);
      if (v == 2) hipLaunchKernelGGL(k<320>, dim3(FRAMES * COLS / 320), dim3(256), 0, 0, buf);
      if (v == 3) hipLaunchKernelGGL(k<640>, dim3(FRAMES * COLS / 640), dim3(256), 0, 0, buf);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (rep > 0 && ms < best) best = ms;
    }
    const int W[4] = { 64, 128, 320, 640 };
    printf("strip width %3d (%4d-byte pieces, %4d workgroups): %.1f us  %.2f TB/s\n", W[v], 4 * W[v], FRAMES * COLS / W[v], best * 1e3, gb / best);
  }
  return 0;
}
