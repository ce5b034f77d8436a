// Micro-benchmark: does the stride between the SoA planes matter (HBM channel aliasing)?  Every thread touches the same element
// of K planes (integral image: 10 planes written by k_unproject_integral / read by k_stats; cloud: 9 information-matrix planes).
// Strides: N floats (= 75 * 16 KiB at VGA) against N + pad.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int N = 307200, FRAMES = 64, K = 10;
__global__ void __launch_bounds__(256) k_write(float* base, size_t stride, size_t frameStride) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  float* p = base + blockIdx.y * frameStride + i;
#pragma unroll
  for (int k = 0; k < K; ++k) p[k * stride] = (float)(i + k);
}
__global__ void __launch_bounds__(256) k_read(const float* base, size_t stride, size_t frameStride, float* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float* p = base + blockIdx.y * frameStride + i;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) s += p[k * stride];
  if (s == 12345.678f) out[i] = s;
}
int main() {
  const size_t maxStride = N + 4096;
  float* buf; float* out;
  CK(hipMalloc(&buf, sizeof(float) * maxStride * K * FRAMES)); CK(hipMalloc(&out, sizeof(float) * N));
  CK(hipMemset(buf, 0, sizeof(float) * maxStride * K * FRAMES));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int pads[] = { 0, 64, 96, 160, 544, 1056, 2080, 4096 };
  for (int pad : pads) {
    const size_t stride = N + pad, fs = stride * K;
    float bw = 1e9f, br = 1e9f;
    for (int rep = 0; rep < 8; ++rep) {
      CK(hipEventRecord(a)); hipLaunchKernelGGL(k_write, dim3(N / 256, FRAMES), dim3(256), 0, 0, buf, stride, fs); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (rep > 1 && ms < bw) bw = ms;
      CK(hipEventRecord(a)); hipLaunchKernelGGL(k_read, dim3(N / 256, FRAMES), dim3(256), 0, 0, (const float*)buf, stride, fs, out); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      CK(hipEventElapsedTime(&ms, a, b)); if (rep > 1 && ms < br) br = ms;
    }
    const double gb = 4.0 * N * K * FRAMES / 1e9;
    printf("plane stride N + %4d floats: write %.1f us (%.2f TB/s)   read %.1f us (%.2f TB/s)\n", pad, bw * 1e3, gb / bw, br * 1e3, gb / br);
  }
  return 0;
}
