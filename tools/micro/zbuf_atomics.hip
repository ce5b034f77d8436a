// Micro-benchmark: what bounds the z-buffer scatter of k_project?  64 "pairs" x 300k points, point i lands on pixel ~ i (one
// wave covers ~64 consecutive pixels, like a projected organised cloud).  Variants:
//   A  64-bit atomicMin without return (the product kernel)
//   B  32-bit atomicMin without return (half the bytes per request)
//   C  32-bit load + atomicCAS with return (an index-only z-buffer would need this)
//   D  plain 64-bit store, E plain 32-bit store
//   F  32-bit atomicMin WITH return (the product kernel since round 2: the returned word tells a collision), 16-byte point loads
//   G  F with 12-byte point loads (the product layout), H = G with four points per thread (loads, then atomics, then the returned words)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int N = 307200, M = 298000, PAIRS = 64;
__device__ __forceinline__ int pixel_of(int i) { int p = i + (i >> 5) + 3; return p < N ? p : p - N; }   // ~3 % collisions, row drift
template <int V> __global__ void __launch_bounds__(256) k(const float4* __restrict__ P, unsigned long long* z64, unsigned* z32) {
  const int pair = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= M) return;
  const float4 p = P[(size_t)pair * M + i];
  const float d = p.z + 1.0f;
  const int pix = pixel_of(i);
  const unsigned long long key64 = ((unsigned long long)__float_as_uint(d) << 21) | (unsigned)i;
  const unsigned key32 = (__float_as_uint(d) & 0xFFFFF800u) | ((unsigned)i & 0x7FFu);
  if (V == 0) atomicMin(&z64[(size_t)pair * N + pix], key64);
  if (V == 1) atomicMin(&z32[(size_t)pair * N + pix], key32);
  if (V == 2) {
    unsigned* a = &z32[(size_t)pair * N + pix];
    unsigned w = *a;
    while (key32 < w) { const unsigned old = atomicCAS(a, w, key32); if (old == w) break; w = old; }
  }
  if (V == 3) z64[(size_t)pair * N + pix] = key64;
  if (V == 4) z32[(size_t)pair * N + pix] = key32;
  if (V == 5) { const unsigned old = atomicMin(&z32[(size_t)pair * N + pix], key32); if (old == 12345u) z32[0] = old; }
}
typedef float v3 __attribute__((ext_vector_type(3)));
typedef v3 v3a __attribute__((aligned(4)));
template <int PPT> __global__ void __launch_bounds__(256) k3(const float* __restrict__ P3, unsigned* z32) {
  const int pair = blockIdx.y, i0 = blockIdx.x * 256 * PPT + threadIdx.x;
  v3 p[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) { const int i = i0 + 256 * j; p[j] = *(const v3a*)(P3 + ((size_t)pair * M + (i < M ? i : M - 1)) * 3); }
  unsigned old[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int i = i0 + 256 * j;
    old[j] = 0u;
    if (i < M) {
      const float d = p[j].z + 1.0f;
      const unsigned key32 = (__float_as_uint(d) & 0xFFFFF800u) | ((unsigned)i & 0x7FFu);
      old[j] = atomicMin(&z32[(size_t)pair * N + pixel_of(i)], key32);
    }
  }
#pragma unroll
  for (int j = 0; j < PPT; ++j) if (old[j] == 12345u) z32[0] = old[j];
}
int main() {
  float4* P; unsigned long long* z64; unsigned* z32;
  CK(hipMalloc(&P, sizeof(float4) * (size_t)PAIRS * M)); CK(hipMalloc(&z64, 8ull * PAIRS * N)); CK(hipMalloc(&z32, 4ull * PAIRS * N));
  std::vector<float4> h((size_t)PAIRS * M);
  for (size_t i = 0; i < h.size(); ++i) h[i] = make_float4(0.f, 0.f, 1.0f + (float)(i % 977) * 1e-3f, 0.f);
  CK(hipMemcpy(P, h.data(), sizeof(float4) * h.size(), hipMemcpyHostToDevice));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const char* names[9] = { "A atomicMin u64 no-return", "B atomicMin u32 no-return", "C load + atomicCAS u32 (return)", "D plain store u64", "E plain store u32",
                           "F atomicMin u32 with return", "G F + 12-byte loads", "H G, 4 points per thread", "I G, 8 points per thread" };
  float* P3; CK(hipMalloc(&P3, sizeof(float) * 3 * (size_t)PAIRS * M)); CK(hipMemset(P3, 0, sizeof(float) * 3 * (size_t)PAIRS * M));
  for (int v = 0; v < 9; ++v) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipMemset(z64, 0xFF, 8ull * PAIRS * N)); CK(hipMemset(z32, 0xFF, 4ull * PAIRS * N)); CK(hipDeviceSynchronize());
      CK(hipEventRecord(a));
      dim3 g((M + 255) / 256, PAIRS);
      if (v == 0) hipLaunchKernelGGL(k<0>, g, dim3(256), 0, 0, P, z64, z32);
      if (v == 1) hipLaunchKernelGGL(k<1>, g, dim3(256), 0, 0, P, z64, z32);
      if (v == 2) hipLaunchKernelGGL(k<2>, g, dim3(256), 0, 0, P, z64, z32);
      if (v == 3) hipLaunchKernelGGL(k<3>, g, dim3(256), 0, 0, P, z64, z32);
      if (v == 4) hipLaunchKernelGGL(k<4>, g, dim3(256), 0, 0, P, z64, z32);
      if (v == 5) hipLaunchKernelGGL(k<5>, g, dim3(256), 0, 0, P, z64, z32);
      if (v == 6) hipLaunchKernelGGL(k3<1>, g, dim3(256), 0, 0, P3, z32);
      if (v == 7) hipLaunchKernelGGL(k3<4>, dim3((M + 1023) / 1024, PAIRS), dim3(256), 0, 0, P3, z32);
      if (v == 8) hipLaunchKernelGGL(k3<8>, dim3((M + 2047) / 2048, PAIRS), dim3(256), 0, 0, P3, z32);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (rep > 0 && ms < best) best = ms;
    }
    printf("%-34s %.1f us per 64-pair launch\n", names[v], best * 1e3f);
  }
  return 0;
}
