// How long does a kernel boundary take on this machine?  Chains of dependent launches in one stream:
//   (a) empty kernel, 1 block;  (b) empty kernel, 300 blocks x 256;  (c) 1 block that spins ~8 us of dependent arithmetic (a stand-in for
//   k_solve_update) followed by (b) -- against one launch of 300 blocks in which block 0 does the arithmetic first.
// hipcc --offload-arch=gfx950 -O3 tools/micro/launch_chain.hip -o /tmp/launch_chain && /tmp/launch_chain
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_empty(int* p) { if (p && threadIdx.x == 12345) *p = 1; }
__global__ void k_serial(float* out, int n) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float x = out[0];
  for (int i = 0; i < n; ++i) x = x * 1.0000001f + 0.5f;      // dependent chain
  out[0] = x;
}
__global__ void k_serial_then_wide(float* out, int n) {
  if (blockIdx.x == 0 && threadIdx.x == 0) { float x = out[0]; for (int i = 0; i < n; ++i) x = x * 1.0000001f + 0.5f; out[0] = x; }
}
int main() {
  hipStream_t s; hipStreamCreate(&s);
  float* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
  auto run = [&](const char* name, int reps, auto body) {
    for (int i = 0; i < 20; ++i) body();
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) body();
    hipStreamSynchronize(s);
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    std::printf("%-62s %8.2f us per iteration\n", name, us);
  };
  run("empty kernel, 1 block", 2000, [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(256), 0, s, (int*)nullptr); });
  run("empty kernel, 300 blocks", 2000, [&] { hipLaunchKernelGGL(k_empty, dim3(300), dim3(256), 0, s, (int*)nullptr); });
  for (int n : { 500, 1000, 2000 }) {
    char nm[128];
    std::snprintf(nm, sizeof nm, "serial(%d) in its own launch + empty 300 blocks", n);
    run(nm, 1000, [&] { hipLaunchKernelGGL(k_serial, dim3(1), dim3(256), 0, s, d, n); hipLaunchKernelGGL(k_empty, dim3(300), dim3(256), 0, s, (int*)nullptr); });
    std::snprintf(nm, sizeof nm, "serial(%d) inside block 0 of the 300-block launch", n);
    run(nm, 1000, [&] { hipLaunchKernelGGL(k_serial_then_wide, dim3(300), dim3(256), 0, s, d, n); });
  }
  // the same dependent chains as ONE hipGraph launch (30 kernel nodes captured from the stream): does a graph shorten the kernel boundary?
  for (int blocks : { 1, 300 }) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 30; ++i) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, s, (int*)nullptr);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    char nm[128];
    std::snprintf(nm, sizeof nm, "graph of 30 empty kernels, %d block(s): per kernel", blocks);
    for (int i = 0; i < 5; ++i) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    const int reps = 200;
    for (int i = 0; i < reps; ++i) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps / 30;
    std::printf("%-62s %8.2f us\n", nm, us);
    // one graph launch at a time, waited for (what an alignment does): per kernel
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) { hipGraphLaunch(ge, s); hipStreamSynchronize(s); }
    us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    std::printf("  ... one graph launch + wait: %8.2f us (= %.2f per kernel)\n", us, us / 30);
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) { for (int k = 0; k < 30; ++k) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, s, (int*)nullptr); hipStreamSynchronize(s); }
    us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    std::printf("  ... 30 stream launches + wait:  %8.2f us (= %.2f per kernel)\n", us, us / 30);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  return 0;
}