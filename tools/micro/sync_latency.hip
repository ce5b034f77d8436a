// How long does the host take to notice that a stream has finished?  One short kernel that spins ~200 us on the device, then three ways to wait for it:
//   (a) hipStreamSynchronize   (b) hipEventSynchronize on an event recorded behind it   (c) a host loop over hipEventQuery
//   (d) the kernel's last act is a store to a page-locked host word the host polls (what k_solve_update does for a single alignment's pose)
// Reported: host time from the launch call to the moment the wait returns, minus the same for (d) as the floor.
// hipcc --offload-arch=gfx950 -O3 tools/micro/sync_latency.hip -o /tmp/sync_latency && /tmp/sync_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <algorithm>
#include <vector>
__global__ void k_spin(long long cycles, volatile int* flag, int v) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) { }
  if (flag) { __threadfence_system(); *flag = v; }
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  int* flag = nullptr; hipHostMalloc((void**)&flag, sizeof(int)); *flag = 0;
  const long long cycles = 20000;      // wall_clock64 ticks at 100 MHz: 200 us
  auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  const int reps = 200;
  std::vector<double> a, b, c, d;
  for (int i = 0; i < reps + 20; ++i) {
    double t0 = now_us();
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, cycles, (volatile int*)nullptr, 0);
    hipStreamSynchronize(s);
    if (i >= 20) a.push_back(now_us() - t0);
    t0 = now_us();
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, cycles, (volatile int*)nullptr, 0);
    hipEventRecord(ev, s); hipEventSynchronize(ev);
    if (i >= 20) b.push_back(now_us() - t0);
    t0 = now_us();
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, cycles, (volatile int*)nullptr, 0);
    hipEventRecord(ev, s); while (hipEventQuery(ev) == hipErrorNotReady) { }
    if (i >= 20) c.push_back(now_us() - t0);
    t0 = now_us();
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, cycles, (volatile int*)flag, i + 1);
    while (*(volatile int*)flag != i + 1) { }
    if (i >= 20) d.push_back(now_us() - t0);
    hipStreamSynchronize(s);
  }
  std::printf("launch + 200 us kernel + wait, median of %d (us): hipStreamSynchronize %.1f   hipEventSynchronize %.1f   hipEventQuery loop %.1f   host word polled %.1f\n",
              reps, med(a), med(b), med(c), med(d));
  return 0;
}
