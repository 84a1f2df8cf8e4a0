// Probe (run on the GPU box): host cost of enqueuing a 7-kernel dependent chain as 7 launches against one hipGraphLaunch of the same chain,
// and of a 4-branch / 12-kernel step with events against one graph launch of the captured step.  Kernels spin ~5 us.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/graph_launch_cost.hip -o /tmp/graph_launch_cost && /tmp/graph_launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin(int* p, long long cycles) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (p && threadIdx.x == 0) p[blockIdx.x] += 1;
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  int* d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
  hipStream_t s[4]; for (auto& x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
  hipEvent_t ev[8]; for (auto& evk : ev) CK(hipEventCreateWithFlags(&evk, hipEventDisableTiming));
  const long long cyc = 500;   // 100 MHz wall clock: 5 us
  const int reps = 300;
  // (a) chain of 7 by launches
  for (int pass = 0; pass < 2; pass++) {
    CK(hipDeviceSynchronize());
    const double t0 = now();
    for (int r = 0; r < reps; r++) for (int k = 0; k < 7; k++) hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s[0], d, cyc);
    const double t1 = now();
    CK(hipDeviceSynchronize());
    const double t2 = now();
    if (pass) std::printf("chain of 7, launches: host %.2f us per chain, wall %.2f us per chain\n", (t1 - t0) / reps, (t2 - t0) / reps);
  }
  // (b) the same chain as a graph
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
  for (int k = 0; k < 7; k++) hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s[0], d, cyc);
  CK(hipStreamEndCapture(s[0], &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int pass = 0; pass < 2; pass++) {
    CK(hipDeviceSynchronize());
    const double t0 = now();
    for (int r = 0; r < reps; r++) CK(hipGraphLaunch(ge, s[0]));
    const double t1 = now();
    CK(hipDeviceSynchronize());
    const double t2 = now();
    if (pass) std::printf("chain of 7, one graph launch: host %.2f us per chain, wall %.2f us per chain\n", (t1 - t0) / reps, (t2 - t0) / reps);
  }
  // (c) a 4-stream step: s1: 7 kernels; s0: wait(s1 prev) 2 kernels; s2: wait(s0) 1 kernel; s3: wait(s2) 2 kernels — by launches and events
  auto step = [&](hipStream_t* q) -> int {
    CK(hipStreamWaitEvent(q[0], ev[0], 0));
    for (int k = 0; k < 2; k++) hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, q[0], d, cyc);
    CK(hipEventRecord(ev[1], q[0]));
    for (int k = 0; k < 7; k++) hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, q[1], d, cyc);
    CK(hipEventRecord(ev[0], q[1]));
    CK(hipStreamWaitEvent(q[2], ev[1], 0));
    hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, q[2], d, cyc);
    CK(hipEventRecord(ev[2], q[2]));
    CK(hipStreamWaitEvent(q[3], ev[2], 0));
    for (int k = 0; k < 2; k++) hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, q[3], d, cyc);
    CK(hipEventRecord(ev[3], q[3]));
    return 0;
  };
  CK(hipEventRecord(ev[0], s[1]));
  for (int pass = 0; pass < 2; pass++) {
    CK(hipDeviceSynchronize());
    const double t0 = now();
    for (int r = 0; r < reps; r++) if (step(s)) return 1;
    const double t1 = now();
    CK(hipDeviceSynchronize());
    const double t2 = now();
    if (pass) std::printf("4-stream step (12 kernels, 4 records, 3 waits), calls: host %.2f us per step, wall %.2f us per step\n", (t1 - t0) / reps, (t2 - t0) / reps);
  }
  // (d) the same step captured (fork / join through events) and launched as one graph
  hipGraph_t g2; hipGraphExec_t ge2;
  CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
  CK(hipEventRecord(ev[4], s[0]));
  CK(hipStreamWaitEvent(s[1], ev[4], 0));
  for (int k = 0; k < 2; k++) hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s[0], d, cyc);
  CK(hipEventRecord(ev[5], s[0]));
  for (int k = 0; k < 7; k++) hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s[1], d, cyc);
  CK(hipStreamWaitEvent(s[2], ev[5], 0));
  hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s[2], d, cyc);
  CK(hipEventRecord(ev[6], s[2]));
  CK(hipStreamWaitEvent(s[3], ev[6], 0));
  for (int k = 0; k < 2; k++) hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s[3], d, cyc);
  CK(hipEventRecord(ev[7], s[3]));
  CK(hipEventRecord(ev[6], s[1]));
  CK(hipStreamWaitEvent(s[0], ev[7], 0));
  CK(hipStreamWaitEvent(s[0], ev[6], 0));
  CK(hipStreamEndCapture(s[0], &g2));
  CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
  for (int pass = 0; pass < 2; pass++) {
    CK(hipDeviceSynchronize());
    const double t0 = now();
    for (int r = 0; r < reps; r++) CK(hipGraphLaunch(ge2, s[0]));
    const double t1 = now();
    CK(hipDeviceSynchronize());
    const double t2 = now();
    if (pass) std::printf("4-branch step as one graph launch: host %.2f us per step, wall %.2f us per step\n", (t1 - t0) / reps, (t2 - t0) / reps);
  }
  // (e) two host threads, each enqueuing the 7-chain on its own stream
  std::printf("done\n");
  return 0;
}
