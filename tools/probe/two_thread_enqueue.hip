// Probe (run on the GPU box): does the HIP runtime take launches from two host threads in parallel?  One step = 12 short kernels, 6 event records
// and 6 stream waits over four streams (the four-stream schedule's call mix).  (a) one thread enqueues everything; (b) thread A enqueues the
// streams 0 / 1 half, thread B the streams 2 / 3 half of the same step, B's first wait ordered behind A's record by a host flag.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/two_thread_enqueue.hip -o /tmp/tte -lpthread && /tmp/tte
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::abort(); } } while (0)
__global__ void spin(int* p, long long cycles) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (p && threadIdx.x == 0) p[blockIdx.x] += 1;
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static int* d;
static hipStream_t s[4];
static hipEvent_t evA[64], evB[64], evC[64];   // per-step events in rotation (a record must not overtake a wait on the same event object)
static const long long cyc = 200;              // 2 us kernels: the host is the limit
static void front(int i) {   // streams 0, 1: 7 kernels, 3 records, 3 waits
  const int k = i % 64;
  if (i >= 2) CK(hipStreamWaitEvent(s[1], evC[(i - 2) % 64], 0));
  for (int j = 0; j < 4; j++) hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s[1], d, cyc);
  CK(hipEventRecord(evA[k], s[1]));
  CK(hipStreamWaitEvent(s[0], evA[k], 0));
  for (int j = 0; j < 3; j++) hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s[0], d, cyc);
  CK(hipEventRecord(evB[k], s[0]));
}
static void back(int i) {    // streams 2, 3: 5 kernels, 3 records, 3 waits
  const int k = i % 64;
  CK(hipStreamWaitEvent(s[2], evB[k], 0));
  for (int j = 0; j < 2; j++) hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s[2], d, cyc);
  CK(hipEventRecord(evC[k], s[2]));
  CK(hipStreamWaitEvent(s[3], evC[k], 0));
  for (int j = 0; j < 3; j++) hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s[3], d, cyc);
  CK(hipEventRecord(evA[(k + 32) % 64], s[3]));
}
int main() {
  CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
  for (auto& x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
  for (int k = 0; k < 64; k++) { CK(hipEventCreateWithFlags(&evA[k], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&evB[k], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&evC[k], hipEventDisableTiming)); }
  const int reps = 2000;
  for (int pass = 0; pass < 2; pass++) {
    CK(hipDeviceSynchronize());
    const double t0 = now();
    for (int i = 0; i < reps; i++) { front(i); back(i); }
    const double t1 = now();
    CK(hipDeviceSynchronize());
    const double t2 = now();
    if (pass) std::printf("one thread : host %.2f us per step, wall %.2f us per step\n", (t1 - t0) / reps, (t2 - t0) / reps);
  }
  for (int pass = 0; pass < 2; pass++) {
    CK(hipDeviceSynchronize());
    std::atomic<int> done_front{0};
    const double t0 = now();
    std::thread tb([&] {
      CK(hipSetDevice(0));
      for (int i = 0; i < reps; i++) {
        while (done_front.load(std::memory_order_acquire) <= i) __builtin_ia32_pause();
        back(i);
      }
    });
    for (int i = 0; i < reps; i++) { front(i); done_front.store(i + 1, std::memory_order_release); }
    const double t1 = now();
    tb.join();
    const double t1b = now();
    CK(hipDeviceSynchronize());
    const double t2 = now();
    if (pass) std::printf("two threads: host A %.2f us per step, both %.2f us per step, wall %.2f us per step\n", (t1 - t0) / reps, (t1b - t0) / reps, (t2 - t0) / reps);
  }
  return 0;
}
