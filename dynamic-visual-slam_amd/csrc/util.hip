// util.hip — error text, device selection, raw device-memory helpers of the C-ABI
#include <stdlib.h>
#include <string.h>
#include "common.h"
#ifdef DVS_TEST_HOOKS
#include "../../include/dvslam_hip_test.h"
#endif

extern "C" char** environ;   // (POSIX)

namespace dvs {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- the environment switches of the library: ONE table (DESIGN.md section 4a' lists them with the test that covers each).  Every
// DVS_* variable of the environment must be one of these with an allowed value: a mistyped name or value fails the first handle
// creation loudly (DVS_ERR_ARG) instead of silently running the default.
namespace {
struct EnvSwitch { const char* name; int allowed[4]; int nallowed; };
const EnvSwitch kSwitches[] = {
    {"DVS_NO_OVERLAP", {0, 1}, 2},         // 1: every kernel alone on the handle's main stream
    {"DVS_CASCADE", {-1, 0, 1}, 3},        // pyramid by the launch chain / by k_pyr_cascade whatever the batch (-1: by batch size)
    {"DVS_CHAIN_GRAPH", {-1, 0, 1}, 3},    // the announced batch's level chain never / always as one graph launch (-1: by batch size)
    {"DVS_BLUR_MFMA", {0, 1, 2}, 3},       // matrix-core blur: LDS-tiled / LDS-free
    {"DVS_HOST_POLL", {0, 1}, 2},          // 0: dvs_orb_extract returns its results by copy commands
    {"DVS_OCT_T", {0, 256, 512}, 3},       // quad-tree workgroup size for every batch size (0: by batch size)
    {"DVS_DESC_ORDER", {0, 1}, 2},         // 0: the descriptor stage visits keypoints in list order
    {"DVS_FAST_BYTE_DMA", {0, 1}, 2},      // 0: FAST tiles staged from dword-aligned origins
    {"DVS_MATCH_MFMA", {0, 1}, 2},         // 0: popcount kernels for every job count
    {"DVS_MATCH_LDS", {0, 1}, 2},          // 0: up to 6 jobs by k_match<16, 1> instead of k_match_lds
    {"DVS_LM_POLL", {0, 1}, 2},            // 0: device LM without host polling
    {"DVS_LM_SPECULATE", {0, 1}, 2},       // 0: ... without the speculatively enqueued accepted-step launches
};
const EnvSwitch* find_switch(const char* name, size_t len) {
  for (const EnvSwitch& s : kSwitches)
    if (strlen(s.name) == len && strncmp(s.name, name, len) == 0) return &s;
  return nullptr;
}
}  // namespace

dvs_status env_check() {
  for (char** e = environ; e && *e; e++) {
    if (strncmp(*e, "DVS_", 4) != 0) continue;
    const char* eq = strchr(*e, '=');
    if (!eq) continue;
    const EnvSwitch* s = find_switch(*e, (size_t)(eq - *e));
    if (!s) {
      set_error("unknown environment switch %.*s (the library reads: DVS_NO_OVERLAP DVS_CASCADE DVS_CHAIN_GRAPH DVS_BLUR_MFMA DVS_HOST_POLL DVS_OCT_T "
                "DVS_DESC_ORDER DVS_FAST_BYTE_DMA DVS_MATCH_MFMA DVS_MATCH_LDS DVS_LM_POLL DVS_LM_SPECULATE)", (int)(eq - *e), *e);
      return DVS_ERR_ARG;
    }
    char* end = nullptr;
    const long v = strtol(eq + 1, &end, 10);
    bool ok = end != eq + 1 && *end == 0;
    if (ok) { ok = false; for (int i = 0; i < s->nallowed; i++) ok = ok || s->allowed[i] == (int)v; }
    if (!ok) {
      char list[64] = ""; size_t o = 0;
      for (int i = 0; i < s->nallowed; i++) o += (size_t)snprintf(list + o, sizeof(list) - o, "%s%d", i ? ", " : "", s->allowed[i]);
      set_error("%s=%s: not a value of this switch (allowed: %s)", s->name, eq + 1, list);
      return DVS_ERR_ARG;
    }
  }
  return DVS_OK;
}

int env_switch(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && find_switch(name, strlen(name)) ? atoi(v) : dflt;
}

dvs_status check_device(int device) {
  DVS_TRY(env_check());   // (before the device: a bad switch is an error with or without a GPU)
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    set_error("no HIP device visible (%s); libdvslam_hip has no CPU fallback", e == hipSuccess ? "count 0" : hipGetErrorString(e));
    (void)hipGetLastError();
    return DVS_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= n) { set_error("device %d out of range (0..%d)", device, n - 1); return DVS_ERR_ARG; }
  DVS_HIP(hipSetDevice(device));
  return DVS_OK;
}

void StageTimer::begin(int stage, hipStream_t s, bool count_call) {
  if (!on) return;
  if (npending >= kMaxPending) resolve();
  Pending& p = pending[npending];
  if (npool < 2 * (npending + 1)) {
    hipEventCreate(&pool[npool++]);
    hipEventCreate(&pool[npool++]);
  }
  p.a = pool[2 * npending]; p.b = pool[2 * npending + 1]; p.stage = stage; p.count = count_call;
  hipEventRecord(p.a, s);
  cur = npending++;
}
void StageTimer::end(hipStream_t s) {
  if (!on || cur < 0) return;
  hipEventRecord(pending[cur].b, s);
  cur = -1;
}
void StageTimer::resolve() {
  for (int i = 0; i < npending; i++) {
    hipEventSynchronize(pending[i].b);
    float t = 0;
    if (hipEventElapsedTime(&t, pending[i].a, pending[i].b) == hipSuccess) { ms[pending[i].stage] += t; if (pending[i].count) calls[pending[i].stage]++; }
  }
  npending = 0;
}
void StageTimer::reset() { resolve(); memset(ms, 0, sizeof(ms)); memset(calls, 0, sizeof(calls)); }
StageTimer::~StageTimer() { for (int i = 0; i < npool; i++) hipEventDestroy(pool[i]); }

}  // namespace dvs

#ifdef DVS_TEST_HOOKS
// test hook: one wavefront that holds its stream for `ticks` of the 100 MHz wall clock (bounded: every lane leaves at the deadline)
__global__ void k_test_spin(unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
#endif

extern "C" {

const char* dvs_last_error(void) { return dvs::g_err; }

int32_t dvs_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

dvs_status dvs_stream_create(int32_t device, int32_t high_priority, void** out_stream) {
  DVS_ARG(out_stream);
  *out_stream = nullptr;
  DVS_TRY(dvs::check_device(device));
  DVS_HIP(hipSetDevice(device));
  hipStream_t s = nullptr;
  if (high_priority) {   // > 0: highest dispatch priority, < 0: lowest
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    DVS_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, high_priority > 0 ? hi : lo));
  } else {
    DVS_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  }
  *out_stream = (void*)s;
  return DVS_OK;
}
dvs_status dvs_stream_synchronize(void* stream) {
  DVS_HIP(hipStreamSynchronize((hipStream_t)stream));
  return DVS_OK;
}
dvs_status dvs_stream_destroy(void* stream) {
  if (!stream) return DVS_OK;
  DVS_HIP(hipStreamSynchronize((hipStream_t)stream));
  DVS_HIP(hipStreamDestroy((hipStream_t)stream));
  return DVS_OK;
}

dvs_status dvs_event_create(int32_t device, void** out_event) {
  DVS_ARG(out_event);
  *out_event = nullptr;
  DVS_TRY(dvs::check_device(device));
  DVS_HIP(hipSetDevice(device));
  hipEvent_t e = nullptr;
  DVS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  *out_event = e;
  return DVS_OK;
}
dvs_status dvs_event_create_timing(int32_t device, void** out_event) {
  DVS_ARG(out_event);
  *out_event = nullptr;
  DVS_TRY(dvs::check_device(device));
  DVS_HIP(hipSetDevice(device));
  hipEvent_t e = nullptr;
  DVS_HIP(hipEventCreate(&e));
  *out_event = e;
  return DVS_OK;
}
dvs_status dvs_event_elapsed_ms(void* start, void* stop, float* ms) {
  DVS_ARG(start && stop && ms);
  DVS_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return DVS_OK;
}
dvs_status dvs_event_destroy(void* event) {
  if (event) DVS_HIP(hipEventDestroy((hipEvent_t)event));
  return DVS_OK;
}
dvs_status dvs_event_synchronize(void* event) {
  DVS_ARG(event);
  DVS_HIP(hipEventSynchronize((hipEvent_t)event));
  return DVS_OK;
}
dvs_status dvs_event_query(void* event, int32_t* done) {
  DVS_ARG(event && done);
  const hipError_t e = hipEventQuery((hipEvent_t)event);
  if (e != hipSuccess && e != hipErrorNotReady) DVS_HIP(e);
  *done = e == hipSuccess;
  return DVS_OK;
}
dvs_status dvs_event_record(void* event, void* stream) {
  DVS_ARG(event);
  DVS_HIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
  return DVS_OK;
}
dvs_status dvs_stream_wait_event(void* stream, void* event) {
  DVS_ARG(event);
  DVS_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
  return DVS_OK;
}

dvs_status dvs_device_arch(int32_t device, char* buf, int32_t cap) {
  DVS_ARG(buf && cap > 0);
  buf[0] = 0;
  DVS_TRY(dvs::check_device(device));
  hipDeviceProp_t p;
  DVS_HIP(hipGetDeviceProperties(&p, device));
  snprintf(buf, cap, "%s", p.gcnArchName);
  return DVS_OK;
}

dvs_status dvs_malloc(int32_t device, size_t bytes, void** out) {
  DVS_ARG(out);
  DVS_TRY(dvs::check_device(device));
  DVS_HIP(hipMalloc(out, bytes ? bytes : 1));
  return DVS_OK;
}
dvs_status dvs_free(int32_t device, void* p) {
  DVS_TRY(dvs::check_device(device));
  DVS_HIP(hipFree(p));
  return DVS_OK;
}
dvs_status dvs_memcpy_h2d(int32_t device, void* dst, const void* src, size_t bytes) {
  DVS_TRY(dvs::check_device(device));
  DVS_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return DVS_OK;
}
dvs_status dvs_memcpy_d2h(int32_t device, void* dst, const void* src, size_t bytes) {
  DVS_TRY(dvs::check_device(device));
  DVS_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return DVS_OK;
}
dvs_status dvs_memset(int32_t device, void* dst, int value, size_t bytes) {
  DVS_TRY(dvs::check_device(device));
  DVS_HIP(hipMemset(dst, value, bytes));
  DVS_HIP(hipStreamSynchronize(nullptr));  // complete before the caller enqueues on a non-blocking stream
  return DVS_OK;
}
#ifdef DVS_TEST_HOOKS
dvs_status dvs_test_stream_delay(void* stream, int32_t microseconds) {
  DVS_ARG(microseconds >= 0 && microseconds <= 200000);   // bounded: 0.2 s
  hipLaunchKernelGGL(k_test_spin, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long)microseconds * 100ull);
  DVS_HIP(hipGetLastError());
  return DVS_OK;
}
#endif
}
