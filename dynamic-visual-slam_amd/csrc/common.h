// common.h — error plumbing shared by the C-ABI translation units of libdvslam_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/dvslam_hip.h"
#include "../../include/dvslam_hip_test.h"   // the extractor's scheduling hooks: used inside the library (pipeline.hip), exported by the test build only

// scheduling / introspection hooks (declared in dvslam_hip_test.h): hidden in lib/libdvslam_hip.so, exported by lib/libdvslam_hip_test.so
#ifdef DVS_TEST_HOOKS
#define DVS_HOOK __attribute__((visibility("default")))
#else
#define DVS_HOOK __attribute__((visibility("hidden")))
#endif

namespace dvs {

void set_error(const char* fmt, ...);  // util.hip
dvs_status check_device(int device);   // selects the device; DVS_ERR_NO_DEVICE if none / not gfx950
dvs_status env_check();                // util.hip: every DVS_* variable of the environment is a known switch with an allowed value
int env_switch(const char* name, int dflt);   // value of a switch of util.hip's table (validated by env_check at handle creation)

#define DVS_HIP(call)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      dvs::set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      return DVS_ERR_HIP;                                                                      \
    }                                                                                          \
  } while (0)

#define DVS_TRY(call)                \
  do {                               \
    dvs_status s_ = (call);          \
    if (s_ != DVS_OK) return s_;     \
  } while (0)

#define DVS_ARG(cond)                                                            \
  do {                                                                           \
    if (!(cond)) {                                                               \
      dvs::set_error("%s:%d: bad argument: %s", __FILE__, __LINE__, #cond);      \
      return DVS_ERR_ARG;                                                        \
    }                                                                            \
  } while (0)

// stage timer: pairs of hipEvents recorded on the handle's stream, resolved lazily
struct StageTimer {
  static const int kMaxPending = 4096;
  struct Pending { hipEvent_t a, b; int stage; bool count; };
  Pending pending[kMaxPending];
  int npending = 0;
  double ms[16] = {0};
  int64_t calls[16] = {0};
  bool on = false;
  void begin(int stage, hipStream_t s, bool count_call = true);
  void end(hipStream_t s);
  void resolve();
  void reset();
  ~StageTimer();
 private:
  hipEvent_t pool[2 * kMaxPending];
  int npool = 0, cur = -1;
};

}  // namespace dvs
