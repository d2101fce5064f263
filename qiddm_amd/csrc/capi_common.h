// capi_common.h -- what the translation units behind include/qiddm_hip.h share: the thread-local
// error string, descriptor validation, launch limits.
#pragma once
#include "../../include/qiddm_hip.h"

#include <hip/hip_runtime.h>

#include <cstddef>

namespace qiddm_capi {

constexpr size_t kMaxLds = 160 * 1024;  // per-workgroup LDS on gfx950

// formats the thread's error message (read back by qiddm_last_error) and returns `code`
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int check_circuit(const qiddm_circuit_t* c);

// "done once" marker per HIP device: hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device,
// so a process that drives a second GPU has to repeat it there.  Races are benign (the call is idempotent).
struct DeviceFlags {
  static constexpr int kMaxDevices = 64;
  bool done[kMaxDevices] = {};
  static int current() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) return -1;
    return d;
  }
  bool get() const { const int d = current(); return d >= 0 && done[d]; }   // unknown device: always re-apply
  void set() { const int d = current(); if (d >= 0) done[d] = true; }
};

}  // namespace qiddm_capi
