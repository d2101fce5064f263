// capi_common.h -- what the translation units behind include/qiddm_hip.h share: the thread-local
// error string, descriptor validation, launch limits.
#pragma once
#include "../../include/qiddm_hip.h"

#include <cstddef>

namespace qiddm_capi {

constexpr size_t kMaxLds = 160 * 1024;  // per-workgroup LDS on gfx950

// formats the thread's error message (read back by qiddm_last_error) and returns `code`
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int check_circuit(const qiddm_circuit_t* c);

}  // namespace qiddm_capi
