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

// the buffer registered with qiddm_set_stamp_buffer if it holds at least `need_words` words, else nullptr
unsigned long long* stamp_buffer(int64_t need_words);
int set_stamp_buffer(void* device_ptr, int64_t n_words);

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

// wide CZ forward (qiddm_wide.hip / qsim_wide_cz.h): n = 11..16, CZ entanglers, no or RZ encoding
inline bool wide_cz_eligible(const qiddm_circuit_t* c) {
  return c->n_qubits > QIDDM_MAX_QUBITS_FUSED && c->imprimitive == QIDDM_IMP_CZ &&
         (c->encoding == QIDDM_ENC_NONE || c->encoding == QIDDM_ENC_RZ);
}
int64_t wide_cz_grid(int64_t batch, int64_t slabs);
// pass-structured reverse sweep (qsim_wide_cz_adjoint.h): one round of >= 2 layers of that family.
// QIDDM_WIDE_TILED=1 keeps the generic per-gate kernels (kernel experiments: A/B on the same box)
bool wide_cz_adjoint_eligible(const qiddm_circuit_t* c);
// register-resident reverse sweep of 10-qubit CZ circuits (qsim_cz10_adjoint.h): one round of >= 2 layers
bool cz10_adjoint_eligible(const qiddm_circuit_t* c);
// 10-qubit CZ circuits with no / RZ encoding carry BOTH tails behind the gate variants: the folded tables of the n <= 10
// forward, then the per-wire tables (2n entries per layer) of the reverse sweep
inline bool cz10_tables(const qiddm_circuit_t* c) {
  return c->n_qubits == 10 && c->imprimitive == QIDDM_IMP_CZ &&
         (c->encoding == QIDDM_ENC_NONE || c->encoding == QIDDM_ENC_RZ);
}

}  // namespace qiddm_capi

namespace qiddm { struct KScalars; }
namespace qiddm_capi {
// `tail`: the per-layer tables behind the gate variants of the gate table; `slabs`: 2^n-amplitude slabs in `ws`
int launch_wide_cz(int dtype, int n, const void* inputs, const void* tail, void* out, void* ws,
                   const qiddm::KScalars& p, int64_t slabs, void* stream);
// `partials`: `grid` slabs of `slab_stride` elements ([layer][theta | alpha][16]); `ws`: `grid` pairs of slabs
int launch_wide_cz_adjoint(int dtype, int n, const void* inputs, const void* tail, const void* gout, void* partials,
                           int64_t slab_stride, void* grad_inputs, int64_t gin_ld, void* ws, const qiddm::KScalars& p,
                           int64_t grid, void* stream);
int launch_cz10_adjoint(int dtype, const void* inputs, const void* tail, const void* gout, void* partials,
                        int64_t slab_stride, void* grad_inputs, int64_t gin_ld, const qiddm::KScalars& p, int64_t grid,
                        void* stream);
}  // namespace qiddm_capi
