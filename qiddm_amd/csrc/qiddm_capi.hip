// qiddm_capi.hip -- extern "C" entry points declared in include/qiddm_hip.h.
// Argument validation, launch geometry and template dispatch only; the device
// code lives in qsim_fused.h.
#include "capi_common.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "qsim_fused.h"
#include "qsim_tiled.h"
#include "qsim_adjoint.h"
#include "qsim_adjoint_wide.h"
#include "qsim_quad.h"

namespace qiddm_capi {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

static std::atomic<unsigned long long*> g_stamp_ptr{nullptr};
static std::atomic<int64_t> g_stamp_words{0};

unsigned long long* stamp_buffer(int64_t need_words) {
  unsigned long long* p = g_stamp_ptr.load(std::memory_order_acquire);
  return (p != nullptr && g_stamp_words.load(std::memory_order_acquire) >= need_words) ? p : nullptr;
}

int set_stamp_buffer(void* device_ptr, int64_t n_words) {
  if (device_ptr == nullptr || n_words == 0) {
    g_stamp_words.store(0, std::memory_order_release);
    g_stamp_ptr.store(nullptr, std::memory_order_release);
    return QIDDM_OK;
  }
  if (n_words < 8) return fail(QIDDM_ERR_INVALID, "stamp buffer needs at least 8 words (got %lld)", (long long)n_words);
  if ((reinterpret_cast<uintptr_t>(device_ptr) & 7u) != 0) return fail(QIDDM_ERR_INVALID, "stamp buffer must be 8-byte aligned");
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, device_ptr) != hipSuccess) {
    (void)hipGetLastError();
    return fail(QIDDM_ERR_INVALID, "stamp buffer %p is not memory the HIP runtime knows", device_ptr);
  }
  if (attr.type != hipMemoryTypeDevice)
    return fail(QIDDM_ERR_INVALID, "stamp buffer %p is not device memory (type %d)", device_ptr, (int)attr.type);
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  if (hipMemGetAddressRange(&base, &size, device_ptr) != hipSuccess) {
    (void)hipGetLastError();
    return fail(QIDDM_ERR_INVALID, "cannot query the allocation of stamp buffer %p", device_ptr);
  }
  const char* end = static_cast<const char*>(device_ptr) + (size_t)n_words * 8;
  if (end > static_cast<const char*>(base) + size)
    return fail(QIDDM_ERR_INVALID, "%lld words at %p run past the end of their allocation", (long long)n_words, device_ptr);
  g_stamp_ptr.store(static_cast<unsigned long long*>(device_ptr), std::memory_order_release);
  g_stamp_words.store(n_words, std::memory_order_release);
  return QIDDM_OK;
}

int check_circuit(const qiddm_circuit_t* c) {
  if (!c) return fail(QIDDM_ERR_INVALID, "circ is NULL");
  if (c->n_qubits < 1) return fail(QIDDM_ERR_INVALID, "n_qubits=%d must be >= 1", c->n_qubits);
  if (c->n_qubits > QIDDM_MAX_QUBITS)
    return fail(QIDDM_ERR_UNSUPPORTED, "n_qubits=%d exceeds the limit %d", c->n_qubits, QIDDM_MAX_QUBITS);
  if (c->encoding < QIDDM_ENC_NONE || c->encoding > QIDDM_ENC_RY_BLOCKS)
    return fail(QIDDM_ERR_INVALID, "unknown encoding %d", c->encoding);
  if (c->imprimitive != QIDDM_IMP_CNOT && c->imprimitive != QIDDM_IMP_CZ)
    return fail(QIDDM_ERR_INVALID, "unknown imprimitive %d", c->imprimitive);
  if (c->measure != QIDDM_MEAS_PROBS && c->measure != QIDDM_MEAS_EXPZ)
    return fail(QIDDM_ERR_INVALID, "unknown measure %d", c->measure);
  if (c->dtype != QIDDM_F32 && c->dtype != QIDDM_F64)
    return fail(QIDDM_ERR_INVALID, "unknown dtype %d", c->dtype);
  if (c->n_rounds < 1 || c->n_blocks < 1 || c->sel_layers < 1)
    return fail(QIDDM_ERR_INVALID, "n_rounds/n_blocks/sel_layers must be >= 1 (got %d/%d/%d)",
                c->n_rounds, c->n_blocks, c->sel_layers);
  if (c->reserved != 0) return fail(QIDDM_ERR_INVALID, "reserved field must be 0");
  const int64_t d = (int64_t)1 << c->n_qubits;
  if (c->encoding == QIDDM_ENC_AMPLITUDE) {
    // PennyLane: "Features must be of length 2^n or smaller"
    if (c->n_features < 1 || c->n_features > d)
      return fail(QIDDM_ERR_INVALID, "Features must be of length %lld or smaller; got length %d.",
                  (long long)d, c->n_features);
    if (c->n_rounds != 1)
      return fail(QIDDM_ERR_UNSUPPORTED, "amplitude encoding supports n_rounds == 1 only");
  } else if (c->encoding >= QIDDM_ENC_RZ) {
    if (c->n_features < c->n_qubits)
      return fail(QIDDM_ERR_INVALID, "angle encoding needs n_features >= n_qubits (%d < %d)",
                  c->n_features, c->n_qubits);
  }
  if ((int64_t)c->n_rounds * c->n_blocks * c->sel_layers * c->n_qubits > (1 << 24))
    return fail(QIDDM_ERR_UNSUPPORTED, "too many Rot gates");
  return QIDDM_OK;
}

}  // namespace qiddm_capi

namespace {

using qiddm_capi::check_circuit;
using qiddm_capi::fail;
using qiddm_capi::g_err;
using qiddm_capi::kMaxLds;

struct Ptrs {
  const void* inputs = nullptr;
  const void* table = nullptr;
  void* out = nullptr;
  const void* gout = nullptr;
  void* dots = nullptr;
};

template <typename T, int N, bool SHIFT>
int launch(const Ptrs& ptr, const qiddm::KScalars& p, int64_t n_replicas, hipStream_t stream) {
  using L = qiddm::Layout<N>;
  using S = qiddm::Smem<T, N>;
  const int64_t groups = (p.batch + L::SPW - 1) / L::SPW;
  if (groups == 0 || (SHIFT && n_replicas == 0)) return QIDDM_OK;
  const int waves = 4;
  int64_t bx = (groups + waves - 1) / waves;
  // enough blocks to fill 256 CUs several times over, then grid-stride
  const int64_t cap = SHIFT ? 1024 : 4096;
  if (bx > cap) bx = cap;
  dim3 grid((unsigned)bx, SHIFT ? (unsigned)n_replicas : 1u, 1u);
  const int64_t n_rot = (int64_t)p.n_rounds * p.n_blocks * p.sel_layers * N;
  const size_t smem = S::bytes(n_rot, p.imprimitive == QIDDM_IMP_CNOT, waves);
  if (smem > kMaxLds)
    return fail(QIDDM_ERR_UNSUPPORTED,
                "circuit with %lld Rot gates needs %zu B of LDS for its gate table (limit %zu)",
                (long long)n_rot, smem, kMaxLds);
  if constexpr (!SHIFT && N >= 8) {
    // forward of a CZ circuit with no / RZ encoding at the wide register-resident sizes: the lean folded-only kernel
    // (100 / 134 / 206 VGPRs at n = 8 / 9 / 10 -> 4 / 3 / 2 waves per SIMD; the all-paths kernel: 235 / 255+34 / 256+205)
    static const bool all_paths = std::getenv("QIDDM_NO_LEAN") != nullptr;   // kernel experiments: A/B
    // ... when there is more than one wave of work per SIMD: below that the all-paths kernel's one-layer-ahead table
    // prefetch wins (n = 10, 256 samples: 69.8 vs 76.5 us; 4096: 274 vs 245; 65 536: 4.18 vs 3.68 ms, gpurun_out/r02g)
    // Round 3: the lean kernel runs its layers in tangent form (Engine::tangent_fold; n = 8, 65 536 samples, G = 232 / 480:
    // 0.40 / 0.66 -> 0.30 / 0.51 ms; n = 10, G = 900, 4096 samples: 245 -> 189 us).  At one wave per SIMD the all-paths
    // kernel still wins at n = 8, 9 (B = 1024: 12.0 vs 14.7, 27.7 vs 31.8 us); at n = 10 the lean kernel wins at every
    // batch (B = 1024: 64.9 vs 69.5 us; tools/ab_lean_threshold.py, gpurun_out/ab_lean_*.log).  QIDDM_LEAN_ABOVE: A/B.
    static const int64_t lean_above_env = std::getenv("QIDDM_LEAN_ABOVE") ? std::atoll(std::getenv("QIDDM_LEAN_ABOVE")) : -1;
    const int64_t lean_above = lean_above_env >= 0 ? lean_above_env : (N == 10 ? 0 : 1024);
    if (p.fold && p.encoding != QIDDM_ENC_AMPLITUDE && !all_paths && groups > lean_above) {
      auto kf = qiddm::circuit_folded_kernel<T, N>;
      static qiddm_capi::DeviceFlags big_lds_folded;
      if (smem > 48 * 1024 && !big_lds_folded.get()) {
        const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(kf),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
        if (ea != hipSuccess)
          return fail(QIDDM_ERR_LAUNCH, "hipFuncSetAttribute(LDS) failed: %s", hipGetErrorString(ea));
        big_lds_folded.set();
      }
      hipLaunchKernelGGL(kf, grid, dim3(waves * qiddm::kWave), smem, stream, static_cast<const T*>(ptr.inputs),
                         static_cast<const T*>(ptr.table), static_cast<T*>(ptr.out), p);
      const hipError_t ef = hipGetLastError();
      if (ef != hipSuccess)
        return fail(QIDDM_ERR_LAUNCH, "circuit_folded_kernel<n=%d> launch failed: %s", N, hipGetErrorString(ef));
      return QIDDM_OK;
    }
  }
  auto kern = qiddm::circuit_kernel<T, N, SHIFT>;
  static qiddm_capi::DeviceFlags big_lds_enabled;  // benign race: the attribute call is idempotent
  if (smem > 48 * 1024 && !big_lds_enabled.get()) {
    const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    if (ea != hipSuccess)
      return fail(QIDDM_ERR_LAUNCH, "hipFuncSetAttribute(LDS) failed: %s", hipGetErrorString(ea));
    big_lds_enabled.set();
  }
  hipLaunchKernelGGL(kern, grid, dim3(waves * qiddm::kWave), smem, stream, static_cast<const T*>(ptr.inputs),
                     static_cast<const T*>(ptr.table), static_cast<T*>(ptr.out),
                     static_cast<const T*>(ptr.gout), static_cast<T*>(ptr.dots), p);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess)
    return fail(QIDDM_ERR_LAUNCH, "circuit_kernel<n=%d> launch failed: %s", N, hipGetErrorString(e));
  return QIDDM_OK;
}

template <typename T, bool SHIFT>
int dispatch_n(int n, const Ptrs& ptr, const qiddm::KScalars& p, int64_t n_replicas, hipStream_t stream) {
  switch (n) {
    case 1: return launch<T, 1, SHIFT>(ptr, p, n_replicas, stream);
    case 2: return launch<T, 2, SHIFT>(ptr, p, n_replicas, stream);
    case 3: return launch<T, 3, SHIFT>(ptr, p, n_replicas, stream);
    case 4: return launch<T, 4, SHIFT>(ptr, p, n_replicas, stream);
    case 5: return launch<T, 5, SHIFT>(ptr, p, n_replicas, stream);
    case 6: return launch<T, 6, SHIFT>(ptr, p, n_replicas, stream);
    case 7: return launch<T, 7, SHIFT>(ptr, p, n_replicas, stream);
    case 8: return launch<T, 8, SHIFT>(ptr, p, n_replicas, stream);
    case 9: return launch<T, 9, SHIFT>(ptr, p, n_replicas, stream);
    case 10: return launch<T, 10, SHIFT>(ptr, p, n_replicas, stream);
    default: return fail(QIDDM_ERR_UNSUPPORTED, "n_qubits=%d not instantiated", n);
  }
}

template <typename T, int N, bool LDSW, int WPB>
int launch_dense_impl(const double* x, const double* wd, const double* bd, const double* angles,
                      const double* wu, const double* bu, double* y, const qiddm::DenseScalars& d,
                      const qiddm::KScalars& p, size_t smem, unsigned blocks, hipStream_t stream) {
  constexpr int waves = WPB;
  auto kern = qiddm::dense_forward_kernel<T, N, LDSW, WPB>;
  static qiddm_capi::DeviceFlags big_lds_enabled;
  if (smem > 48 * 1024 && !big_lds_enabled.get()) {
    const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    if (ea != hipSuccess)
      return fail(QIDDM_ERR_LAUNCH, "hipFuncSetAttribute(LDS) failed: %s", hipGetErrorString(ea));
    big_lds_enabled.set();
  }
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * qiddm::kWave), smem, stream, x, wd, bd, angles, wu,
                     bu, y, d, p);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess)
    return fail(QIDDM_ERR_LAUNCH, "dense_forward_kernel<n=%d> launch failed: %s", N, hipGetErrorString(e));
  return QIDDM_OK;
}

template <typename T, int N>
int launch_dense(const double* x, const double* wd, const double* bd, const double* angles,
                 const double* wu, const double* bu, double* y, const qiddm::DenseScalars& d,
                 const qiddm::KScalars& p, hipStream_t stream) {
  using L = qiddm::Layout<N>;
  using S = qiddm::Smem<T, N>;
  const int64_t groups = (p.batch + L::SPW - 1) / L::SPW;
  if (groups == 0) return QIDDM_OK;
  // 8-wave workgroups (two waves per SIMD share one LDS copy of the weights) once the chip is
  // full and the register budget allows it; 4-wave workgroups otherwise
  constexpr bool kCanUse8 = sizeof(T) == 4 && N <= 8;
  // small batches: one wave per workgroup so every sample gets a CU (and its L1) of its own
  const int waves = groups <= 1024 ? 1 : ((kCanUse8 && groups > 2048) ? 8 : 4);
  int64_t bx = (groups + waves - 1) / waves;
  if (bx > 4096) bx = 4096;
  const int64_t n_rot = (int64_t)p.n_rounds * p.n_blocks * p.sel_layers * N;
  const size_t base = S::bytes(n_rot, p.imprimitive == QIDDM_IMP_CNOT, waves);
  if (base > kMaxLds)
    return fail(QIDDM_ERR_UNSUPPORTED, "circuit with %lld Rot gates needs %zu B of LDS (limit %zu)",
                (long long)n_rot, base, kMaxLds);
  // both weight matrices in LDS when they fit next to the gate tables
  const size_t wbytes = (size_t)N * ((size_t)d.in_features + (size_t)d.out_features) * sizeof(double);
  static const bool no_ldsw = std::getenv("QIDDM_DENSE_NO_LDSW") != nullptr;  // tuning switch
  const bool ldsw = !no_ldsw && waves > 1 && base + wbytes <= kMaxLds;
  const size_t smem = ldsw ? base + wbytes : base;
  const unsigned blocks = (unsigned)bx;
  if constexpr (kCanUse8) {
    if (waves == 8)
      return ldsw ? launch_dense_impl<T, N, true, 8>(x, wd, bd, angles, wu, bu, y, d, p, smem, blocks, stream)
                  : launch_dense_impl<T, N, false, 8>(x, wd, bd, angles, wu, bu, y, d, p, smem, blocks, stream);
  }
  if (waves == 1)
    return launch_dense_impl<T, N, false, 1>(x, wd, bd, angles, wu, bu, y, d, p, smem, blocks, stream);
  return ldsw ? launch_dense_impl<T, N, true, 4>(x, wd, bd, angles, wu, bu, y, d, p, smem, blocks, stream)
              : launch_dense_impl<T, N, false, 4>(x, wd, bd, angles, wu, bu, y, d, p, smem, blocks, stream);
}

template <typename T>
int dispatch_dense(int n, const double* x, const double* wd, const double* bd, const double* angles,
                   const double* wu, const double* bu, double* y, const qiddm::DenseScalars& d,
                   const qiddm::KScalars& p, hipStream_t st) {
  switch (n) {
    case 1: return launch_dense<T, 1>(x, wd, bd, angles, wu, bu, y, d, p, st);
    case 2: return launch_dense<T, 2>(x, wd, bd, angles, wu, bu, y, d, p, st);
    case 3: return launch_dense<T, 3>(x, wd, bd, angles, wu, bu, y, d, p, st);
    case 4: return launch_dense<T, 4>(x, wd, bd, angles, wu, bu, y, d, p, st);
    case 5: return launch_dense<T, 5>(x, wd, bd, angles, wu, bu, y, d, p, st);
    case 6: return launch_dense<T, 6>(x, wd, bd, angles, wu, bu, y, d, p, st);
    case 7: return launch_dense<T, 7>(x, wd, bd, angles, wu, bu, y, d, p, st);
    case 8: return launch_dense<T, 8>(x, wd, bd, angles, wu, bu, y, d, p, st);
    case 9: return launch_dense<T, 9>(x, wd, bd, angles, wu, bu, y, d, p, st);
    case 10: return launch_dense<T, 10>(x, wd, bd, angles, wu, bu, y, d, p, st);
    default: return fail(QIDDM_ERR_UNSUPPORTED, "n_qubits=%d not instantiated", n);
  }
}

template <typename T, int N>
int launch_qconv(const double* x, const double* angles, double* y, const qiddm::ConvScalars& cv,
                 const qiddm::KScalars& p, hipStream_t stream) {
  using L = qiddm::Layout<N>;
  using S = qiddm::Smem<T, N>;
  const int64_t groups = (p.batch + L::SPW - 1) / L::SPW;
  if (groups == 0) return QIDDM_OK;
  const int waves = 4;
  int64_t bx = (groups + waves - 1) / waves;
  if (bx > 4096) bx = 4096;
  const int64_t n_rot = (int64_t)p.sel_layers * N;
  const size_t smem = S::bytes(n_rot, p.imprimitive == QIDDM_IMP_CNOT, waves);
  if (smem > kMaxLds)
    return fail(QIDDM_ERR_UNSUPPORTED, "circuit with %lld Rot gates needs %zu B of LDS (limit %zu)",
                (long long)n_rot, smem, kMaxLds);
  auto kern = qiddm::qconv_forward_kernel<T, N>;
  static qiddm_capi::DeviceFlags big_lds_enabled;
  if (smem > 48 * 1024 && !big_lds_enabled.get()) {
    const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    if (ea != hipSuccess)
      return fail(QIDDM_ERR_LAUNCH, "hipFuncSetAttribute(LDS) failed: %s", hipGetErrorString(ea));
    big_lds_enabled.set();
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)bx), dim3(waves * qiddm::kWave), smem, stream, x, angles, y, cv, p);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess)
    return fail(QIDDM_ERR_LAUNCH, "qconv_forward_kernel<n=%d> launch failed: %s", N, hipGetErrorString(e));
  return QIDDM_OK;
}

template <typename T>
int dispatch_qconv(int n, const double* x, const double* angles, double* y, const qiddm::ConvScalars& cv,
                   const qiddm::KScalars& p, hipStream_t st) {
  switch (n) {
    case 1: return launch_qconv<T, 1>(x, angles, y, cv, p, st);
    case 2: return launch_qconv<T, 2>(x, angles, y, cv, p, st);
    case 3: return launch_qconv<T, 3>(x, angles, y, cv, p, st);
    case 4: return launch_qconv<T, 4>(x, angles, y, cv, p, st);
    case 5: return launch_qconv<T, 5>(x, angles, y, cv, p, st);
    case 6: return launch_qconv<T, 6>(x, angles, y, cv, p, st);
    case 7: return launch_qconv<T, 7>(x, angles, y, cv, p, st);
    case 8: return launch_qconv<T, 8>(x, angles, y, cv, p, st);
    case 9: return launch_qconv<T, 9>(x, angles, y, cv, p, st);
    case 10: return launch_qconv<T, 10>(x, angles, y, cv, p, st);
    default: return fail(QIDDM_ERR_UNSUPPORTED, "fused QConv2d needs n_qubits <= 10 (got %d)", n);
  }
}

// ---- quad layout (4 waves per sample) dense sampler ---------------------------------------------------
template <typename T>
size_t quad_lds_bytes(int n, int64_t n_rot) {
  switch (n) {
    case 2: return qiddm::QuadSmem<T, 2>::bytes(n_rot);
    case 3: return qiddm::QuadSmem<T, 3>::bytes(n_rot);
    case 4: return qiddm::QuadSmem<T, 4>::bytes(n_rot);
    case 5: return qiddm::QuadSmem<T, 5>::bytes(n_rot);
    case 6: return qiddm::QuadSmem<T, 6>::bytes(n_rot);
    case 7: return qiddm::QuadSmem<T, 7>::bytes(n_rot);
    case 8: return qiddm::QuadSmem<T, 8>::bytes(n_rot);
    case 9: return qiddm::QuadSmem<T, 9>::bytes(n_rot);
    default: return qiddm::QuadSmem<T, 10>::bytes(n_rot);
  }
}

bool quad_supported(const qiddm_circuit_t* c, int64_t in_features, int64_t out_features) {
  if (!(c->n_qubits >= 2 && c->n_qubits <= 10 && c->imprimitive == QIDDM_IMP_CZ &&
        c->encoding == QIDDM_ENC_RZ && c->measure == QIDDM_MEAS_EXPZ && in_features <= 2048 && out_features <= 2048))
    return false;
  // its per-layer phase tables must fit in LDS (deep float64 circuits at n = 10 do not)
  const int64_t n_rot = (int64_t)c->n_rounds * c->n_blocks * c->sel_layers * c->n_qubits;
  const size_t lds = c->dtype == QIDDM_F32 ? quad_lds_bytes<float>(c->n_qubits, n_rot)
                                           : quad_lds_bytes<double>(c->n_qubits, n_rot);
  return lds <= kMaxLds;
}

template <typename T, int N>
int launch_quad(const double* x, const double* wd, const double* bd, const double* angles, const double* wu,
                const double* bu, double* y, const void* tables, const qiddm::QuadScalars& d,
                const qiddm::KScalars& p, hipStream_t stream) {
  if (p.batch == 0 || d.n_steps == 0) return QIDDM_OK;
  const int64_t n_rot = (int64_t)p.n_rounds * p.n_blocks * p.sel_layers * N;
  const size_t smem = qiddm::QuadSmem<T, N>::bytes(n_rot);
  if (smem > kMaxLds)
    return fail(QIDDM_ERR_UNSUPPORTED, "circuit with %lld Rot gates needs %zu B of LDS (limit %zu)",
                (long long)n_rot, smem, kMaxLds);
  const bool small = d.in_features <= 1024 && d.out_features <= 1024;
  auto kern = small ? qiddm::dense_quad_kernel<T, N, 4> : qiddm::dense_quad_kernel<T, N, 8>;
  static qiddm_capi::DeviceFlags big_lds_enabled[2];
  if (smem > 48 * 1024 && !big_lds_enabled[small].get()) {
    const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    if (ea != hipSuccess)
      return fail(QIDDM_ERR_LAUNCH, "hipFuncSetAttribute(LDS) failed: %s", hipGetErrorString(ea));
    big_lds_enabled[small].set();
  }
  const unsigned blocks = (unsigned)(p.batch < 2048 ? p.batch : 2048);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), smem, stream, x, wd, bd, angles, wu, bu, y,
                     static_cast<const T*>(tables), d, p);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess)
    return fail(QIDDM_ERR_LAUNCH, "dense_quad_kernel<n=%d> launch failed: %s", N, hipGetErrorString(e));
  return QIDDM_OK;
}

template <typename T, int N>
size_t quad_tables_bytes_n(int64_t n_rot) {
  using QS = qiddm::QuadSmem<T, N>;
  return (QS::ry_bytes(n_rot) + 15) / 16 * 16 + QS::tlo_bytes(n_rot) + QS::thi_bytes(n_rot);
}
template <typename T>
size_t quad_tables_bytes(int n, int64_t n_rot) {
  switch (n) {
    case 2: return quad_tables_bytes_n<T, 2>(n_rot);
    case 3: return quad_tables_bytes_n<T, 3>(n_rot);
    case 4: return quad_tables_bytes_n<T, 4>(n_rot);
    case 5: return quad_tables_bytes_n<T, 5>(n_rot);
    case 6: return quad_tables_bytes_n<T, 6>(n_rot);
    case 7: return quad_tables_bytes_n<T, 7>(n_rot);
    case 8: return quad_tables_bytes_n<T, 8>(n_rot);
    case 9: return quad_tables_bytes_n<T, 9>(n_rot);
    default: return quad_tables_bytes_n<T, 10>(n_rot);
  }
}
template <typename T, int N>
int launch_quad_tables(const double* angles, void* tables, const qiddm::KScalars& p, int64_t n_rot, hipStream_t st) {
  const size_t smem = (size_t)n_rot * sizeof(double);
  if (smem > 48 * 1024) return fail(QIDDM_ERR_UNSUPPORTED, "too many Rot gates (%lld) for the table builder", (long long)n_rot);
  hipLaunchKernelGGL((qiddm::quad_tables_kernel<T, N>), dim3(1), dim3(256), smem, st, angles, static_cast<T*>(tables), p);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "quad_tables_kernel launch failed: %s", hipGetErrorString(e));
  return QIDDM_OK;
}

template <typename T>
int dispatch_quad(int n, const double* x, const double* wd, const double* bd, const double* angles,
                  const double* wu, const double* bu, double* y, const void* tables, const qiddm::QuadScalars& d,
                  const qiddm::KScalars& p, hipStream_t st) {
  switch (n) {
    case 2: return launch_quad<T, 2>(x, wd, bd, angles, wu, bu, y, tables, d, p, st);
    case 3: return launch_quad<T, 3>(x, wd, bd, angles, wu, bu, y, tables, d, p, st);
    case 4: return launch_quad<T, 4>(x, wd, bd, angles, wu, bu, y, tables, d, p, st);
    case 5: return launch_quad<T, 5>(x, wd, bd, angles, wu, bu, y, tables, d, p, st);
    case 6: return launch_quad<T, 6>(x, wd, bd, angles, wu, bu, y, tables, d, p, st);
    case 7: return launch_quad<T, 7>(x, wd, bd, angles, wu, bu, y, tables, d, p, st);
    case 8: return launch_quad<T, 8>(x, wd, bd, angles, wu, bu, y, tables, d, p, st);
    case 9: return launch_quad<T, 9>(x, wd, bd, angles, wu, bu, y, tables, d, p, st);
    case 10: return launch_quad<T, 10>(x, wd, bd, angles, wu, bu, y, tables, d, p, st);
    default: return fail(QIDDM_ERR_UNSUPPORTED, "quad layout needs 8 <= n <= 10 (got %d)", n);
  }
}

// ---- adjoint backward (n <= 10) -------------------------------------------------------------------------
template <int N>
int64_t adjoint_blocks(int64_t batch) {
  using L = qiddm::Layout<N>;
  const int64_t groups = (batch + L::SPW - 1) / L::SPW;
  int64_t bx = (groups + 3) / 4;
  if (bx > 512) bx = 512;
  return bx < 1 ? 1 : bx;
}
int64_t adjoint_blocks_n(int n, int64_t batch) {
  switch (n) {
    case 1: return adjoint_blocks<1>(batch);
    case 2: return adjoint_blocks<2>(batch);
    case 3: return adjoint_blocks<3>(batch);
    case 4: return adjoint_blocks<4>(batch);
    case 5: return adjoint_blocks<5>(batch);
    default: return adjoint_blocks<6>(batch);  // SPW == 1 from n = 6 on
  }
}

struct ConvPtrs {  // quantum-convolution backward: image, dL/dy and geometry (unused otherwise)
  const double* img = nullptr;
  const double* gy = nullptr;
  qiddm::ConvScalars cv{};
};

template <typename T, int N, bool CONV>
int launch_adjoint(const Ptrs& ptr, T* k_partials, T* grad_inputs, const qiddm::KScalars& p,
                   const qiddm::AdjointScalars& ad, const ConvPtrs& conv, hipStream_t stream) {
  using S = qiddm::Smem<T, N>;
  const int waves = 4;
  const int64_t n_rot = (int64_t)p.n_blocks * p.sel_layers * N;
  const size_t smem = S::bytes(n_rot, p.imprimitive == QIDDM_IMP_CNOT, waves) +
                      (size_t)n_rot * qiddm::kLdsGateReals * sizeof(T) + (size_t)waves * n_rot * 8 * sizeof(T);
  if (smem > kMaxLds)
    return fail(QIDDM_ERR_UNSUPPORTED, "circuit with %lld Rot gates needs %zu B of LDS for the adjoint pass",
                (long long)n_rot, smem);
  auto kern = qiddm::adjoint_kernel<T, N, CONV>;
  static qiddm_capi::DeviceFlags big_lds_enabled;
  if (smem > 48 * 1024 && !big_lds_enabled.get()) {
    const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    if (ea != hipSuccess)
      return fail(QIDDM_ERR_LAUNCH, "hipFuncSetAttribute(LDS) failed: %s", hipGetErrorString(ea));
    big_lds_enabled.set();
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)adjoint_blocks<N>(p.batch)), dim3(waves * qiddm::kWave), smem, stream,
                     static_cast<const T*>(ptr.inputs), static_cast<const T*>(ptr.table),
                     static_cast<const T*>(ptr.gout), k_partials, grad_inputs, p, ad, conv.img, conv.gy, conv.cv);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess)
    return fail(QIDDM_ERR_LAUNCH, "adjoint_kernel<n=%d> launch failed: %s", N, hipGetErrorString(e));
  return QIDDM_OK;
}

template <typename T, bool CONV>
int dispatch_adjoint(int n, const Ptrs& ptr, void* kp, void* gi, const qiddm::KScalars& p,
                     const qiddm::AdjointScalars& ad, const ConvPtrs& conv, hipStream_t st) {
  T* k = static_cast<T*>(kp);
  T* g = static_cast<T*>(gi);
  switch (n) {
    case 1: return launch_adjoint<T, 1, CONV>(ptr, k, g, p, ad, conv, st);
    case 2: return launch_adjoint<T, 2, CONV>(ptr, k, g, p, ad, conv, st);
    case 3: return launch_adjoint<T, 3, CONV>(ptr, k, g, p, ad, conv, st);
    case 4: return launch_adjoint<T, 4, CONV>(ptr, k, g, p, ad, conv, st);
    case 5: return launch_adjoint<T, 5, CONV>(ptr, k, g, p, ad, conv, st);
    case 6: return launch_adjoint<T, 6, CONV>(ptr, k, g, p, ad, conv, st);
    case 7: return launch_adjoint<T, 7, CONV>(ptr, k, g, p, ad, conv, st);
    case 8: return launch_adjoint<T, 8, CONV>(ptr, k, g, p, ad, conv, st);
    case 9: return launch_adjoint<T, 9, CONV>(ptr, k, g, p, ad, conv, st);
    case 10: return launch_adjoint<T, 10, CONV>(ptr, k, g, p, ad, conv, st);
    default: return fail(QIDDM_ERR_UNSUPPORTED, "adjoint backward needs n_qubits <= 10 (got %d)", n);
  }
}

// ---- n = 11..16: tiled kernel -----------------------------------------------------------------------
int64_t tiled_blocks_x(int64_t batch, int64_t n_replicas) {
  if (n_replicas <= 0) return batch < 512 ? batch : 512;
  int64_t bx = 512 / n_replicas;
  if (bx < 1) bx = 1;
  if (bx > 64) bx = 64;
  return batch < bx ? batch : bx;
}

int64_t tiled_workspace_bytes(const qiddm_circuit_t* c, int64_t batch, int64_t n_replicas) {
  const int64_t slabs = tiled_blocks_x(batch, n_replicas) * (n_replicas > 0 ? n_replicas : 1);
  const int64_t csize = c->dtype == QIDDM_F32 ? 8 : 16;
  return 2 * slabs * ((int64_t)1 << c->n_qubits) * csize;  // ping-pong pair per workgroup
}

template <typename T, bool SHIFT>
int launch_tiled(int n, const Ptrs& ptr, const qiddm::KScalars& p, int64_t n_replicas, void* ws,
                 hipStream_t stream) {
  if (p.batch == 0 || (SHIFT && n_replicas == 0)) return QIDDM_OK;
  const int64_t n_rot = (int64_t)p.n_rounds * p.n_blocks * p.sel_layers * n;
  const int64_t layers = (int64_t)p.n_blocks * p.sel_layers;
  if (n_rot + 2 * layers + p.n_blocks + 2 * n > qiddm::kTiledMaxOps ||
      2 * layers + 4 > qiddm::kTiledMaxPasses)
    return fail(QIDDM_ERR_UNSUPPORTED, "circuit too deep for the tiled pass program (%lld layers)",
                (long long)layers);
  const size_t smem = qiddm::TiledSmem<T>::bytes(n_rot);
  if (smem > kMaxLds)
    return fail(QIDDM_ERR_UNSUPPORTED, "circuit with %lld Rot gates needs %zu B of LDS (limit %zu)",
                (long long)n_rot, smem, kMaxLds);
  auto kern = qiddm::tiled_circuit_kernel<T, SHIFT>;
  static qiddm_capi::DeviceFlags big_lds_enabled;
  if (smem > 48 * 1024 && !big_lds_enabled.get()) {
    const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    if (ea != hipSuccess)
      return fail(QIDDM_ERR_LAUNCH, "hipFuncSetAttribute(LDS) failed: %s", hipGetErrorString(ea));
    big_lds_enabled.set();
  }
  dim3 grid((unsigned)tiled_blocks_x(p.batch, SHIFT ? n_replicas : 0), SHIFT ? (unsigned)n_replicas : 1u, 1u);
  qiddm::TiledScalars tp;
  tp.n = n;
  tp.pad_ = 0;
  hipLaunchKernelGGL(kern, grid, dim3(qiddm::kTiledWaves * qiddm::kWave), smem, stream,
                     static_cast<const T*>(ptr.inputs), static_cast<const T*>(ptr.table),
                     static_cast<T*>(ptr.out), static_cast<const T*>(ptr.gout), static_cast<T*>(ptr.dots),
                     static_cast<qiddm::V2<T>*>(ws), p, tp);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess)
    return fail(QIDDM_ERR_LAUNCH, "tiled_circuit_kernel<n=%d> launch failed: %s", n, hipGetErrorString(e));
  return QIDDM_OK;
}

int check_workspace(const qiddm_circuit_t* c, int64_t batch, int64_t n_replicas, const void* ws,
                    int64_t ws_bytes) {
  if (c->n_qubits <= QIDDM_MAX_QUBITS_FUSED) return QIDDM_OK;
  const int64_t need = tiled_workspace_bytes(c, batch, n_replicas);
  if (!ws || ws_bytes < need)
    return fail(QIDDM_ERR_INVALID, "n_qubits=%d needs a %lld-byte workspace (got %lld)", c->n_qubits,
                (long long)need, (long long)ws_bytes);
  return QIDDM_OK;
}

qiddm::KScalars make_params(const qiddm_circuit_t* c) {
  qiddm::KScalars p;
  std::memset(&p, 0, sizeof(p));
  p.encoding = c->encoding;
  p.imprimitive = c->imprimitive;
  p.measure = c->measure;
  p.n_rounds = c->n_rounds;
  p.n_blocks = c->n_blocks;
  p.sel_layers = c->sel_layers;
  p.n_features = c->n_features;
  p.enc_scale = c->enc_scale;
  p.enc_offset = c->enc_offset;
  p.pad_with = c->pad_with;
  // QIDDM_NO_FOLD=1: keep the general gate-by-gate forward (kernel experiments)
  static const bool no_fold = std::getenv("QIDDM_NO_FOLD") != nullptr;
  p.fold = (!no_fold && qiddm::can_fold(c->imprimitive, c->encoding)) ? 1 : 0;
  return p;
}

int64_t out_cols(const qiddm_circuit_t* c) {
  return c->measure == QIDDM_MEAS_PROBS ? ((int64_t)1 << c->n_qubits) : c->n_qubits;
}

int64_t wide_adjoint_blocks(int64_t batch) { return batch < 1 ? 1 : (batch < 512 ? batch : 512); }

template <typename T>
int launch_wide_adjoint(const qiddm_circuit_t* c, const void* inputs, const void* table, const void* gout,
                               void* k_partials, void* grad_inputs, void* ws, const qiddm::KScalars& p,
                               int64_t gin_ld, hipStream_t st, bool raw = false) {
  const int64_t n_rot = (int64_t)c->n_blocks * c->sel_layers * c->n_qubits;
  const size_t smem = ((size_t)n_rot * 8 + 64 + 48) * sizeof(T);
  if (smem > kMaxLds)
    return fail(QIDDM_ERR_UNSUPPORTED, "circuit with %lld Rot gates needs %zu B of LDS for the adjoint pass",
                (long long)n_rot, smem);
  auto kern = qiddm::wide_adjoint_kernel<T>;
  static qiddm_capi::DeviceFlags big_lds_enabled;
  if (smem > 48 * 1024 && !big_lds_enabled.get()) {
    const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    if (ea != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "hipFuncSetAttribute(LDS) failed: %s", hipGetErrorString(ea));
    big_lds_enabled.set();
  }
  qiddm::WideAdjointScalars ad;
  ad.gin_ld = gin_ld;
  ad.n = c->n_qubits;
  ad.want_inputs = (!raw && grad_inputs != nullptr && c->encoding != QIDDM_ENC_NONE) ? 1 : 0;
  ad.raw = raw ? 1 : 0;
  ad.pad_ = 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)wide_adjoint_blocks(p.batch)), dim3(qiddm::kWideThreads), smem, st,
                     static_cast<const T*>(inputs), static_cast<const T*>(table), static_cast<const T*>(gout),
                     static_cast<T*>(k_partials), static_cast<T*>(grad_inputs), static_cast<qiddm::V2<T>*>(ws), p, ad);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "wide_adjoint_kernel launch failed: %s", hipGetErrorString(e));
  return QIDDM_OK;
}

}  // namespace

extern "C" {

int qiddm_abi_version(void) { return QIDDM_ABI_VERSION; }
int qiddm_set_stamp_buffer(void* device_ptr, int64_t n_words) { return qiddm_capi::set_stamp_buffer(device_ptr, n_words); }
int qiddm_max_qubits(void) { return QIDDM_MAX_QUBITS; }

int64_t qiddm_workspace_bytes(const qiddm_circuit_t* c, int64_t batch, int64_t n_replicas) {
  if (check_circuit(c) != QIDDM_OK) return -1;
  if (batch < 0 || n_replicas < 0) return -1;
  if (c->n_qubits <= QIDDM_MAX_QUBITS_FUSED || batch == 0) return 0;
  return tiled_workspace_bytes(c, batch, n_replicas);
}
const char* qiddm_last_error(void) { return g_err; }

int64_t qiddm_num_rot_gates(const qiddm_circuit_t* c) {
  if (check_circuit(c) != QIDDM_OK) return -1;
  return (int64_t)c->n_rounds * c->n_blocks * c->sel_layers * c->n_qubits;
}

int64_t qiddm_gate_count(const qiddm_circuit_t* c) {
  if (check_circuit(c) != QIDDM_OK) return -1;
  const int64_t n = c->n_qubits;
  int64_t per_block = (int64_t)c->sel_layers * (n + (n > 1 ? n : 0));
  if (c->encoding == QIDDM_ENC_RZ) per_block += n;
  int64_t g = (int64_t)c->n_rounds * c->n_blocks * per_block;
  if (c->encoding == QIDDM_ENC_RY) g += (int64_t)c->n_rounds * n;
  if (c->encoding == QIDDM_ENC_RY_BLOCKS) g += (int64_t)c->n_rounds * c->n_blocks * n;
  if (c->encoding == QIDDM_ENC_AMPLITUDE) g += c->n_rounds;
  return g;
}

// entries (complex numbers) of the folded per-layer tables appended to the gate table (n <= 10 only)
static int64_t fold_entries(const qiddm_circuit_t* c, int64_t n_rot) {
  if (c->n_qubits > QIDDM_MAX_QUBITS_FUSED)   // wide CZ forward: (cos, sin)(theta/2) and (cos, sin)(alpha/2) per layer and wire
    return qiddm_capi::wide_cz_eligible(c) ? 2 * n_rot : 0;
  const int n = c->n_qubits, lb = n < 6 ? n : 6;
  int64_t e = (n_rot / n) * (n + ((int64_t)1 << lb) + ((int64_t)1 << (n - lb)));
  if (qiddm_capi::cz10_tables(c)) e += 2 * n_rot;   // + the per-wire tables of cz10_adjoint_kernel
  return e;
}

int64_t qiddm_gate_table_elems(const qiddm_circuit_t* c) {
  const int64_t g = qiddm_num_rot_gates(c);
  return g < 0 ? g : g * qiddm::kVariants * qiddm::kGateReals + 2 * fold_entries(c, g);
}

int64_t qiddm_num_shift_replicas(const qiddm_circuit_t* c, int with_inputs) {
  const int64_t g = qiddm_num_rot_gates(c);
  if (g < 0) return g;
  int64_t r = 6 * g;
  if (with_inputs && c->encoding >= QIDDM_ENC_RZ)
    r += 2 * (int64_t)c->n_blocks * c->n_qubits;
  return r;
}

int qiddm_prepare_gates(const qiddm_circuit_t* c, const double* angles, void* gate_table,
                        void* stream) {
  int rc = check_circuit(c);
  if (rc != QIDDM_OK) return rc;
  if (!angles || !gate_table) return fail(QIDDM_ERR_INVALID, "angles/gate_table is NULL");
  const int64_t n_rot = (int64_t)c->n_rounds * c->n_blocks * c->sel_layers * c->n_qubits;
  const int64_t fe = fold_entries(c, n_rot);
  const int64_t total = n_rot * qiddm::kVariants + fe;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  const int lpr = c->n_blocks * c->sel_layers;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (c->dtype == QIDDM_F32)
    hipLaunchKernelGGL(qiddm::prepare_gates_kernel<float>, dim3(blocks), dim3(256), 0, st, angles,
                       static_cast<float*>(gate_table), n_rot, c->n_qubits, lpr, fe);
  else
    hipLaunchKernelGGL(qiddm::prepare_gates_kernel<double>, dim3(blocks), dim3(256), 0, st, angles,
                       static_cast<double*>(gate_table), n_rot, c->n_qubits, lpr, fe);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess)
    return fail(QIDDM_ERR_LAUNCH, "prepare_gates launch failed: %s", hipGetErrorString(e));
  return QIDDM_OK;
}

int qiddm_forward_post(const qiddm_circuit_t* c, const void* inputs, int64_t batch, int64_t in_ld,
                       const void* gate_table, double* out, int64_t out_ld, int32_t post_cols, double post_scale,
                       void* stream) {
  int rc = check_circuit(c);
  if (rc != QIDDM_OK) return rc;
  if (c->measure != QIDDM_MEAS_PROBS || c->n_qubits > QIDDM_MAX_QUBITS_FUSED)
    return fail(QIDDM_ERR_UNSUPPORTED, "post-processed read-out: probabilities of an n <= %d circuit", QIDDM_MAX_QUBITS_FUSED);
  if (batch < 0) return fail(QIDDM_ERR_INVALID, "batch=%lld < 0", (long long)batch);
  if (post_cols < 1 || post_cols > (1 << c->n_qubits))
    return fail(QIDDM_ERR_INVALID, "post_cols=%d outside 1 .. 2^n", post_cols);
  if (out_ld < post_cols) return fail(QIDDM_ERR_INVALID, "out_ld=%lld < post_cols=%d", (long long)out_ld, post_cols);
  if (batch == 0) return QIDDM_OK;
  if (!gate_table || !out) return fail(QIDDM_ERR_INVALID, "gate_table/out is NULL");
  if (c->encoding != QIDDM_ENC_NONE) {
    if (!inputs) return fail(QIDDM_ERR_INVALID, "inputs is NULL but the encoding reads them");
    if (in_ld < c->n_features)
      return fail(QIDDM_ERR_INVALID, "in_ld=%lld < n_features=%d", (long long)in_ld, c->n_features);
  }
  qiddm::KScalars p = make_params(c);
  Ptrs ptr;
  ptr.inputs = inputs;
  ptr.table = gate_table;
  ptr.out = out;
  p.in_ld = in_ld;
  p.out_ld = out_ld;          // in float64 elements
  p.batch = batch;
  p.post_cols = post_cols;
  p.post_scale = post_scale;
  hipStream_t st = static_cast<hipStream_t>(stream);
  return c->dtype == QIDDM_F32 ? dispatch_n<float, false>(c->n_qubits, ptr, p, 0, st)
                               : dispatch_n<double, false>(c->n_qubits, ptr, p, 0, st);
}

int qiddm_forward(const qiddm_circuit_t* c, const void* inputs, int64_t batch, int64_t in_ld,
                  const void* gate_table, void* out, int64_t out_ld, void* workspace,
                  int64_t workspace_bytes, void* stream) {
  int rc = check_circuit(c);
  if (rc != QIDDM_OK) return rc;
  if (batch < 0) return fail(QIDDM_ERR_INVALID, "batch=%lld < 0", (long long)batch);
  if (batch == 0) return QIDDM_OK;
  if (!gate_table || !out) return fail(QIDDM_ERR_INVALID, "gate_table/out is NULL");
  if (c->encoding != QIDDM_ENC_NONE) {
    if (!inputs) return fail(QIDDM_ERR_INVALID, "inputs is NULL but the encoding reads them");
    if (in_ld < c->n_features)
      return fail(QIDDM_ERR_INVALID, "in_ld=%lld < n_features=%d", (long long)in_ld, c->n_features);
  }
  if (out_ld < out_cols(c))
    return fail(QIDDM_ERR_INVALID, "out_ld=%lld < %lld output columns", (long long)out_ld,
                (long long)out_cols(c));
  qiddm::KScalars p = make_params(c);
  Ptrs ptr;
  ptr.inputs = inputs;
  ptr.table = gate_table;
  ptr.out = out;
  p.in_ld = in_ld;
  p.out_ld = out_ld;
  p.batch = batch;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (c->n_qubits > QIDDM_MAX_QUBITS_FUSED) {
    rc = check_workspace(c, batch, 0, workspace, workspace_bytes);
    if (rc != QIDDM_OK) return rc;
    static const bool force_tiled = std::getenv("QIDDM_WIDE_TILED") != nullptr;   // kernel experiments: A/B
    if (qiddm_capi::wide_cz_eligible(c) && !force_tiled) {
      const int64_t n_rot = (int64_t)c->n_rounds * c->n_blocks * c->sel_layers * c->n_qubits;
      const size_t esz = c->dtype == QIDDM_F32 ? 4 : 8;
      const char* tail = static_cast<const char*>(gate_table) + (size_t)n_rot * qiddm::kVariants * qiddm::kGateReals * esz;
      return qiddm_capi::launch_wide_cz(c->dtype, c->n_qubits, inputs, tail, out, workspace, p, 2 * tiled_blocks_x(batch, 0), stream);
    }
    return c->dtype == QIDDM_F32 ? launch_tiled<float, false>(c->n_qubits, ptr, p, 0, workspace, st)
                                 : launch_tiled<double, false>(c->n_qubits, ptr, p, 0, workspace, st);
  }
  return c->dtype == QIDDM_F32 ? dispatch_n<float, false>(c->n_qubits, ptr, p, 0, st)
                               : dispatch_n<double, false>(c->n_qubits, ptr, p, 0, st);
}

int qiddm_forward_shifted(const qiddm_circuit_t* c, const void* inputs, int64_t batch, int64_t in_ld,
                          const void* gate_table, const void* grad_out, int64_t g_ld,
                          int64_t first_replica, int64_t n_replicas, void* dots, void* workspace,
                          int64_t workspace_bytes, void* stream) {
  int rc = check_circuit(c);
  if (rc != QIDDM_OK) return rc;
  if (c->n_rounds != 1)
    return fail(QIDDM_ERR_UNSUPPORTED,
                "parameter-shift sweeps run one QNode round at a time (n_rounds=%d)", c->n_rounds);
  if (batch < 0 || n_replicas < 0 || first_replica < 0)
    return fail(QIDDM_ERR_INVALID, "negative batch/replica range");
  const int64_t total = qiddm_num_shift_replicas(c, 1);
  if (first_replica + n_replicas > total)
    return fail(QIDDM_ERR_INVALID, "replicas [%lld, %lld) exceed the schedule of %lld",
                (long long)first_replica, (long long)(first_replica + n_replicas), (long long)total);
  if (n_replicas > 65535)
    return fail(QIDDM_ERR_INVALID, "at most 65535 replicas per call (got %lld)", (long long)n_replicas);
  if (batch == 0 || n_replicas == 0) return QIDDM_OK;
  if (!gate_table || !grad_out || !dots)
    return fail(QIDDM_ERR_INVALID, "gate_table/grad_out/dots is NULL");
  if (c->encoding != QIDDM_ENC_NONE) {
    if (!inputs) return fail(QIDDM_ERR_INVALID, "inputs is NULL but the encoding reads them");
    if (in_ld < c->n_features)
      return fail(QIDDM_ERR_INVALID, "in_ld=%lld < n_features=%d", (long long)in_ld, c->n_features);
  }
  if (g_ld < out_cols(c))
    return fail(QIDDM_ERR_INVALID, "g_ld=%lld < %lld output columns", (long long)g_ld,
                (long long)out_cols(c));
  qiddm::KScalars p = make_params(c);
  Ptrs ptr;
  ptr.inputs = inputs;
  ptr.table = gate_table;
  ptr.gout = grad_out;
  ptr.dots = dots;
  p.in_ld = in_ld;
  p.g_ld = g_ld;
  p.batch = batch;
  p.first_replica = (int32_t)first_replica;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (c->n_qubits > QIDDM_MAX_QUBITS_FUSED) {
    rc = check_workspace(c, batch, n_replicas, workspace, workspace_bytes);
    if (rc != QIDDM_OK) return rc;
    return c->dtype == QIDDM_F32
               ? launch_tiled<float, true>(c->n_qubits, ptr, p, n_replicas, workspace, st)
               : launch_tiled<double, true>(c->n_qubits, ptr, p, n_replicas, workspace, st);
  }
  return c->dtype == QIDDM_F32 ? dispatch_n<float, true>(c->n_qubits, ptr, p, n_replicas, st)
                               : dispatch_n<double, true>(c->n_qubits, ptr, p, n_replicas, st);
}

int64_t qiddm_adjoint_partials(const qiddm_circuit_t* c, int64_t batch) {
  if (check_circuit(c) != QIDDM_OK || batch < 0) return -1;
  if (c->n_qubits > QIDDM_MAX_QUBITS_FUSED) return wide_adjoint_blocks(batch);
  return adjoint_blocks_n(c->n_qubits, batch);
}

int64_t qiddm_adjoint_workspace_bytes(const qiddm_circuit_t* c, int64_t batch) {
  if (check_circuit(c) != QIDDM_OK || batch < 0) return -1;
  if (c->n_qubits <= QIDDM_MAX_QUBITS_FUSED) return 0;
  return wide_adjoint_blocks(batch) * 2 * ((int64_t)1 << c->n_qubits) * (c->dtype == QIDDM_F32 ? 8 : 16);
}

int qiddm_backward_adjoint_wide(const qiddm_circuit_t* c, const void* inputs, int64_t batch, int64_t in_ld,
                                const void* gate_table, const void* grad_out, int64_t g_ld, void* k_partials,
                                void* grad_inputs, int64_t gin_ld, void* workspace, int64_t workspace_bytes,
                                void* stream) {
  int rc = check_circuit(c);
  if (rc != QIDDM_OK) return rc;
  if (c->n_rounds != 1)
    return fail(QIDDM_ERR_UNSUPPORTED, "the adjoint pass differentiates one QNode round (n_rounds=%d)", c->n_rounds);
  if (c->n_qubits <= QIDDM_MAX_QUBITS_FUSED)
    return fail(QIDDM_ERR_UNSUPPORTED, "n_qubits=%d: use qiddm_backward_adjoint (register-resident)", c->n_qubits);
  if (batch < 0) return fail(QIDDM_ERR_INVALID, "batch < 0");
  if (!gate_table || !grad_out || !k_partials) return fail(QIDDM_ERR_INVALID, "gate_table/grad_out/k_partials is NULL");
  if (c->encoding != QIDDM_ENC_NONE && batch > 0 && !inputs) return fail(QIDDM_ERR_INVALID, "inputs is NULL");
  if (g_ld < out_cols(c)) return fail(QIDDM_ERR_INVALID, "g_ld smaller than the output row");
  const int64_t need = qiddm_adjoint_workspace_bytes(c, batch);
  if (!workspace || workspace_bytes < need)
    return fail(QIDDM_ERR_INVALID, "workspace of %lld B needed (qiddm_adjoint_workspace_bytes), got %lld",
                (long long)need, (long long)workspace_bytes);
  qiddm::KScalars p = make_params(c);
  p.in_ld = in_ld;
  p.g_ld = g_ld;
  p.batch = batch;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (batch > 0 && qiddm_capi::wide_cz_adjoint_eligible(c)) {
    // pass-structured reverse sweep; the slabs then hold per-layer angle-gradient sums (qiddm_adjoint_finalize knows)
    const int64_t n_rot = (int64_t)c->n_blocks * c->sel_layers * c->n_qubits;
    const size_t esz = c->dtype == QIDDM_F32 ? 4 : 8;
    const char* tail = static_cast<const char*>(gate_table) + (size_t)n_rot * qiddm::kVariants * qiddm::kGateReals * esz;
    return qiddm_capi::launch_wide_cz_adjoint(c->dtype, c->n_qubits, inputs, tail, grad_out, k_partials, n_rot * 8,
                                              grad_inputs, gin_ld, workspace, p, wide_adjoint_blocks(batch), stream);
  }
  return c->dtype == QIDDM_F32
             ? launch_wide_adjoint<float>(c, inputs, gate_table, grad_out, k_partials, grad_inputs, workspace, p, gin_ld, st)
             : launch_wide_adjoint<double>(c, inputs, gate_table, grad_out, k_partials, grad_inputs, workspace, p, gin_ld, st);
}

int64_t qiddm_matrix_adjoint_partials(int64_t count) { return count < 0 ? -1 : wide_adjoint_blocks(count); }

int64_t qiddm_matrix_adjoint_workspace_bytes(const qiddm_circuit_t* c, int64_t count) {
  if (check_circuit(c) != QIDDM_OK || count < 0) return -1;
  return wide_adjoint_blocks(count) * 2 * ((int64_t)1 << c->n_qubits) * 16;
}

// The slab format is a function of the descriptor alone (and of the kernel-experiment switches read by make_params /
// *_eligible): the folded reverse sweeps write per-layer angle-gradient sums, every other producer writes K matrices.
// qiddm_adjoint_finalize decides by this predicate, so every PRODUCER of slabs has to agree with it.
static bool finalize_reads_folded_slabs(const qiddm_circuit_t* c) {
  return (make_params(c).fold && c->n_qubits >= 2 && c->n_qubits <= qiddm::kFoldedAdjointMaxQubits) ||
         (c->n_qubits > QIDDM_MAX_QUBITS_FUSED && qiddm_capi::wide_cz_adjoint_eligible(c)) ||
         qiddm_capi::cz10_adjoint_eligible(c);
}

int qiddm_matrix_adjoint(const qiddm_circuit_t* c, const double* psi0, const double* lambda, int64_t count,
                         const void* gate_table, void* k_partials, void* workspace, int64_t workspace_bytes,
                         void* stream) {
  int rc = check_circuit(c);
  if (rc != QIDDM_OK) return rc;
  // this entry point always writes K slabs (per-gate sweep); a descriptor that qiddm_adjoint_finalize reads as the
  // folded layout would come back with wrong gradients and no error -- refuse it here instead
  if (finalize_reads_folded_slabs(c))
    return fail(QIDDM_ERR_UNSUPPORTED, "matrix-element sweep writes K slabs, but qiddm_adjoint_finalize reads the "
                "folded per-layer layout for this circuit (n=%d, %s ring, encoding %d): use a CNOT-ring descriptor "
                "or qiddm_backward_adjoint[_wide]", c->n_qubits, c->imprimitive == QIDDM_IMP_CZ ? "CZ" : "CNOT",
                c->encoding);
  if (c->n_rounds != 1) return fail(QIDDM_ERR_UNSUPPORTED, "one round only (n_rounds=%d)", c->n_rounds);
  if (c->dtype != QIDDM_F64) return fail(QIDDM_ERR_UNSUPPORTED, "the matrix-element sweep runs in float64");
  if (c->n_qubits < 2) return fail(QIDDM_ERR_UNSUPPORTED, "the matrix-element sweep needs n_qubits >= 2");
  if (count < 0) return fail(QIDDM_ERR_INVALID, "count < 0");
  if (!gate_table || !k_partials) return fail(QIDDM_ERR_INVALID, "gate_table/k_partials is NULL");
  if (count > 0 && (!psi0 || !lambda)) return fail(QIDDM_ERR_INVALID, "psi0/lambda is NULL");
  const int64_t need = qiddm_matrix_adjoint_workspace_bytes(c, count);
  if (!workspace || workspace_bytes < need)
    return fail(QIDDM_ERR_INVALID, "workspace of %lld B needed (qiddm_matrix_adjoint_workspace_bytes), got %lld",
                (long long)need, (long long)workspace_bytes);
  qiddm::KScalars p = make_params(c);
  p.in_ld = (int64_t)2 << c->n_qubits;
  p.g_ld = p.in_ld;
  p.batch = count;
  p.encoding = QIDDM_ENC_NONE;  // weight-only layers; the start vector is given
  return launch_wide_adjoint<double>(c, psi0, gate_table, lambda, k_partials, nullptr, workspace, p, 0,
                                     static_cast<hipStream_t>(stream), true);
}

int qiddm_backward_adjoint(const qiddm_circuit_t* c, const void* inputs, int64_t batch, int64_t in_ld,
                           const void* gate_table, const void* grad_out, int64_t g_ld, void* k_partials,
                           void* grad_inputs, int64_t gin_ld, void* stream) {
  int rc = check_circuit(c);
  if (rc != QIDDM_OK) return rc;
  if (c->n_rounds != 1)
    return fail(QIDDM_ERR_UNSUPPORTED, "the adjoint pass differentiates one QNode round (n_rounds=%d)",
                c->n_rounds);
  if (c->n_qubits > QIDDM_MAX_QUBITS_FUSED)
    return fail(QIDDM_ERR_UNSUPPORTED, "adjoint backward needs n_qubits <= %d (got %d); use the "
                "parameter-shift sweep", QIDDM_MAX_QUBITS_FUSED, c->n_qubits);
  if (batch < 0) return fail(QIDDM_ERR_INVALID, "batch < 0");
  if (!gate_table || !grad_out || !k_partials) return fail(QIDDM_ERR_INVALID, "gate_table/grad_out/k_partials is NULL");
  if (c->encoding != QIDDM_ENC_NONE) {
    if (!inputs) return fail(QIDDM_ERR_INVALID, "inputs is NULL but the encoding reads them");
    if (in_ld < c->n_features)
      return fail(QIDDM_ERR_INVALID, "in_ld=%lld < n_features=%d", (long long)in_ld, c->n_features);
  }
  if (g_ld < out_cols(c)) return fail(QIDDM_ERR_INVALID, "g_ld smaller than the output width");
  const int64_t gin_cols = c->encoding == QIDDM_ENC_AMPLITUDE ? c->n_features : c->n_qubits;
  if (grad_inputs && gin_ld < gin_cols) return fail(QIDDM_ERR_INVALID, "gin_ld=%lld < %lld", (long long)gin_ld, (long long)gin_cols);
  qiddm::KScalars p = make_params(c);
  Ptrs ptr;
  ptr.inputs = inputs;
  ptr.table = gate_table;
  ptr.gout = grad_out;
  p.in_ld = in_ld;
  p.g_ld = g_ld;
  p.batch = batch;
  qiddm::AdjointScalars ad;
  ad.gin_ld = gin_ld;
  ad.want_inputs = (grad_inputs != nullptr && c->encoding != QIDDM_ENC_NONE) ? 1 : 0;
  ad.pad_ = 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (batch > 0 && qiddm_capi::cz10_adjoint_eligible(c)) {
    // register-resident folded reverse sweep; the slabs then hold per-layer angle-gradient sums (finalize knows)
    const int64_t n_rot = (int64_t)c->n_blocks * c->sel_layers * c->n_qubits;
    const int64_t layers = n_rot / c->n_qubits;
    const size_t esz = c->dtype == QIDDM_F32 ? 4 : 8;
    const int64_t fold_reals = 2 * layers * (10 + 64 + 16);
    const char* tail = static_cast<const char*>(gate_table) +
                       ((size_t)n_rot * qiddm::kVariants * qiddm::kGateReals + (size_t)fold_reals) * esz;
    return qiddm_capi::launch_cz10_adjoint(c->dtype, inputs, tail, grad_out, k_partials, n_rot * 8,
                                           ad.want_inputs ? grad_inputs : nullptr, gin_ld, p, adjoint_blocks_n(10, batch),
                                           stream);
  }
  // batch == 0 still has to zero the (single) partial slab: launch with no samples
  const ConvPtrs none;
  return c->dtype == QIDDM_F32
             ? dispatch_adjoint<float, false>(c->n_qubits, ptr, k_partials, grad_inputs, p, ad, none, st)
             : dispatch_adjoint<double, false>(c->n_qubits, ptr, k_partials, grad_inputs, p, ad, none, st);
}

int qiddm_qconv_backward(const qiddm_circuit_t* c, const double* x, int64_t batch, int64_t in_channels,
                         int64_t height, int64_t width, int64_t kh, int64_t kw, int64_t pad_h, int64_t pad_w,
                         const void* gate_table, const double* grad_y, int64_t out_channels, void* k_partials,
                         void* grad_features, double* grad_x, void* stream) {
  int rc = check_circuit(c);
  if (rc != QIDDM_OK) return rc;
  if (c->encoding != QIDDM_ENC_AMPLITUDE || c->measure != QIDDM_MEAS_PROBS || c->n_rounds != 1 ||
      c->n_blocks != 1)
    return fail(QIDDM_ERR_UNSUPPORTED, "QConv2d needs amplitude encoding, probs, one round, one block");
  if (c->n_qubits > QIDDM_MAX_QUBITS_FUSED)
    return fail(QIDDM_ERR_UNSUPPORTED, "fused QConv2d backward needs n_qubits <= %d (got %d)",
                QIDDM_MAX_QUBITS_FUSED, c->n_qubits);
  if (batch < 0 || in_channels < 1 || height < 1 || width < 1 || kh < 1 || kw < 1 || pad_h < 0 || pad_w < 0 ||
      out_channels < 1)
    return fail(QIDDM_ERR_INVALID, "bad convolution geometry");
  if (in_channels * kh * kw != c->n_features)
    return fail(QIDDM_ERR_INVALID, "n_features=%d != in_channels*kh*kw=%lld", c->n_features,
                (long long)(in_channels * kh * kw));
  const int64_t ho = height + 2 * pad_h - kh + 1, wo = width + 2 * pad_w - kw + 1;
  if (ho < 1 || wo < 1) return fail(QIDDM_ERR_INVALID, "kernel larger than the padded image");
  const int64_t d = (int64_t)1 << c->n_qubits;
  if (2 * out_channels > d && !(d == 2 && out_channels == 1))
    return fail(QIDDM_ERR_INVALID, "out_channels=%lld exceeds the %lld even-index probabilities",
                (long long)out_channels, (long long)(d / 2));
  if (batch * ho * wo >= ((int64_t)1 << 40)) return fail(QIDDM_ERR_INVALID, "too many output pixels");
  if (!gate_table || !k_partials) return fail(QIDDM_ERR_INVALID, "gate_table/k_partials is NULL");
  if (batch > 0 && (!x || !grad_y)) return fail(QIDDM_ERR_INVALID, "x/grad_y is NULL");
  if ((grad_x != nullptr) != (grad_features != nullptr))
    return fail(QIDDM_ERR_INVALID, "grad_x and grad_features go together (both or neither)");
  qiddm::KScalars p = make_params(c);
  p.batch = batch * ho * wo;  // one circuit per output pixel
  p.in_ld = c->n_features;
  p.g_ld = d;
  ConvPtrs conv;
  conv.img = x;
  conv.gy = grad_y;
  std::memset(&conv.cv, 0, sizeof(conv.cv));
  conv.cv.C = (int32_t)in_channels;
  conv.cv.H = (int32_t)height;
  conv.cv.W = (int32_t)width;
  conv.cv.kh = (int32_t)kh;
  conv.cv.kw = (int32_t)kw;
  conv.cv.ph = (int32_t)pad_h;
  conv.cv.pw = (int32_t)pad_w;
  conv.cv.Ho = (int32_t)ho;
  conv.cv.Wo = (int32_t)wo;
  conv.cv.C_out = (int32_t)out_channels;
  conv.cv.post_scale = 0.5 * (double)d;
  qiddm::AdjointScalars ad;
  ad.gin_ld = c->n_features;
  ad.want_inputs = grad_x != nullptr ? 1 : 0;
  ad.pad_ = 0;
  Ptrs ptr;
  ptr.table = gate_table;
  hipStream_t st = static_cast<hipStream_t>(stream);
  rc = c->dtype == QIDDM_F32
           ? dispatch_adjoint<float, true>(c->n_qubits, ptr, k_partials, grad_features, p, ad, conv, st)
           : dispatch_adjoint<double, true>(c->n_qubits, ptr, k_partials, grad_features, p, ad, conv, st);
  if (rc != QIDDM_OK || grad_x == nullptr || batch == 0) return rc;
  const int64_t total = batch * in_channels * height * width;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (c->dtype == QIDDM_F32)
    hipLaunchKernelGGL(qiddm::qconv_fold_kernel<float>, dim3(blocks), dim3(256), 0, st,
                       static_cast<const float*>(grad_features), grad_x, total, conv.cv);
  else
    hipLaunchKernelGGL(qiddm::qconv_fold_kernel<double>, dim3(blocks), dim3(256), 0, st,
                       static_cast<const double*>(grad_features), grad_x, total, conv.cv);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "qconv_fold launch failed: %s", hipGetErrorString(e));
  return QIDDM_OK;
}

int qiddm_adjoint_finalize(const qiddm_circuit_t* c, const double* angles, const void* k_partials,
                           int64_t n_partials, double* grad_angles, void* stream) {
  int rc = check_circuit(c);
  if (rc != QIDDM_OK) return rc;
  if (!angles || !k_partials || !grad_angles) return fail(QIDDM_ERR_INVALID, "angles/k_partials/grad_angles is NULL");
  if (n_partials < 0) return fail(QIDDM_ERR_INVALID, "n_partials < 0");
  const int64_t n_rot = (int64_t)c->n_rounds * c->n_blocks * c->sel_layers * c->n_qubits;
  const unsigned blocks = (unsigned)n_rot;  // one wavefront per gate
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (finalize_reads_folded_slabs(c)) {
    // the slabs hold per-layer angle-gradient sums (folded reverse sweep), not K
    const int slots = c->n_qubits <= 8 ? 8 : 16;
    if (c->dtype == QIDDM_F32)
      hipLaunchKernelGGL(qiddm::adjoint_finalize_folded_kernel<float>, dim3(blocks), dim3(qiddm::kWave), 0, st,
                         static_cast<const float*>(k_partials), n_partials, n_rot * 8, c->n_qubits,
                         c->n_blocks * c->sel_layers, slots, n_rot, grad_angles);
    else
      hipLaunchKernelGGL(qiddm::adjoint_finalize_folded_kernel<double>, dim3(blocks), dim3(qiddm::kWave), 0, st,
                         static_cast<const double*>(k_partials), n_partials, n_rot * 8, c->n_qubits,
                         c->n_blocks * c->sel_layers, slots, n_rot, grad_angles);
    const hipError_t ef = hipGetLastError();
    if (ef != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "adjoint_finalize launch failed: %s", hipGetErrorString(ef));
    return QIDDM_OK;
  }
  if (c->dtype == QIDDM_F32)
    hipLaunchKernelGGL(qiddm::adjoint_finalize_kernel<float>, dim3(blocks), dim3(qiddm::kWave), 0, st,
                       static_cast<const float*>(k_partials), n_partials, n_rot, angles, grad_angles);
  else
    hipLaunchKernelGGL(qiddm::adjoint_finalize_kernel<double>, dim3(blocks), dim3(qiddm::kWave), 0, st,
                       static_cast<const double*>(k_partials), n_partials, n_rot, angles, grad_angles);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "adjoint_finalize launch failed: %s", hipGetErrorString(e));
  return QIDDM_OK;
}

int qiddm_dense_forward(const qiddm_circuit_t* c, const double* x, int64_t batch, int64_t x_ld,
                        int64_t in_features, const double* w_down, const double* b_down,
                        const double* angles, const double* w_up, const double* b_up,
                        int64_t out_features, int32_t post_mode, double noise_factor, double* y,
                        int64_t y_ld, void* stream) {
  int rc = check_circuit(c);
  if (rc != QIDDM_OK) return rc;
  if (c->encoding != QIDDM_ENC_RZ || c->measure != QIDDM_MEAS_EXPZ)
    return fail(QIDDM_ERR_UNSUPPORTED, "dense forward needs the RZ encoding and the <Z> read-out");
  if (c->n_qubits > QIDDM_MAX_QUBITS_FUSED)
    return fail(QIDDM_ERR_UNSUPPORTED, "dense forward is fused for n <= %d; use qiddm_forward for n=%d",
                QIDDM_MAX_QUBITS_FUSED, c->n_qubits);
  if (batch < 0) return fail(QIDDM_ERR_INVALID, "batch=%lld < 0", (long long)batch);
  if (in_features < 1 || out_features < 1 || in_features > (1 << 24) || out_features > (1 << 24))
    return fail(QIDDM_ERR_INVALID, "bad feature counts %lld / %lld", (long long)in_features,
                (long long)out_features);
  if (post_mode != 0 && post_mode != 1) return fail(QIDDM_ERR_INVALID, "post_mode must be 0 or 1");
  if (post_mode == 1 && in_features != out_features)
    return fail(QIDDM_ERR_INVALID, "post_mode 1 needs out_features == in_features");
  if (batch == 0) return QIDDM_OK;
  if (!x || !w_down || !angles || !w_up || !y)
    return fail(QIDDM_ERR_INVALID, "x/w_down/angles/w_up/y is NULL");
  if (x == y) return fail(QIDDM_ERR_INVALID, "y must not alias x");
  if (x_ld < in_features || y_ld < out_features)
    return fail(QIDDM_ERR_INVALID, "row strides smaller than the feature counts");
  static const bool no_quad = std::getenv("QIDDM_NO_QUAD") != nullptr;  // tuning switch
  if (!no_quad && batch <= 1024 && quad_supported(c, in_features, out_features))
    return qiddm_dense_sample(c, x, batch, x_ld, in_features, w_down, b_down, angles, w_up, b_up, out_features,
                              post_mode, noise_factor, 1, y, y_ld, batch * y_ld, nullptr, stream);
  qiddm::KScalars p = make_params(c);
  p.batch = batch;
  qiddm::DenseScalars d;
  std::memset(&d, 0, sizeof(d));
  d.x_ld = x_ld;
  d.y_ld = y_ld;
  d.in_features = (int32_t)in_features;
  d.out_features = (int32_t)out_features;
  d.post_mode = post_mode;
  d.noise_factor = noise_factor;
  d.stamps = qiddm_capi::stamp_buffer(8);
  hipStream_t st = static_cast<hipStream_t>(stream);
  return c->dtype == QIDDM_F32
             ? dispatch_dense<float>(c->n_qubits, x, w_down, b_down, angles, w_up, b_up, y, d, p, st)
             : dispatch_dense<double>(c->n_qubits, x, w_down, b_down, angles, w_up, b_up, y, d, p, st);
}

int qiddm_qconv_forward(const qiddm_circuit_t* c, const double* x, int64_t batch, int64_t in_channels,
                        int64_t height, int64_t width, int64_t kh, int64_t kw, int64_t pad_h,
                        int64_t pad_w, const double* angles, int64_t out_channels, double* y,
                        void* stream) {
  int rc = check_circuit(c);
  if (rc != QIDDM_OK) return rc;
  if (c->encoding != QIDDM_ENC_AMPLITUDE || c->measure != QIDDM_MEAS_PROBS || c->n_rounds != 1 ||
      c->n_blocks != 1)
    return fail(QIDDM_ERR_UNSUPPORTED, "QConv2d needs amplitude encoding, probs, one round, one block");
  if (batch < 0 || in_channels < 1 || height < 1 || width < 1 || kh < 1 || kw < 1 || pad_h < 0 || pad_w < 0 ||
      out_channels < 1)
    return fail(QIDDM_ERR_INVALID, "bad convolution geometry");
  if (in_channels * kh * kw != c->n_features)
    return fail(QIDDM_ERR_INVALID, "n_features=%d != in_channels*kh*kw=%lld", c->n_features,
                (long long)(in_channels * kh * kw));
  const int64_t ho = height + 2 * pad_h - kh + 1, wo = width + 2 * pad_w - kw + 1;
  if (ho < 1 || wo < 1) return fail(QIDDM_ERR_INVALID, "kernel larger than the padded image");
  const int64_t d = (int64_t)1 << c->n_qubits;
  if (2 * out_channels > d && !(d == 2 && out_channels == 1))
    return fail(QIDDM_ERR_INVALID, "out_channels=%lld exceeds the %lld even-index probabilities",
                (long long)out_channels, (long long)(d / 2));
  if (batch * ho * wo >= ((int64_t)1 << 40)) return fail(QIDDM_ERR_INVALID, "too many output pixels");
  if (batch == 0) return QIDDM_OK;
  if (!x || !angles || !y) return fail(QIDDM_ERR_INVALID, "x/angles/y is NULL");
  qiddm::KScalars p = make_params(c);
  p.batch = batch * ho * wo;  // one circuit per output pixel
  qiddm::ConvScalars cv;
  std::memset(&cv, 0, sizeof(cv));
  cv.C = (int32_t)in_channels;
  cv.H = (int32_t)height;
  cv.W = (int32_t)width;
  cv.kh = (int32_t)kh;
  cv.kw = (int32_t)kw;
  cv.ph = (int32_t)pad_h;
  cv.pw = (int32_t)pad_w;
  cv.Ho = (int32_t)ho;
  cv.Wo = (int32_t)wo;
  cv.C_out = (int32_t)out_channels;
  cv.post_scale = 0.5 * (double)d;
  hipStream_t st = static_cast<hipStream_t>(stream);
  return c->dtype == QIDDM_F32 ? dispatch_qconv<float>(c->n_qubits, x, angles, y, cv, p, st)
                               : dispatch_qconv<double>(c->n_qubits, x, angles, y, cv, p, st);
}

int qiddm_dense_sample(const qiddm_circuit_t* c, const double* x, int64_t batch, int64_t x_ld,
                       int64_t in_features, const double* w_down, const double* b_down,
                       const double* angles, const double* w_up, const double* b_up, int64_t out_features,
                       int32_t post_mode, double noise_factor, int32_t n_steps, double* y, int64_t y_ld,
                       int64_t y_step_stride, const void* tables, void* stream) {
  int rc = check_circuit(c);
  if (rc != QIDDM_OK) return rc;
  if (batch < 0 || n_steps < 0) return fail(QIDDM_ERR_INVALID, "negative batch / n_steps");
  if (in_features < 1 || out_features < 1) return fail(QIDDM_ERR_INVALID, "bad feature counts");
  if (!quad_supported(c, in_features, out_features))
    return fail(QIDDM_ERR_UNSUPPORTED, "fused sampling loop: needs 2 <= n <= 10, CZ, RZ encoding, <Z>, "
                "features <= 2048");
  if (post_mode != 0 && post_mode != 1) return fail(QIDDM_ERR_INVALID, "post_mode must be 0 or 1");
  if ((post_mode == 1 || n_steps > 1) && in_features != out_features)
    return fail(QIDDM_ERR_INVALID, "chained steps / post_mode 1 need out_features == in_features");
  if (batch == 0 || n_steps == 0) return QIDDM_OK;
  if (!x || !w_down || !angles || !w_up || !y) return fail(QIDDM_ERR_INVALID, "x/w_down/angles/w_up/y is NULL");
  if (x == y) return fail(QIDDM_ERR_INVALID, "y must not alias x");
  if (x_ld < in_features || y_ld < out_features || y_step_stride < batch * y_ld - (y_ld - out_features))
    return fail(QIDDM_ERR_INVALID, "strides smaller than the tensor extents");
  qiddm::KScalars p = make_params(c);
  p.batch = batch;
  qiddm::QuadScalars d;
  std::memset(&d, 0, sizeof(d));
  d.x_ld = x_ld;
  d.y_ld = y_ld;
  d.y_step_stride = y_step_stride;
  d.in_features = (int32_t)in_features;
  d.out_features = (int32_t)out_features;
  d.post_mode = post_mode;
  d.n_steps = n_steps;
  d.noise_factor = noise_factor;
  d.stamps = qiddm_capi::stamp_buffer(8);
  hipStream_t st = static_cast<hipStream_t>(stream);
  return c->dtype == QIDDM_F32
             ? dispatch_quad<float>(c->n_qubits, x, w_down, b_down, angles, w_up, b_up, y, tables, d, p, st)
             : dispatch_quad<double>(c->n_qubits, x, w_down, b_down, angles, w_up, b_up, y, tables, d, p, st);
}

int64_t qiddm_dense_sample_tables_bytes(const qiddm_circuit_t* c) {
  if (check_circuit(c) != QIDDM_OK) return -1;
  if (!quad_supported(c, 1, 1)) {
    fail(QIDDM_ERR_UNSUPPORTED, "fused sampling loop: needs 8 <= n <= 10, CZ, RZ encoding, <Z>");
    return QIDDM_ERR_UNSUPPORTED;
  }
  const int64_t n_rot = (int64_t)c->n_rounds * c->n_blocks * c->sel_layers * c->n_qubits;
  return (int64_t)(c->dtype == QIDDM_F32 ? quad_tables_bytes<float>(c->n_qubits, n_rot)
                                         : quad_tables_bytes<double>(c->n_qubits, n_rot));
}

int qiddm_dense_sample_prepare(const qiddm_circuit_t* c, const double* angles, void* tables, void* stream) {
  const int64_t need = qiddm_dense_sample_tables_bytes(c);
  if (need < 0) return (int)need;
  if (!angles || !tables) return fail(QIDDM_ERR_INVALID, "angles/tables is NULL");
  const qiddm::KScalars p = make_params(c);
  const int64_t n_rot = (int64_t)c->n_rounds * c->n_blocks * c->sel_layers * c->n_qubits;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool f32 = c->dtype == QIDDM_F32;
  switch (c->n_qubits) {
    case 2: return f32 ? launch_quad_tables<float, 2>(angles, tables, p, n_rot, st) : launch_quad_tables<double, 2>(angles, tables, p, n_rot, st);
    case 3: return f32 ? launch_quad_tables<float, 3>(angles, tables, p, n_rot, st) : launch_quad_tables<double, 3>(angles, tables, p, n_rot, st);
    case 4: return f32 ? launch_quad_tables<float, 4>(angles, tables, p, n_rot, st) : launch_quad_tables<double, 4>(angles, tables, p, n_rot, st);
    case 5: return f32 ? launch_quad_tables<float, 5>(angles, tables, p, n_rot, st) : launch_quad_tables<double, 5>(angles, tables, p, n_rot, st);
    case 6: return f32 ? launch_quad_tables<float, 6>(angles, tables, p, n_rot, st) : launch_quad_tables<double, 6>(angles, tables, p, n_rot, st);
    case 7: return f32 ? launch_quad_tables<float, 7>(angles, tables, p, n_rot, st) : launch_quad_tables<double, 7>(angles, tables, p, n_rot, st);
    case 8: return f32 ? launch_quad_tables<float, 8>(angles, tables, p, n_rot, st) : launch_quad_tables<double, 8>(angles, tables, p, n_rot, st);
    case 9: return f32 ? launch_quad_tables<float, 9>(angles, tables, p, n_rot, st) : launch_quad_tables<double, 9>(angles, tables, p, n_rot, st);
    default: return f32 ? launch_quad_tables<float, 10>(angles, tables, p, n_rot, st) : launch_quad_tables<double, 10>(angles, tables, p, n_rot, st);
  }
}

}  // extern "C"
