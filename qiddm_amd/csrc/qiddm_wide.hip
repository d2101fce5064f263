// qiddm_wide.hip -- launch side of the wide CZ forward (qsim_wide_cz.h): n = 11..16 qubits, CZ entanglers, RZ
// re-upload (or no) encoding, probabilities or <Z>.  Reached from qiddm_forward (qiddm_capi.hip); a translation
// unit of its own because of the twelve kernel instantiations.
#include "capi_common.h"

#include <hip/hip_runtime.h>

#include <cstdlib>

#include "qsim_wide_cz.h"
#include "qsim_wide_cz_adjoint.h"

namespace qiddm_capi {

namespace {

template <typename T, int N>
int launch_n(const void* inputs, const void* tail, void* out, void* ws, const qiddm::KScalars& p, int64_t slabs,
             hipStream_t st) {
  const int64_t layers_all = (int64_t)p.n_rounds * p.n_blocks * p.sel_layers;
  const size_t smem = qiddm::WideCzSmem<T>::bytes(layers_all, N);
  if (smem > kMaxLds)
    return fail(QIDDM_ERR_UNSUPPORTED, "circuit with %lld layers needs %zu B of LDS for its layer tables (limit %zu)",
                (long long)layers_all, smem, kMaxLds);
  auto kern = qiddm::wide_cz_kernel<T, N>;
  static DeviceFlags big_lds_enabled;
  if (smem > 48 * 1024 && !big_lds_enabled.get()) {
    const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    if (ea != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "hipFuncSetAttribute(LDS) failed: %s", hipGetErrorString(ea));
    big_lds_enabled.set();
  }
  // launch shape from the sweep of tools/tune_wide.py (gpurun_out/r02c/tune_wide.log): four waves per workgroup and
  // as many resident workgroups as the workspace has slabs -- n = 16: 3.97 ms per 1024 samples at (4 waves, 768..1024
  // workgroups) against 4.1 at 512 and 7.6 at 128; 8 waves per workgroup is no better and 1.4x worse at n = 12
  constexpr int NT = qiddm::WideGeom<N>::NT;
  int waves = NT >= 4 ? 4 : NT;
  int64_t grid = wide_cz_grid(p.batch, slabs);
  // kernel experiments: QIDDM_WIDE_WAVES / QIDDM_WIDE_GRID override the launch shape (within the workspace)
  if (const char* e = std::getenv("QIDDM_WIDE_WAVES")) {
    const int v = std::atoi(e);
    if (v == 1 || v == 2 || v == 4 || v == 8) waves = v;
  }
  if (const char* e = std::getenv("QIDDM_WIDE_GRID")) {
    const int64_t v = std::atoll(e);
    if (v >= 1 && v <= slabs) grid = v < p.batch ? v : p.batch;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3((unsigned)(waves * qiddm::kWave)), smem, st,
                     static_cast<const T*>(inputs), static_cast<const T*>(tail), static_cast<T*>(out),
                     static_cast<qiddm::V2<T>*>(ws), p);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess)
    return fail(QIDDM_ERR_LAUNCH, "wide_cz_kernel<n=%d> launch failed: %s", N, hipGetErrorString(e));
  return QIDDM_OK;
}

template <typename T>
int launch_t(int n, const void* inputs, const void* tail, void* out, void* ws, const qiddm::KScalars& p,
             int64_t slabs, hipStream_t st) {
  switch (n) {
    case 11: return launch_n<T, 11>(inputs, tail, out, ws, p, slabs, st);
    case 12: return launch_n<T, 12>(inputs, tail, out, ws, p, slabs, st);
    case 13: return launch_n<T, 13>(inputs, tail, out, ws, p, slabs, st);
    case 14: return launch_n<T, 14>(inputs, tail, out, ws, p, slabs, st);
    case 15: return launch_n<T, 15>(inputs, tail, out, ws, p, slabs, st);
    case 16: return launch_n<T, 16>(inputs, tail, out, ws, p, slabs, st);
    default: return fail(QIDDM_ERR_UNSUPPORTED, "wide CZ forward covers 11..16 qubits (got %d)", n);
  }
}

template <typename T, int N>
int launch_adj_n(const void* inputs, const void* tail, const void* gout, void* partials, int64_t slab_stride,
                 void* grad_inputs, int64_t gin_ld, void* ws, const qiddm::KScalars& p, int64_t grid, hipStream_t st) {
  const int64_t layers = (int64_t)p.n_blocks * p.sel_layers;
  const size_t smem = qiddm::WideCzAdjSmem<T>::bytes(layers, N);
  if (smem > kMaxLds)
    return fail(QIDDM_ERR_UNSUPPORTED, "circuit with %lld layers needs %zu B of LDS for the reverse sweep (limit %zu)",
                (long long)layers, smem, kMaxLds);
  auto kern = qiddm::wide_cz_adjoint_kernel<T, N>;
  static DeviceFlags big_lds_enabled;
  if (smem > 48 * 1024 && !big_lds_enabled.get()) {
    const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    if (ea != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "hipFuncSetAttribute(LDS) failed: %s", hipGetErrorString(ea));
    big_lds_enabled.set();
  }
  constexpr int NT = qiddm::WideGeom<N>::NT;
  const int waves = NT >= 4 ? 4 : NT;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3((unsigned)(waves * qiddm::kWave)), smem, st,
                     static_cast<const T*>(inputs), static_cast<const T*>(tail), static_cast<const T*>(gout),
                     static_cast<T*>(partials), slab_stride, static_cast<T*>(grad_inputs), gin_ld,
                     static_cast<qiddm::V2<T>*>(ws), p);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess)
    return fail(QIDDM_ERR_LAUNCH, "wide_cz_adjoint_kernel<n=%d> launch failed: %s", N, hipGetErrorString(e));
  return QIDDM_OK;
}

template <typename T>
int launch_adj_t(int n, const void* inputs, const void* tail, const void* gout, void* partials, int64_t slab_stride,
                 void* grad_inputs, int64_t gin_ld, void* ws, const qiddm::KScalars& p, int64_t grid, hipStream_t st) {
  switch (n) {
#define QIDDM_ADJ_CASE(NN) \
    case NN: return launch_adj_n<T, NN>(inputs, tail, gout, partials, slab_stride, grad_inputs, gin_ld, ws, p, grid, st);
    QIDDM_ADJ_CASE(11)
    QIDDM_ADJ_CASE(12)
    QIDDM_ADJ_CASE(13)
    QIDDM_ADJ_CASE(14)
    QIDDM_ADJ_CASE(15)
    QIDDM_ADJ_CASE(16)
#undef QIDDM_ADJ_CASE
    default: return fail(QIDDM_ERR_UNSUPPORTED, "wide CZ reverse sweep covers 11..16 qubits (got %d)", n);
  }
}

}  // namespace

bool wide_cz_adjoint_eligible(const qiddm_circuit_t* c) {
  static const bool force_tiled = std::getenv("QIDDM_WIDE_TILED") != nullptr;
  return !force_tiled && wide_cz_eligible(c) && c->n_rounds == 1 && (int64_t)c->n_blocks * c->sel_layers >= 2;
}

int launch_wide_cz_adjoint(int dtype, int n, const void* inputs, const void* tail, const void* gout, void* partials,
                           int64_t slab_stride, void* grad_inputs, int64_t gin_ld, void* ws, const qiddm::KScalars& p,
                           int64_t grid, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  return dtype == QIDDM_F32
             ? launch_adj_t<float>(n, inputs, tail, gout, partials, slab_stride, grad_inputs, gin_ld, ws, p, grid, st)
             : launch_adj_t<double>(n, inputs, tail, gout, partials, slab_stride, grad_inputs, gin_ld, ws, p, grid, st);
}

// resident workgroups (= slabs in use): up to four 4-wave workgroups per CU
int64_t wide_cz_grid(int64_t batch, int64_t slabs) {
  int64_t g = 1024;
  if (g > slabs) g = slabs;
  return batch < g ? batch : g;
}

int launch_wide_cz(int dtype, int n, const void* inputs, const void* tail, void* out, void* ws,
                   const qiddm::KScalars& p, int64_t slabs, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  return dtype == QIDDM_F32 ? launch_t<float>(n, inputs, tail, out, ws, p, slabs, st)
                            : launch_t<double>(n, inputs, tail, out, ws, p, slabs, st);
}

}  // namespace qiddm_capi
