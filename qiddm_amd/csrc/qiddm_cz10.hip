// qiddm_cz10.hip -- launch side of the register-resident reverse sweep of 10-qubit CZ circuits (qsim_cz10_adjoint.h).
// Reached from qiddm_backward_adjoint (qiddm_capi.hip).
#include "capi_common.h"

#include <hip/hip_runtime.h>

#include <cstdlib>

#include "qsim_cz10_adjoint.h"

namespace qiddm_capi {

bool cz10_adjoint_eligible(const qiddm_circuit_t* c) {
  static const bool off = std::getenv("QIDDM_NO_LEAN") != nullptr;   // kernel experiments: A/B on the same box
  return !off && c->n_qubits == 10 && c->imprimitive == QIDDM_IMP_CZ &&
         (c->encoding == QIDDM_ENC_NONE || c->encoding == QIDDM_ENC_RZ) && c->n_rounds == 1 &&
         (int64_t)c->n_blocks * c->sel_layers >= 2;
}

namespace {
template <typename T>
int launch_t(const void* inputs, const void* tail, const void* gout, void* partials, int64_t slab_stride,
             void* grad_inputs, int64_t gin_ld, const qiddm::KScalars& p, int64_t grid, hipStream_t st) {
  const int64_t layers = (int64_t)p.n_blocks * p.sel_layers;
  const size_t smem = qiddm::Cz10AdjSmem<T>::bytes(layers);
  if (smem > kMaxLds)
    return fail(QIDDM_ERR_UNSUPPORTED, "circuit with %lld layers needs %zu B of LDS for the reverse sweep (limit %zu)",
                (long long)layers, smem, kMaxLds);
  auto kern = qiddm::cz10_adjoint_kernel<T>;
  static DeviceFlags big_lds_enabled;
  if (smem > 48 * 1024 && !big_lds_enabled.get()) {
    const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    if (ea != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "hipFuncSetAttribute(LDS) failed: %s", hipGetErrorString(ea));
    big_lds_enabled.set();
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(qiddm::kCz10Waves * qiddm::kWave), smem, st,
                     static_cast<const T*>(inputs), static_cast<const T*>(tail), static_cast<const T*>(gout),
                     static_cast<T*>(partials), slab_stride, static_cast<T*>(grad_inputs), gin_ld, p);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "cz10_adjoint_kernel launch failed: %s", hipGetErrorString(e));
  return QIDDM_OK;
}
}  // namespace

int launch_cz10_adjoint(int dtype, const void* inputs, const void* tail, const void* gout, void* partials,
                        int64_t slab_stride, void* grad_inputs, int64_t gin_ld, const qiddm::KScalars& p, int64_t grid,
                        void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  return dtype == QIDDM_F32 ? launch_t<float>(inputs, tail, gout, partials, slab_stride, grad_inputs, gin_ld, p, grid, st)
                            : launch_t<double>(inputs, tail, gout, partials, slab_stride, grad_inputs, gin_ld, p, grid, st);
}

}  // namespace qiddm_capi
