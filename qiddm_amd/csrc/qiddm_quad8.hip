// qiddm_quad8.hip -- C entry points of the lean sampling loop of the 8-qubit dense nets (qsim_quad8.h):
// qiddm_dense_sample_lean_tables_bytes / _prepare / _check / qiddm_dense_sample_lean (include/qiddm_hip.h).
#include "capi_common.h"

#include <hip/hip_runtime.h>

#include <cstring>

#include "qsim_quad8.h"

namespace {

using qiddm_capi::fail;
using qiddm_capi::kMaxLds;

// the family the lean kernel is written for: 8 wires, RZ data encoding, CZ rings, <Z> read-out
int lean_layers(const qiddm_circuit_t* c) {
  if (c->n_qubits != 8 || c->imprimitive != QIDDM_IMP_CZ || c->encoding != QIDDM_ENC_RZ || c->measure != QIDDM_MEAS_EXPZ)
    return fail(QIDDM_ERR_UNSUPPORTED, "lean sampling loop: 8 qubits, CZ rings, RZ encoding, <Z> read-out only");
  const int64_t layers = (int64_t)c->n_rounds * c->n_blocks * c->sel_layers;
  const size_t lds = c->dtype == QIDDM_F32 ? qiddm::Quad8Tables<float>::lds_bytes((int)layers, c->n_rounds)
                                           : qiddm::Quad8Tables<double>::lds_bytes((int)layers, c->n_rounds);
  if (layers > 128 || lds > kMaxLds)
    return fail(QIDDM_ERR_UNSUPPORTED, "lean sampling loop: %lld layers need %zu B of LDS (limit %zu, 128 layers)",
                (long long)layers, lds, kMaxLds);
  return (int)layers;
}

qiddm::KScalars params_of(const qiddm_circuit_t* c) {
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
  return p;
}

template <typename T, int PPT, bool REUP, int LPR>
int launch_lean(const double* x, const double* wd, const double* bd, const double* wu, const double* bu, double* y,
                const void* tables, const qiddm::QuadScalars& d, const qiddm::KScalars& p, int layers, hipStream_t st) {
  const size_t smem = qiddm::Quad8Tables<T>::lds_bytes(layers, p.n_rounds);
  auto kern = qiddm::dense_quad8_kernel<T, PPT, REUP, LPR>;
  static qiddm_capi::DeviceFlags big_lds_enabled;
  if (smem > 48 * 1024 && !big_lds_enabled.get()) {
    const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    if (ea != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "hipFuncSetAttribute(LDS) failed: %s", hipGetErrorString(ea));
    big_lds_enabled.set();
  }
  const unsigned blocks = (unsigned)(p.batch < 2048 ? p.batch : 2048);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), smem, st, x, wd, bd, wu, bu, y,
                     static_cast<const unsigned char*>(tables), d, p);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "dense_quad8_kernel launch failed: %s", hipGetErrorString(e));
  return QIDDM_OK;
}

// the instantiation for this circuit: re-upload or not; layers per round compiled in for the two shapes the reference's
// drivers use at 8 qubits (14: QNN_noise(784, 8, 14), src/mnist_exm.py:48; 12: the (8, 6, 2) LL / PL nets,
// src/fashion_exm.py:45), a runtime count otherwise
template <typename T>
int dispatch_lean(const double* x, const double* wd, const double* bd, const double* wu, const double* bu, double* y,
                  const void* tables, const qiddm::QuadScalars& d, const qiddm::KScalars& p, int layers, hipStream_t st) {
  const bool reup = p.n_blocks > 1;
  const int lpr = p.n_blocks * p.sel_layers;
  if (d.in_features > 1024)
    return reup ? launch_lean<T, 8, true, 0>(x, wd, bd, wu, bu, y, tables, d, p, layers, st)
                : launch_lean<T, 8, false, 0>(x, wd, bd, wu, bu, y, tables, d, p, layers, st);
  if (reup) {
    if (lpr == 12) return launch_lean<T, 4, true, 12>(x, wd, bd, wu, bu, y, tables, d, p, layers, st);
    return launch_lean<T, 4, true, 0>(x, wd, bd, wu, bu, y, tables, d, p, layers, st);
  }
  if (lpr == 14) return launch_lean<T, 4, false, 14>(x, wd, bd, wu, bu, y, tables, d, p, layers, st);
  return launch_lean<T, 4, false, 0>(x, wd, bd, wu, bu, y, tables, d, p, layers, st);
}

}  // namespace

extern "C" {

int64_t qiddm_dense_sample_lean_tables_bytes(const qiddm_circuit_t* c) {
  int rc = qiddm_capi::check_circuit(c);
  if (rc != QIDDM_OK) return rc;
  const int layers = lean_layers(c);
  if (layers < 0) return layers;
  return (int64_t)(c->dtype == QIDDM_F32 ? qiddm::Quad8Tables<float>::bytes(layers, c->n_rounds)
                                         : qiddm::Quad8Tables<double>::bytes(layers, c->n_rounds));
}

int qiddm_dense_sample_lean_prepare(const qiddm_circuit_t* c, const double* angles, const double* w_down,
                                    const double* b_down, const double* w_up, const double* b_up, int64_t features,
                                    void* tables, void* stream) {
  const int64_t need = qiddm_dense_sample_lean_tables_bytes(c);
  if (need < 0) return (int)need;
  if (!angles || !tables || !w_down || !w_up) return fail(QIDDM_ERR_INVALID, "angles/tables/w_down/w_up is NULL");
  if (features < 1 || features > 2048) return fail(QIDDM_ERR_INVALID, "features=%lld outside 1..2048", (long long)features);
  const qiddm::KScalars p = params_of(c);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (c->dtype == QIDDM_F32)
    hipLaunchKernelGGL(qiddm::quad8_tables_kernel<float>, dim3(1), dim3(256), 0, st, angles, w_down, b_down, w_up, b_up,
                       (int)features, static_cast<unsigned char*>(tables), p);
  else
    hipLaunchKernelGGL(qiddm::quad8_tables_kernel<double>, dim3(1), dim3(256), 0, st, angles, w_down, b_down, w_up, b_up,
                       (int)features, static_cast<unsigned char*>(tables), p);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "quad8_tables_kernel launch failed: %s", hipGetErrorString(e));
  return QIDDM_OK;
}

int qiddm_dense_sample_lean_check(const qiddm_circuit_t* c, const void* tables, void* stream) {
  const int64_t need = qiddm_dense_sample_lean_tables_bytes(c);
  if (need < 0) return (int)need;
  if (!tables) return fail(QIDDM_ERR_INVALID, "tables is NULL");
  double tmax = 0.0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipError_t e = hipMemcpyAsync(&tmax, tables, sizeof(double), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "reading the tables' max |tan| failed: %s", hipGetErrorString(e));
  return (tmax == tmax && tmax <= qiddm::kQuad8MaxTan) ? 1 : 0;
}

int qiddm_dense_sample_lean(const qiddm_circuit_t* c, const double* x, int64_t batch, int64_t x_ld, int64_t features,
                            const double* w_down, const double* b_down, const double* w_up, const double* b_up,
                            int32_t n_steps, double* y, int64_t y_ld, int64_t y_step_stride, const void* tables,
                            void* stream) {
  int rc = qiddm_capi::check_circuit(c);
  if (rc != QIDDM_OK) return rc;
  const int layers = lean_layers(c);
  if (layers < 0) return layers;
  if (batch < 0 || n_steps < 0) return fail(QIDDM_ERR_INVALID, "negative batch / n_steps");
  if (features < 1 || features > 2048) return fail(QIDDM_ERR_UNSUPPORTED, "features=%lld outside 1..2048", (long long)features);
  if (batch == 0 || n_steps == 0) return QIDDM_OK;
  if (!x || !w_down || !w_up || !y || !tables) return fail(QIDDM_ERR_INVALID, "x/w_down/w_up/y/tables is NULL");
  if (x == y) return fail(QIDDM_ERR_INVALID, "y must not alias x");
  if (x_ld < features || y_ld < features || y_step_stride < batch * y_ld - (y_ld - features))
    return fail(QIDDM_ERR_INVALID, "strides smaller than the tensor extents");
  qiddm::KScalars p = params_of(c);
  p.batch = batch;
  qiddm::QuadScalars d;
  std::memset(&d, 0, sizeof(d));
  d.x_ld = x_ld;
  d.y_ld = y_ld;
  d.y_step_stride = y_step_stride;
  d.in_features = (int32_t)features;
  d.out_features = (int32_t)features;
  d.post_mode = 0;
  d.n_steps = n_steps;
  d.noise_factor = 1.0;
  d.stamps = qiddm_capi::stamp_buffer(8);
  hipStream_t st = static_cast<hipStream_t>(stream);
  return c->dtype == QIDDM_F32 ? dispatch_lean<float>(x, w_down, b_down, w_up, b_up, y, tables, d, p, layers, st)
                               : dispatch_lean<double>(x, w_down, b_down, w_up, b_up, y, tables, d, p, layers, st);
}

}  // extern "C"
