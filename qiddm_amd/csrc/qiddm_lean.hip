// qiddm_lean.hip -- C entry points of the lean sampling loop of the 8-qubit dense nets (qsim_lean.h):
// qiddm_dense_sample_lean_tables_bytes / _prepare / _check / qiddm_dense_sample_lean (include/qiddm_hip.h).
#include "capi_common.h"

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "qsim_lean.h"

namespace {

using qiddm_capi::fail;
using qiddm_capi::kMaxLds;

template <typename T>
size_t lean_lds(int n, int layers, int rounds) {
  return n == 8 ? qiddm::LeanTables<T, 8>::lds_bytes(layers, rounds) : qiddm::LeanTables<T, 6>::lds_bytes(layers, rounds);
}
template <typename T>
size_t lean_bytes(int n, int layers, int rounds) {
  return n == 8 ? qiddm::LeanTables<T, 8>::bytes(layers, rounds) : qiddm::LeanTables<T, 6>::bytes(layers, rounds);
}

// the family the lean kernel is written for: 8 or 6 wires, RZ data encoding, CZ rings, <Z> read-out
int lean_layers(const qiddm_circuit_t* c) {
  if ((c->n_qubits != 8 && c->n_qubits != 6) || c->imprimitive != QIDDM_IMP_CZ || c->encoding != QIDDM_ENC_RZ ||
      c->measure != QIDDM_MEAS_EXPZ)
    return fail(QIDDM_ERR_UNSUPPORTED, "lean sampling loop: 6 or 8 qubits, CZ rings, RZ encoding, <Z> read-out only");
  const int64_t layers = (int64_t)c->n_rounds * c->n_blocks * c->sel_layers;
  if (layers > 128)
    return fail(QIDDM_ERR_UNSUPPORTED, "lean sampling loop: %lld layers (limit 128)", (long long)layers);
  const size_t lds = c->dtype == QIDDM_F32 ? lean_lds<float>(c->n_qubits, (int)layers, c->n_rounds)
                                           : lean_lds<double>(c->n_qubits, (int)layers, c->n_rounds);
  if (lds > kMaxLds)
    return fail(QIDDM_ERR_UNSUPPORTED, "lean sampling loop: %lld layers need %zu B of LDS (limit %zu)",
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

template <typename T, int N, int PPT, bool REUP, int LPR, bool POST>
int launch_lean(const double* x, const double* wd, const double* bd, const double* wu, const double* bu, double* y,
                const void* tables, const qiddm::QuadScalars& d, const qiddm::KScalars& p, int layers, hipStream_t st) {
  const size_t smem = qiddm::LeanTables<T, N>::lds_bytes(layers, p.n_rounds);
  auto kern = qiddm::dense_lean_kernel<T, N, PPT, REUP, LPR, POST>;
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
  if (e != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "dense_lean_kernel launch failed: %s", hipGetErrorString(e));
  return QIDDM_OK;
}

// the instantiation for this circuit: re-upload or not; layers per round compiled in for the shapes the reference's
// drivers use (src/mnist_exm.py:46-48, src/fashion_exm.py:45) -- 8 qubits: 14 (QNN_noise(784, 8, 14)) and 12 (the
// (8, 6, 2) LL / PL nets); 6 qubits: 28 (QIDDM_LL_noise(784, 6, 14, 2), the MNIST default) and 14 -- a runtime count
// otherwise
template <typename T, int N, bool POST>
int dispatch_lean_n(const double* x, const double* wd, const double* bd, const double* wu, const double* bu, double* y,
                    const void* tables, const qiddm::QuadScalars& d, const qiddm::KScalars& p, int layers, hipStream_t st) {
  const bool reup = p.n_blocks > 1;
  const int lpr = p.n_blocks * p.sel_layers;
  constexpr int kReupLpr = N == 8 ? 12 : 28;
  if (d.in_features > 1024)
    return reup ? launch_lean<T, N, 8, true, 0, POST>(x, wd, bd, wu, bu, y, tables, d, p, layers, st)
                : launch_lean<T, N, 8, false, 0, POST>(x, wd, bd, wu, bu, y, tables, d, p, layers, st);
  if (reup) {
    if (lpr == kReupLpr && p.sel_layers == 2)
      return launch_lean<T, N, 4, true, kReupLpr, POST>(x, wd, bd, wu, bu, y, tables, d, p, layers, st);
    return launch_lean<T, N, 4, true, 0, POST>(x, wd, bd, wu, bu, y, tables, d, p, layers, st);
  }
  if (lpr == 14) return launch_lean<T, N, 4, false, 14, POST>(x, wd, bd, wu, bu, y, tables, d, p, layers, st);
  return launch_lean<T, N, 4, false, 0, POST>(x, wd, bd, wu, bu, y, tables, d, p, layers, st);
}
template <typename T>
int dispatch_lean(int n, const double* x, const double* wd, const double* bd, const double* wu, const double* bu, double* y,
                  const void* tables, const qiddm::QuadScalars& d, const qiddm::KScalars& p, int layers, hipStream_t st) {
  if (d.post_mode == 1)
    return n == 8 ? dispatch_lean_n<T, 8, true>(x, wd, bd, wu, bu, y, tables, d, p, layers, st)
                  : dispatch_lean_n<T, 6, true>(x, wd, bd, wu, bu, y, tables, d, p, layers, st);
  return n == 8 ? dispatch_lean_n<T, 8, false>(x, wd, bd, wu, bu, y, tables, d, p, layers, st)
                : dispatch_lean_n<T, 6, false>(x, wd, bd, wu, bu, y, tables, d, p, layers, st);
}

}  // namespace

extern "C" {

int64_t qiddm_dense_sample_lean_tables_bytes(const qiddm_circuit_t* c) {
  int rc = qiddm_capi::check_circuit(c);
  if (rc != QIDDM_OK) return rc;
  const int layers = lean_layers(c);
  if (layers < 0) return layers;
  return (int64_t)(c->dtype == QIDDM_F32 ? lean_bytes<float>(c->n_qubits, layers, c->n_rounds)
                                         : lean_bytes<double>(c->n_qubits, layers, c->n_rounds));
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
  unsigned char* tb = static_cast<unsigned char*>(tables);
  const int f = (int)features;
  if (c->n_qubits == 8) {
    if (c->dtype == QIDDM_F32)
      hipLaunchKernelGGL((qiddm::lean_tables_kernel<float, 8>), dim3(1), dim3(256), 0, st, angles, w_down, b_down, w_up, b_up, f, tb, p);
    else
      hipLaunchKernelGGL((qiddm::lean_tables_kernel<double, 8>), dim3(1), dim3(256), 0, st, angles, w_down, b_down, w_up, b_up, f, tb, p);
  } else {
    if (c->dtype == QIDDM_F32)
      hipLaunchKernelGGL((qiddm::lean_tables_kernel<float, 6>), dim3(1), dim3(256), 0, st, angles, w_down, b_down, w_up, b_up, f, tb, p);
    else
      hipLaunchKernelGGL((qiddm::lean_tables_kernel<double, 6>), dim3(1), dim3(256), 0, st, angles, w_down, b_down, w_up, b_up, f, tb, p);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "lean_tables_kernel launch failed: %s", hipGetErrorString(e));
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
  return (tmax == tmax && tmax <= qiddm::kLeanMaxTan) ? 1 : 0;
}

int qiddm_dense_sample_lean(const qiddm_circuit_t* c, const double* x, int64_t batch, int64_t x_ld, int64_t features,
                            const double* w_down, const double* b_down, const double* w_up, const double* b_up,
                            int32_t post_mode, double noise_factor, int32_t n_steps, double* y, int64_t y_ld,
                            int64_t y_step_stride, const void* tables, void* stream) {
  int rc = qiddm_capi::check_circuit(c);
  if (rc != QIDDM_OK) return rc;
  const int layers = lean_layers(c);
  if (layers < 0) return layers;
  if (batch < 0 || n_steps < 0) return fail(QIDDM_ERR_INVALID, "negative batch / n_steps");
  if (post_mode != 0 && post_mode != 1) return fail(QIDDM_ERR_INVALID, "post_mode must be 0 or 1");
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
  d.post_mode = post_mode;
  d.n_steps = n_steps;
  d.noise_factor = noise_factor;
  d.stamps = qiddm_capi::stamp_buffer(8);
  hipStream_t st = static_cast<hipStream_t>(stream);
  return c->dtype == QIDDM_F32
             ? dispatch_lean<float>(c->n_qubits, x, w_down, b_down, w_up, b_up, y, tables, d, p, layers, st)
             : dispatch_lean<double>(c->n_qubits, x, w_down, b_down, w_up, b_up, y, tables, d, p, layers, st);
}

}  // extern "C"
