// qiddm_mixed.hip -- extern "C" entry points of the density-matrix executor (include/qiddm_hip.h,
// "hardware-noise study"); device code in qsim_mixed.h.
#include "capi_common.h"

#include <hip/hip_runtime.h>

#include "qsim_mixed.h"

namespace {

using qiddm_capi::fail;
using qiddm_capi::kMaxLds;

constexpr int kMixedMaxQubits = 8;
constexpr int64_t kMixedMaxBlocks = 256;
constexpr size_t kMixedLdsSlab = 128 * 1024;

struct MixedGeometry {
  int64_t blocks, slab_bytes, off_prog, off_slabs, total;
  bool in_lds;
};

int mixed_geometry(int32_t n, int32_t dtype, int64_t batch, int32_t n_ops, MixedGeometry* g) {
  if (n < 1 || n > kMixedMaxQubits)
    return fail(QIDDM_ERR_UNSUPPORTED, "density-matrix execution needs 1 <= n_qubits <= %d (got %d)", kMixedMaxQubits, n);
  if (dtype != QIDDM_F32 && dtype != QIDDM_F64) return fail(QIDDM_ERR_INVALID, "unknown dtype %d", dtype);
  if (batch < 0 || n_ops < 0) return fail(QIDDM_ERR_INVALID, "negative batch / n_ops");
  g->slab_bytes = ((int64_t)1 << (2 * n)) * (dtype == QIDDM_F32 ? 8 : 16);
  g->in_lds = (size_t)g->slab_bytes <= kMixedLdsSlab;
  g->blocks = batch < kMixedMaxBlocks ? batch : kMixedMaxBlocks;
  if (g->blocks < 1) g->blocks = 1;
  g->off_prog = 0;
  g->off_slabs = ((int64_t)n_ops * (int64_t)sizeof(qiddm::MixedOp) + 255) / 256 * 256;
  g->total = g->off_slabs + (g->in_lds ? 0 : g->blocks * g->slab_bytes);
  return QIDDM_OK;
}

static_assert(sizeof(qiddm::MixedOp) == sizeof(qiddm_mixed_op_t), "program layout");

}  // namespace

extern "C" {

int64_t qiddm_mixed_workspace_bytes(int32_t n_qubits, int32_t dtype, int64_t batch, int32_t n_ops) {
  MixedGeometry g;
  const int rc = mixed_geometry(n_qubits, dtype, batch, n_ops, &g);
  if (rc != QIDDM_OK) return rc;
  return g.total;
}

int qiddm_mixed_forward(int32_t n_qubits, int32_t dtype, const qiddm_mixed_op_t* program, int32_t n_ops,
                        const double* angle_rows, int64_t rows_ld, int32_t n_rows, const double* features,
                        int64_t feat_ld, int32_t n_features, double enc_offset, double pad_with, const double* gates,
                        int32_t n_gates, int32_t measure, int64_t batch, double* out, int64_t out_ld, void* workspace,
                        int64_t workspace_bytes, void* stream) {
  MixedGeometry g;
  int rc = mixed_geometry(n_qubits, dtype, batch, n_ops, &g);
  if (rc != QIDDM_OK) return rc;
  if (measure != QIDDM_MEAS_PROBS && measure != QIDDM_MEAS_EXPZ) return fail(QIDDM_ERR_INVALID, "unknown measure %d", measure);
  if (batch == 0) return QIDDM_OK;
  if (!program || n_ops < 1) return fail(QIDDM_ERR_INVALID, "empty program");
  if (!out) return fail(QIDDM_ERR_INVALID, "out is NULL");
  if (n_rows < 0 || n_gates < 0) return fail(QIDDM_ERR_INVALID, "negative n_rows / n_gates");
  if (n_rows > 0 && (!angle_rows || rows_ld < batch)) return fail(QIDDM_ERR_INVALID, "angle_rows missing or rows_ld < batch");
  if (n_gates > 0 && !gates) return fail(QIDDM_ERR_INVALID, "gates is NULL");
  if (program[0].kind != qiddm::kMixZero && program[0].kind != qiddm::kMixAmpEmbed)
    return fail(QIDDM_ERR_INVALID, "the program must start by preparing the state");
  const int64_t d = (int64_t)1 << n_qubits;
  for (int i = 0; i < n_ops; ++i) {
    const qiddm_mixed_op_t& op = program[i];
    if (op.kind < qiddm::kMixZero || op.kind > qiddm::kMixDepol) return fail(QIDDM_ERR_INVALID, "op %d: unknown kind %d", i, op.kind);
    if (op.kind == qiddm::kMixZero) continue;
    if (op.kind == qiddm::kMixAmpEmbed) {
      if (!features || n_features < 1 || n_features > d || feat_ld < n_features)
        return fail(QIDDM_ERR_INVALID, "Features must be of length %lld or smaller; got length %d.", (long long)d, n_features);
      continue;
    }
    if (op.wire < 0 || op.wire >= n_qubits) return fail(QIDDM_ERR_INVALID, "op %d: wire %d out of range", i, op.wire);
    switch (op.kind) {
      case qiddm::kMixPhase:
      case qiddm::kMixRY:
        if (op.a >= n_rows) return fail(QIDDM_ERR_INVALID, "op %d: angle row %d out of range", i, op.a);
        break;
      case qiddm::kMixGate:
        if (op.a < 0 || op.a >= n_gates) return fail(QIDDM_ERR_INVALID, "op %d: gate %d out of range", i, op.a);
        break;
      case qiddm::kMixCZ:
      case qiddm::kMixCNOT:
        if (op.a < 0 || op.a >= n_qubits || op.a == op.wire)
          return fail(QIDDM_ERR_INVALID, "op %d: bad target wire %d", i, op.a);
        break;
      default:
        if (!(op.p >= 0.0 && op.p <= 1.0))
          return fail(QIDDM_ERR_INVALID, "op %d: channel probability %g outside [0, 1]", i, op.p);
    }
  }
  if (!workspace || workspace_bytes < g.total)
    return fail(QIDDM_ERR_INVALID, "workspace of %lld B needed (qiddm_mixed_workspace_bytes), got %lld",
                (long long)g.total, (long long)workspace_bytes);
  hipStream_t st = static_cast<hipStream_t>(stream);
  unsigned char* ws = static_cast<unsigned char*>(workspace);
  hipError_t e = hipMemcpyAsync(ws + g.off_prog, program, (size_t)n_ops * sizeof(qiddm_mixed_op_t), hipMemcpyHostToDevice, st);
  if (e != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "program upload failed: %s", hipGetErrorString(e));
  qiddm::MixedScalars m{};
  m.n = n_qubits;
  m.n_ops = n_ops;
  m.measure = measure;
  m.n_features = n_features;
  m.batch = batch;
  m.rows_ld = rows_ld;
  m.feat_ld = feat_ld;
  m.out_ld = out_ld;
  m.enc_offset = enc_offset;
  m.pad_with = pad_with;
  m.slab_in_lds = g.in_lds ? 1 : 0;
  const size_t smem = g.in_lds ? (size_t)g.slab_bytes : 0;
  const qiddm::MixedOp* prog = reinterpret_cast<const qiddm::MixedOp*>(ws + g.off_prog);
  if (dtype == QIDDM_F32) {
    auto kern = qiddm::mixed_kernel<float>;
    static qiddm_capi::DeviceFlags big;
    if (smem > 48 * 1024 && !big.get()) {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kMaxLds - 4096));  // the kernel also has 2 KiB of static LDS
      if (e != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "hipFuncSetAttribute(LDS) failed: %s", hipGetErrorString(e));
      big.set();
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)g.blocks), dim3(256), smem, st, prog, angle_rows, features, gates, out,
                       reinterpret_cast<qiddm::V2<float>*>(ws + g.off_slabs), m);
  } else {
    auto kern = qiddm::mixed_kernel<double>;
    static qiddm_capi::DeviceFlags big;
    if (smem > 48 * 1024 && !big.get()) {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kMaxLds - 4096));  // the kernel also has 2 KiB of static LDS
      if (e != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "hipFuncSetAttribute(LDS) failed: %s", hipGetErrorString(e));
      big.set();
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)g.blocks), dim3(256), smem, st, prog, angle_rows, features, gates, out,
                       reinterpret_cast<qiddm::V2<double>*>(ws + g.off_slabs), m);
  }
  e = hipGetLastError();
  if (e != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "mixed_kernel launch failed: %s", hipGetErrorString(e));
  return QIDDM_OK;
}

}  // extern "C"
