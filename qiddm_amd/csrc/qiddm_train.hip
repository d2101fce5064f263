// qiddm_train.hip -- extern "C" entry points of the fused training step (include/qiddm_hip.h,
// "device-resident Diffusion step").  Validation, workspace carving, launch geometry; device code in
// qsim_train.h.
#include "capi_common.h"

#include <hip/hip_runtime.h>

#include <cstdlib>

#include "qsim_train.h"

namespace {

using qiddm_capi::check_circuit;
using qiddm_capi::fail;
using qiddm_capi::kMaxLds;

constexpr int64_t kMaxRowBlocks = 1024;

struct Geometry {
  int64_t rows, tiles, samples_per_chunk, n_chunks;
  int64_t n_rot_all;
  // workspace offsets (bytes)
  int64_t off_proj, off_ev, off_gxr, off_partials, off_loss, off_k, total;
};

Geometry geometry(const qiddm_circuit_t* c, int64_t batch, int32_t pixels, int32_t tau) {
  Geometry g;
  const int n = c->n_qubits;
  g.rows = batch * tau;
  g.tiles = (pixels + qiddm::kWave - 1) / qiddm::kWave;
  int64_t target = 1024 / g.tiles;
  if (target < 1) target = 1;
  g.samples_per_chunk = (batch + target - 1) / target;
  if (g.samples_per_chunk < 1) g.samples_per_chunk = 1;
  g.n_chunks = batch > 0 ? (batch + g.samples_per_chunk - 1) / g.samples_per_chunk : 0;
  g.n_rot_all = (int64_t)c->n_rounds * c->n_blocks * c->sel_layers * n;
  auto up = [](int64_t v) { return (v + 255) / 256 * 256; };
  int64_t o = 0;
  g.off_proj = o;     o += up((batch * (tau + 1) + n + 2) * 2 * n * 8);
  g.off_ev = o;       o += up(g.rows * n * 8);
  g.off_gxr = o;      o += up(g.rows * n * 8);
  g.off_partials = o; o += up(g.n_chunks * (2 * n + 1) * (int64_t)pixels * 8);
  g.off_loss = o;     o += up(g.n_chunks * g.tiles * 8);
  g.off_k = o;        o += up(kMaxRowBlocks * g.n_rot_all * 8 * (c->dtype == QIDDM_F32 ? 4 : 8));
  g.total = o;
  return g;
}

template <typename T, int N, bool QUANTUM, int WPB, bool FOLD>
int launch_rows(const qiddm_train_args_t* a, const Geometry& g, unsigned char* ws, const qiddm::TrainScalars& d,
                const qiddm::KScalars& p, size_t smem, int64_t blocks, hipStream_t st) {
  auto kern = qiddm::train_rows_kernel<T, N, QUANTUM, WPB, FOLD>;
  static qiddm_capi::DeviceFlags big_lds_enabled;
  if (smem > 48 * 1024 && !big_lds_enabled.get()) {
    const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    if (ea != hipSuccess)
      return fail(QIDDM_ERR_LAUNCH, "hipFuncSetAttribute(LDS) failed: %s", hipGetErrorString(ea));
    big_lds_enabled.set();
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(WPB * qiddm::kWave), smem, st,
                     reinterpret_cast<const double*>(ws + g.off_proj), a->b_down, a->angles,
                     reinterpret_cast<double*>(ws + g.off_ev), reinterpret_cast<double*>(ws + g.off_gxr),
                     reinterpret_cast<T*>(ws + g.off_k), a->batch, d, p);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess)
    return fail(QIDDM_ERR_LAUNCH, "train_rows_kernel<n=%d> launch failed: %s", N, hipGetErrorString(e));
  return QIDDM_OK;
}

template <typename T, int N>
int run_step(const qiddm_circuit_t* c, const qiddm_train_args_t* a, const Geometry& g, unsigned char* ws,
             hipStream_t st) {
  using L = qiddm::Layout<N>;
  qiddm::KScalars p{};
  p.batch = g.rows;
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
  qiddm::TrainScalars d{};
  d.x_ld = a->x_ld;
  d.noise_ld = a->noise_ld;
  d.rows = g.rows;
  d.pixels = a->pixels;
  d.T = a->tau;
  d.goal = a->goal;
  d.train_quantum = a->train_quantum ? 1 : 0;
  d.samples_per_chunk = (int32_t)g.samples_per_chunk;
  d.n_chunks = (int32_t)g.n_chunks;
  d.want_recon = a->recon != nullptr;
  d.want_elem = a->elem_loss != nullptr;
  d.grad_scale = (a->goal == 0 ? 2.0 : 0.2) / ((double)g.rows * (double)a->pixels);
  const bool q = d.train_quantum != 0;
  static const bool no_fold = std::getenv("QIDDM_NO_FOLD") != nullptr;  // kernel experiments: general reverse sweep
  d.fold = (!no_fold && qiddm::can_fold(c->imprimitive, c->encoding) && N >= 2 &&
            N <= qiddm::kFoldedAdjointMaxQubits) ? 1 : 0;   // (the forward-only step folds too)
  d.layers_per_round = c->n_blocks * c->sel_layers;
  p.fold = d.fold;
  hipError_t e;

  // ---- 1a. projections (only the reverse sweep and linear_down consume them) ---------------------------------
  {
    const int64_t units = a->batch * ((a->tau + qiddm::kProjLevels) / qiddm::kProjLevels) + (q ? N + 2 : 0);
    hipLaunchKernelGGL(qiddm::train_project_kernel<N>, dim3((unsigned)units), dim3(qiddm::kProjWaves * qiddm::kWave), 0,
                       st, a->x, a->noise, a->rng_state, a->schedule, a->w_down, a->w_up, a->b_up,
                       reinterpret_cast<double*>(ws + g.off_proj), a->batch, d);
    e = hipGetLastError();
    if (e != hipSuccess)
      return fail(QIDDM_ERR_LAUNCH, "train_project_kernel launch failed: %s", hipGetErrorString(e));
  }

  // ---- 1b. rows ---------------------------------------------------------------------------------------------
  const bool cnot = c->imprimitive == QIDDM_IMP_CNOT;
  const int64_t groups = (g.rows + L::SPW - 1) / L::SPW;
  constexpr int WPB = 4;
  const size_t smem = qiddm::train_lds_bytes<T, N>(g.n_rot_all, c->n_rounds, cnot, WPB, q);
  if (smem > kMaxLds)
    return fail(QIDDM_ERR_UNSUPPORTED, "circuit with %lld Rot gates needs %zu B of LDS for the training step",
                (long long)g.n_rot_all, smem);
  int64_t k_blocks = (groups + WPB - 1) / WPB;
  if (k_blocks > kMaxRowBlocks) k_blocks = kMaxRowBlocks;
  int rc;
  if (!q && d.fold) {
    if constexpr (N >= 2 && N <= qiddm::kFoldedAdjointMaxQubits)
      rc = launch_rows<T, N, false, WPB, true>(a, g, ws, d, p, smem, k_blocks, st);
    else
      rc = fail(QIDDM_ERR_UNSUPPORTED, "folded forward is not instantiated for n_qubits=%d", N);
  } else if (!q) {
    rc = launch_rows<T, N, false, WPB, false>(a, g, ws, d, p, smem, k_blocks, st);
  } else if (d.fold) {
    if constexpr (N >= 2 && N <= qiddm::kFoldedAdjointMaxQubits)
      rc = launch_rows<T, N, true, WPB, true>(a, g, ws, d, p, smem, k_blocks, st);
    else
      rc = fail(QIDDM_ERR_UNSUPPORTED, "folded training sweep is not instantiated for n_qubits=%d", N);
  } else {
    rc = launch_rows<T, N, true, WPB, false>(a, g, ws, d, p, smem, k_blocks, st);
  }
  if (rc != QIDDM_OK) return rc;

  // ---- 2. weight-gradient partials ----------------------------------------------------------------------
  hipLaunchKernelGGL(qiddm::train_weight_grads_kernel<N>, dim3((unsigned)g.tiles, (unsigned)g.n_chunks),
                     dim3(qiddm::kGradWaves * qiddm::kWave), 0, st, a->x, a->noise, a->schedule, a->w_up, a->b_up,
                     reinterpret_cast<const double*>(ws + g.off_ev), reinterpret_cast<const double*>(ws + g.off_gxr),
                     reinterpret_cast<double*>(ws + g.off_partials), reinterpret_cast<double*>(ws + g.off_loss),
                     a->recon, a->elem_loss, a->batch, d);
  e = hipGetLastError();
  if (e != hipSuccess)
    return fail(QIDDM_ERR_LAUNCH, "train_weight_grads_kernel launch failed: %s", hipGetErrorString(e));

  // ---- 3. finalize -----------------------------------------------------------------------------------------
  const int rows_out = q ? 2 * N + 1 : N + 1;
  const int wblocks = (int)(((int64_t)rows_out * a->pixels + qiddm::kWave - 1) / qiddm::kWave);
  const int64_t total_blocks = wblocks + 1 + N + (q ? g.n_rot_all : 0);
  hipLaunchKernelGGL(qiddm::train_finalize_kernel<T>, dim3((unsigned)total_blocks), dim3(qiddm::kWave), 0, st,
                     reinterpret_cast<const double*>(ws + g.off_partials),
                     reinterpret_cast<const double*>(ws + g.off_loss), g.n_chunks * g.tiles,
                     reinterpret_cast<const double*>(ws + g.off_gxr), reinterpret_cast<const T*>(ws + g.off_k),
                     k_blocks, a->angles, N, g.n_rot_all, wblocks, a->loss, a->g_w_down, a->g_b_down, a->g_angles,
                     a->g_w_up, a->g_b_up, a->rng_state, d);
  e = hipGetLastError();
  if (e != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "train_finalize_kernel launch failed: %s", hipGetErrorString(e));
  return QIDDM_OK;
}

template <typename T>
int dispatch(const qiddm_circuit_t* c, const qiddm_train_args_t* a, const Geometry& g, unsigned char* ws,
             hipStream_t st) {
  switch (c->n_qubits) {
    case 1: return run_step<T, 1>(c, a, g, ws, st);
    case 2: return run_step<T, 2>(c, a, g, ws, st);
    case 3: return run_step<T, 3>(c, a, g, ws, st);
    case 4: return run_step<T, 4>(c, a, g, ws, st);
    case 5: return run_step<T, 5>(c, a, g, ws, st);
    case 6: return run_step<T, 6>(c, a, g, ws, st);
    case 7: return run_step<T, 7>(c, a, g, ws, st);
    case 8: return run_step<T, 8>(c, a, g, ws, st);
    case 9: return run_step<T, 9>(c, a, g, ws, st);
    case 10: return run_step<T, 10>(c, a, g, ws, st);
    default: return fail(QIDDM_ERR_UNSUPPORTED, "the fused training step needs n_qubits <= 10 (got %d)", c->n_qubits);
  }
}

int check_args(const qiddm_circuit_t* c, int64_t batch, int32_t pixels, int32_t tau) {
  int rc = check_circuit(c);
  if (rc != QIDDM_OK) return rc;
  if (c->measure != QIDDM_MEAS_EXPZ || c->encoding == QIDDM_ENC_AMPLITUDE)
    return fail(QIDDM_ERR_UNSUPPORTED,
                "the fused training step covers the linear -> angle-encoded circuit -> <Z> -> linear family");
  if (c->n_qubits > QIDDM_MAX_QUBITS_FUSED)
    return fail(QIDDM_ERR_UNSUPPORTED, "the fused training step needs n_qubits <= %d (got %d)",
                QIDDM_MAX_QUBITS_FUSED, c->n_qubits);
  if (batch < 1 || pixels < 1 || tau < 1)
    return fail(QIDDM_ERR_INVALID, "batch/pixels/tau must be >= 1 (got %lld/%d/%d)", (long long)batch, pixels, tau);
  if (batch * (int64_t)tau > ((int64_t)1 << 31))
    return fail(QIDDM_ERR_UNSUPPORTED, "batch * tau too large");
  return QIDDM_OK;
}

}  // namespace

extern "C" {

int64_t qiddm_train_workspace_bytes(const qiddm_circuit_t* circ, int64_t batch, int32_t pixels, int32_t tau) {
  const int rc = check_args(circ, batch, pixels, tau);
  if (rc != QIDDM_OK) return rc;
  return geometry(circ, batch, pixels, tau).total;
}

int qiddm_train_step(const qiddm_circuit_t* circ, const qiddm_train_args_t* a, void* workspace,
                     int64_t workspace_bytes, void* stream) {
  if (!a) return fail(QIDDM_ERR_INVALID, "args is NULL");
  int rc = check_args(circ, a->batch, a->pixels, a->tau);
  if (rc != QIDDM_OK) return rc;
  if (a->goal != 0 && a->goal != 1) return fail(QIDDM_ERR_INVALID, "goal must be 0 (data) or 1 (noise), got %d", a->goal);
  if (!a->x || !a->noise || !a->schedule || !a->w_down || !a->angles || !a->w_up)
    return fail(QIDDM_ERR_INVALID, "x/noise/schedule/w_down/angles/w_up must not be NULL");
  if (!a->loss || !a->g_w_up || !a->g_b_up) return fail(QIDDM_ERR_INVALID, "loss/g_w_up/g_b_up must not be NULL");
  if (a->train_quantum && (!a->g_w_down || !a->g_b_down || !a->g_angles))
    return fail(QIDDM_ERR_INVALID, "train_quantum needs g_w_down/g_b_down/g_angles");
  if (a->x_ld < a->pixels || a->noise_ld < a->pixels)
    return fail(QIDDM_ERR_INVALID, "x_ld/noise_ld smaller than pixels");
  const Geometry g = geometry(circ, a->batch, a->pixels, a->tau);
  if (!workspace || workspace_bytes < g.total)
    return fail(QIDDM_ERR_INVALID, "workspace of %lld B needed (qiddm_train_workspace_bytes), got %lld",
                (long long)g.total, (long long)workspace_bytes);
  unsigned char* ws = static_cast<unsigned char*>(workspace);
  hipStream_t st = static_cast<hipStream_t>(stream);
  return circ->dtype == QIDDM_F32 ? dispatch<float>(circ, a, g, ws, st) : dispatch<double>(circ, a, g, ws, st);
}

int qiddm_adam_step(const qiddm_adam_tensor_t* tensors, int32_t n_tensors, double lr, double beta1, double beta2,
                    double eps, double weight_decay, uint32_t* sync, void* stream) {
  if (n_tensors < 0) return fail(QIDDM_ERR_INVALID, "n_tensors < 0");
  if (n_tensors > 0 && !tensors) return fail(QIDDM_ERR_INVALID, "tensors is NULL");
  if (!sync) return fail(QIDDM_ERR_INVALID, "sync must not be NULL");
  if (!(lr >= 0.0) || !(eps >= 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) ||
      !(weight_decay >= 0.0))
    return fail(QIDDM_ERR_INVALID, "Invalid Adam hyper-parameter (lr %g, betas %g/%g, eps %g, weight_decay %g)", lr,
                beta1, beta2, eps, weight_decay);
  for (int i = 0; i < n_tensors; ++i) {
    const qiddm_adam_tensor_t& t = tensors[i];
    if (t.numel < 1 || (t.dtype != QIDDM_F32 && t.dtype != QIDDM_F64))
      return fail(QIDDM_ERR_INVALID, "tensor %d: bad numel/dtype", i);
    if (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq || !t.step)
      return fail(QIDDM_ERR_INVALID, "tensor %d: NULL pointer", i);
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int done = 0; done < n_tensors;) {  // batches of kAdamMaxTensors, each a self-contained launch
    qiddm::AdamBatch a{};
    int64_t blocks = 0;
    int k = 0;
    for (; k < qiddm::kAdamMaxTensors && done + k < n_tensors; ++k) {
      const qiddm_adam_tensor_t& t = tensors[done + k];
      a.param[k] = t.param;
      a.grad[k] = t.grad;
      a.exp_avg[k] = t.exp_avg;
      a.exp_avg_sq[k] = t.exp_avg_sq;
      a.step[k] = t.step;
      a.numel[k] = t.numel;
      a.is_f64[k] = t.dtype == QIDDM_F64;
      blocks += (t.numel + 255) / 256;
      a.block_end[k] = blocks;
    }
    done += k;
    a.n_tensors = k;
    a.lr = lr;
    a.beta1 = beta1;
    a.beta2 = beta2;
    a.eps = eps;
    a.weight_decay = weight_decay;
    if (blocks > 0x7fffffff) return fail(QIDDM_ERR_UNSUPPORTED, "too many elements for one Adam launch");
    hipLaunchKernelGGL(qiddm::adam_step_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, sync);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "adam_step_kernel launch failed: %s", hipGetErrorString(e));
  }
  return QIDDM_OK;
}

}  // extern "C"
