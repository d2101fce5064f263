// qiddm_qconv.hip -- extern "C" entry points of the eval-mode quantum convolution (circuit unitary + MFMA GEMM);
// device code in qsim_unitary.h.
#include "capi_common.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "qsim_qconv_dx.h"
#include "qsim_qconv_fwd.h"
#include "qsim_qconv_train.h"
#include "qsim_qconv_train_mfma.h"
#include "qsim_unitary.h"

namespace {

using qiddm_capi::check_circuit;
using qiddm_capi::fail;
using qiddm_capi::kMaxLds;

int check_unitary_circuit(const qiddm_circuit_t* c) {
  int rc = check_circuit(c);
  if (rc != QIDDM_OK) return rc;
  if (c->n_rounds != 1 || c->n_blocks != 1)
    return fail(QIDDM_ERR_UNSUPPORTED, "the circuit unitary is defined for one round and one block of weight-only layers");
  if (c->n_qubits > QIDDM_MAX_QUBITS_FUSED)
    return fail(QIDDM_ERR_UNSUPPORTED, "circuit unitary needs n_qubits <= %d (got %d)", QIDDM_MAX_QUBITS_FUSED,
                c->n_qubits);
  return QIDDM_OK;
}

template <typename T, int N>
int launch_unitary(const qiddm_circuit_t* c, const double* angles, double* u, hipStream_t st) {
  using L = qiddm::Layout<N>;
  using S = qiddm::Smem<T, N>;
  qiddm::KScalars p{};
  p.batch = L::D;
  p.encoding = 1;       // amplitude embedding of the one-hot rows
  p.imprimitive = c->imprimitive;
  p.measure = 0;
  p.n_rounds = 1;
  p.n_blocks = 1;
  p.sel_layers = c->sel_layers;
  p.n_features = L::D;
  p.enc_scale = 1.0;
  p.enc_offset = 0.0;
  p.pad_with = 0.0;
  const int waves = 4;
  const int64_t n_rot = (int64_t)c->sel_layers * N;
  const size_t smem = S::bytes(n_rot, c->imprimitive == QIDDM_IMP_CNOT, waves);
  if (smem > kMaxLds)
    return fail(QIDDM_ERR_UNSUPPORTED, "circuit with %lld Rot gates needs %zu B of LDS", (long long)n_rot, smem);
  auto kern = qiddm::unitary_kernel<T, N>;
  static qiddm_capi::DeviceFlags big_lds_enabled;
  if (smem > 48 * 1024 && !big_lds_enabled.get()) {
    const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    if (ea != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "hipFuncSetAttribute(LDS) failed: %s", hipGetErrorString(ea));
    big_lds_enabled.set();
  }
  const int groups = (L::D + L::SPW - 1) / L::SPW;
  const int blocks = (groups + waves - 1) / waves;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(waves * qiddm::kWave), smem, st, angles, u, p);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "unitary_kernel<n=%d> launch failed: %s", N, hipGetErrorString(e));
  return QIDDM_OK;
}

template <typename T>
int dispatch_unitary(const qiddm_circuit_t* c, const double* angles, double* u, hipStream_t st) {
  switch (c->n_qubits) {
    case 1: return launch_unitary<T, 1>(c, angles, u, st);
    case 2: return launch_unitary<T, 2>(c, angles, u, st);
    case 3: return launch_unitary<T, 3>(c, angles, u, st);
    case 4: return launch_unitary<T, 4>(c, angles, u, st);
    case 5: return launch_unitary<T, 5>(c, angles, u, st);
    case 6: return launch_unitary<T, 6>(c, angles, u, st);
    case 7: return launch_unitary<T, 7>(c, angles, u, st);
    case 8: return launch_unitary<T, 8>(c, angles, u, st);
    case 9: return launch_unitary<T, 9>(c, angles, u, st);
    default: return launch_unitary<T, 10>(c, angles, u, st);
  }
}

struct ConvGeometry {
  int64_t ho, wo, m, k_pad, n_pad, off_w, off_padv, off_bn, total;
  int packed;  // 0 wide, 1 = <= 16 channels in one 32-column tile, 2 = <= 8 channels in one 16-column tile
};

int conv_geometry(int n_qubits, int64_t batch, int64_t in_channels, int64_t height, int64_t width, int64_t kh,
                  int64_t kw, int64_t pad_h, int64_t pad_w, int64_t out_channels, ConvGeometry* g) {
  if (n_qubits < 1 || n_qubits > 12)
    return fail(QIDDM_ERR_UNSUPPORTED, "the unitary route covers n_qubits <= 12 (got %d)", n_qubits);
  if (batch < 0 || in_channels < 1 || height < 1 || width < 1 || kh < 1 || kw < 1 || pad_h < 0 || pad_w < 0 ||
      out_channels < 1)
    return fail(QIDDM_ERR_INVALID, "bad convolution geometry");
  const int64_t d = (int64_t)1 << n_qubits;
  const int64_t f = in_channels * kh * kw;
  if (f > d)
    return fail(QIDDM_ERR_INVALID, "Features must be of length %lld or smaller; got length %lld.", (long long)d,
                (long long)f);
  g->ho = height + 2 * pad_h - kh + 1;
  g->wo = width + 2 * pad_w - kw + 1;
  if (g->ho < 1 || g->wo < 1) return fail(QIDDM_ERR_INVALID, "kernel larger than the padded image");
  if (2 * out_channels > d && !(d == 2 && out_channels == 1))
    return fail(QIDDM_ERR_INVALID, "out_channels=%lld exceeds the %lld even-index probabilities",
                (long long)out_channels, (long long)(d / 2));
  g->m = batch * g->ho * g->wo;
  if (g->m >= ((int64_t)1 << 38)) return fail(QIDDM_ERR_INVALID, "too many output pixels");
  g->k_pad = (f + qiddm::kGemmK - 1) / qiddm::kGemmK * qiddm::kGemmK;
  // 3: four 32-channel tiles per workgroup (many channels: the gather is shared by 128 channels)
  g->packed = out_channels <= 8 ? 2 : out_channels <= 16 ? 1 : out_channels >= 96 ? 3 : 0;
  g->n_pad = g->packed == 2 ? 16 : g->packed == 1 ? 32
             : g->packed == 3 ? (out_channels + 127) / 128 * 256 : (out_channels + 31) / 32 * 64;
  g->off_w = 0;
  g->off_padv = (g->k_pad * g->n_pad * 4 + 255) / 256 * 256;
  g->off_bn = g->off_padv + (g->n_pad * 4 + 255) / 256 * 256;
  g->total = g->off_bn + (g->n_pad * 8 + 255) / 256 * 256;
  return QIDDM_OK;
}

}  // namespace

extern "C" {

int qiddm_circuit_unitary(const qiddm_circuit_t* circ, const double* angles, double* u, void* stream) {
  int rc = check_unitary_circuit(circ);
  if (rc != QIDDM_OK) return rc;
  if (!angles || !u) return fail(QIDDM_ERR_INVALID, "angles/u is NULL");
  hipStream_t st = static_cast<hipStream_t>(stream);
  return circ->dtype == QIDDM_F32 ? dispatch_unitary<float>(circ, angles, u, st)
                                  : dispatch_unitary<double>(circ, angles, u, st);
}

int qiddm_circuit_unitary_wide(const qiddm_circuit_t* circ, const double* angles, double* ut, void* stream) {
  int rc = check_circuit(circ);
  if (rc != QIDDM_OK) return rc;
  if (circ->n_rounds != 1 || circ->n_blocks != 1)
    return fail(QIDDM_ERR_UNSUPPORTED, "the circuit unitary is defined for one round and one block of weight-only layers");
  if (circ->n_qubits < 2 || circ->n_qubits > 12)
    return fail(QIDDM_ERR_UNSUPPORTED, "qiddm_circuit_unitary_wide covers 2 <= n_qubits <= 12 (got %d)", circ->n_qubits);
  if (!angles || !ut) return fail(QIDDM_ERR_INVALID, "angles/ut is NULL");
  const unsigned d = 1u << circ->n_qubits;
  hipLaunchKernelGGL(qiddm::wide_unitary_kernel<0>, dim3(d < 2048 ? d : 2048), dim3(qiddm::kWideThreads), 0,
                     static_cast<hipStream_t>(stream), angles, reinterpret_cast<qiddm::V2<double>*>(ut), circ->n_qubits,
                     circ->sel_layers, circ->imprimitive == QIDDM_IMP_CNOT ? 1 : 0);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "wide_unitary_kernel launch failed: %s", hipGetErrorString(e));
  return QIDDM_OK;
}

int64_t qiddm_qconv_unitary_workspace_bytes(int32_t n_qubits, int64_t in_channels, int64_t kh, int64_t kw,
                                            int64_t out_channels) {
  ConvGeometry g;
  const int rc = conv_geometry(n_qubits, 1, in_channels, kh, kw, kh, kw, 0, 0, out_channels, &g);
  if (rc != QIDDM_OK) return rc;
  return g.total;
}

int qiddm_qconv_unitary_forward(int32_t n_qubits, const double* u, const double* x, int64_t batch,
                                int64_t in_channels, int64_t height, int64_t width, int64_t kh, int64_t kw,
                                int64_t pad_h, int64_t pad_w, int64_t out_channels, int32_t upsample2x,
                                const qiddm_batchnorm_t* bn, int32_t u_transposed, double* y, void* workspace,
                                int64_t workspace_bytes, void* stream) {
  // with upsample2x the stored image is (height, width) and the convolution sees its bilinear x2
  const int64_t h_eff = upsample2x ? 2 * height : height, w_eff = upsample2x ? 2 * width : width;
  ConvGeometry g;
  int rc = conv_geometry(n_qubits, batch, in_channels, h_eff, w_eff, kh, kw, pad_h, pad_w, out_channels, &g);
  if (rc != QIDDM_OK) return rc;
  if (batch == 0) return QIDDM_OK;
  // u == NULL: `workspace` still holds the packing of an earlier call for the same unitary and batch norm (an eval-mode
  // layer whose weights have not changed): the pack launch is skipped
  if (!x || !y) return fail(QIDDM_ERR_INVALID, "x/y is NULL");
  if (bn && (!bn->running_mean || !bn->running_var || !(bn->eps >= 0.0)))
    return fail(QIDDM_ERR_INVALID, "batch norm needs running_mean / running_var and eps >= 0");
  if (!workspace || workspace_bytes < g.total)
    return fail(QIDDM_ERR_INVALID, "workspace of %lld B needed (qiddm_qconv_unitary_workspace_bytes), got %lld",
                (long long)g.total, (long long)workspace_bytes);
  hipStream_t st = static_cast<hipStream_t>(stream);
  unsigned char* ws = static_cast<unsigned char*>(workspace);
  float* w = reinterpret_cast<float*>(ws + g.off_w);
  float* padv = reinterpret_cast<float*>(ws + g.off_padv);
  double* bnv = reinterpret_cast<double*>(ws + g.off_bn);
  const int d = 1 << n_qubits;
  const int f = (int)(in_channels * kh * kw);
  if (u)
    hipLaunchKernelGGL(qiddm::qconv_pack_kernel, dim3((unsigned)g.n_pad), dim3(256), 0, st, u, u_transposed ? 1 : 0, d, f,
                       (int)out_channels, (int)g.k_pad, (int)g.n_pad, g.packed == 3 ? 0 : g.packed, w, padv,
                       bn ? bn->weight : nullptr, bn ? bn->bias : nullptr, bn ? bn->running_mean : nullptr,
                       bn ? bn->running_var : nullptr, bn ? bn->eps : 0.0, bnv, (int)(g.n_pad / 2));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "qconv_pack_kernel launch failed: %s", hipGetErrorString(e));
  qiddm::GemmConv gc{};
  gc.C = (int32_t)in_channels;
  gc.H = (int32_t)h_eff;
  gc.W = (int32_t)w_eff;
  gc.kh = (int32_t)kh;
  gc.kw = (int32_t)kw;
  gc.ph = (int32_t)pad_h;
  gc.pw = (int32_t)pad_w;
  gc.Ho = (int32_t)g.ho;
  gc.Wo = (int32_t)g.wo;
  gc.C_out = (int32_t)out_channels;
  gc.F = f;
  gc.K_pad = (int32_t)g.k_pad;
  gc.N_pad = (int32_t)g.n_pad;
  gc.bn_stride = (int32_t)(g.n_pad / 2);
  gc.upsample = upsample2x ? 1 : 0;
  gc.Hs = (int32_t)height;
  gc.Ws = (int32_t)width;
  gc.has_bn = bn ? 1 : 0;
  gc.M = g.m;
  gc.pad_norm2 = 0.25 * (double)(d - f);
  gc.post_scale = 0.5 * (double)d;
  gc.stamps = qiddm_capi::stamp_buffer(8);
  // same-size convolutions up to 32 output channels: the patch matrix from an LDS copy of the image (qsim_qconv_fwd.h)
  // instead of kh kw gathered loads per input element.  QIDDM_QCONV_GATHER=1 keeps the gathering kernel (A/B).
  static const bool env_gather = std::getenv("QIDDM_QCONV_GATHER") != nullptr;
  // (1 x 1 layers have no patch reuse to exploit: the gathering kernel streams them faster -- 93 vs 110 us at 16 -> 8 channels,
  //  2560 x 28 x 28 pixels)
  if (!env_gather && !upsample2x && kh * kw > 1 && g.ho == h_eff && g.wo == w_eff && g.packed != 3 && g.n_pad <= 64 &&
      in_channels < 0xffff && kh * kw <= 31 && qiddm::fwd_xs(gc) < 0xffff &&
      batch * in_channels * h_eff * w_eff < ((int64_t)1 << 31) && g.m * out_channels < ((int64_t)1 << 31)) {
    const void* kern = nullptr;
    size_t smem = 0;
    const int64_t need = ((int64_t)gc.C * qiddm::fwd_xs(gc) + qiddm::kFwdThreads - 1) / qiddm::kFwdThreads;
#define QIDDM_FWD_PICK(NCOL)                                                                                       \
  do {                                                                                                            \
    smem = qiddm::fwd_lds_bytes<NCOL>(gc);                                                                         \
    kern = need <= 1 ? reinterpret_cast<const void*>(qiddm::qconv_fwd_halo_kernel<NCOL, 1>)                         \
           : need <= 4 ? reinterpret_cast<const void*>(qiddm::qconv_fwd_halo_kernel<NCOL, 4>)                      \
           : need <= 12 ? reinterpret_cast<const void*>(qiddm::qconv_fwd_halo_kernel<NCOL, 12>)                    \
                        : reinterpret_cast<const void*>(qiddm::qconv_fwd_halo_kernel<NCOL, 20>);                   \
  } while (0)
    if (g.n_pad == 16)
      QIDDM_FWD_PICK(16);
    else if (g.n_pad == 32)
      QIDDM_FWD_PICK(32);
    else
      QIDDM_FWD_PICK(64);
#undef QIDDM_FWD_PICK
    if (need <= 40 && smem <= kMaxLds) {
      if (smem > 48 * 1024) {
        const hipError_t ea = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
        if (ea != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "hipFuncSetAttribute(LDS) failed: %s", hipGetErrorString(ea));
      }
      const int64_t tiles = (g.m + qiddm::kFwdTile - 1) / qiddm::kFwdTile;
      // resident workgroups only (each stages the packed operand once): by LDS and by the registers of the variant
      const int64_t by_regs = need <= 1 ? 6 : need <= 4 ? 5 : need <= 12 ? 3 : 2;
      const int64_t per_cu = std::max<int64_t>(1, std::min<int64_t>(by_regs, (int64_t)(kMaxLds / (smem + 1024))));
      const unsigned grid = (unsigned)std::min<int64_t>(tiles, 256 * per_cu);
      const float* wc = w;
      const float* pc = padv;
      const double* bc = bnv;
      void* args[] = {(void*)&x, (void*)&wc, (void*)&pc, (void*)&bc, (void*)&y, (void*)&gc};
      e = hipLaunchKernel(kern, dim3(grid), dim3(qiddm::kFwdThreads), args, smem, st);
      if (e != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "qconv_fwd_halo_kernel launch failed: %s", hipGetErrorString(e));
      return QIDDM_OK;
    }
  }
  const int64_t mblocks = (g.m + qiddm::kGemmM - 1) / qiddm::kGemmM;
  if (mblocks > 0x7fffffff) return fail(QIDDM_ERR_UNSUPPORTED, "too many output pixels for one launch");
#define QIDDM_GEMM_LAUNCH(GRID_Y, ...)                                                                             \
  do {                                                                                                            \
    if (upsample2x)                                                                                               \
      hipLaunchKernelGGL((qiddm::__VA_ARGS__ true>), dim3((unsigned)mblocks, (unsigned)(GRID_Y)),                  \
                         dim3(4 * qiddm::kWave), 0, st, x, w, padv, bnv, y, gc);                                  \
    else                                                                                                          \
      hipLaunchKernelGGL((qiddm::__VA_ARGS__ false>), dim3((unsigned)mblocks, (unsigned)(GRID_Y)),                 \
                         dim3(4 * qiddm::kWave), 0, st, x, w, padv, bnv, y, gc);                                  \
  } while (0)
  if (g.packed == 3)
    QIDDM_GEMM_LAUNCH(g.n_pad / 256, qconv_gemm_wide_kernel<);
  else if (g.packed == 2)
    QIDDM_GEMM_LAUNCH(1, qconv_gemm_kernel<2,);
  else if (g.packed == 1)
    QIDDM_GEMM_LAUNCH(1, qconv_gemm_kernel<1,);
  else
    QIDDM_GEMM_LAUNCH(g.n_pad / 64, qconv_gemm_kernel<0,);
#undef QIDDM_GEMM_LAUNCH
  e = hipGetLastError();
  if (e != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "qconv_gemm_kernel launch failed: %s", hipGetErrorString(e));
  return QIDDM_OK;
}

namespace {
// resident workgroups of the thin-product backward (= per-workgroup h slabs).  The kernel's tiles wait on gathered
// global loads, so it wants every workgroup the LDS admits: up to four per CU for narrow layers (<= 160 patch
// features: <= 40 KB of tile per workgroup with 8 channels), two otherwise.
int64_t train_grid(int64_t pixels_total, int64_t features) {
  const int64_t tiles = (pixels_total + qiddm::kTcTile - 1) / qiddm::kTcTile;
  const int64_t cap = features <= 40 ? 2048 : (features <= 160 ? 1024 : 512);
  return tiles < 1 ? 1 : (tiles < cap ? tiles : cap);
}
// thread groups of the h product (each: all columns, a share of the tile's pixels)
int train_groups(int64_t features) {
  const int64_t g = qiddm::kTcThreads / (features + 1);
  return g < 1 ? 1 : (g > 8 ? 8 : (int)g);
}
}  // namespace

int64_t qiddm_qconv_train_partials(int64_t batch, int64_t height_out, int64_t width_out, int64_t features) {
  if (batch < 0 || height_out < 1 || width_out < 1 || features < 1) return -1;
  return train_grid(batch * height_out * width_out, features);
}

extern "C++" {
namespace {
// the matrix-core variant of the thin-product backward for this layer (nullptr: the VALU kernel keeps it)
struct TmChoice {
  const void* kern = nullptr;
  size_t smem = 0;
};
// the BatchNorm-folding variant of a shape (8 / 16 row channels only: see the kernel's BN parameter); nullptr otherwise
template <int CO, int J, int SK, int SC>
const void* tm_bn_kernel() {
  if constexpr (CO <= 16)
    return reinterpret_cast<const void*>(qiddm::qconv_train_backward_mfma_kernel<CO, J, double, SK, SC, true>);
  else
    return nullptr;
}
TmChoice train_mfma_choice(int64_t f, int32_t row_channels, bool fits32, bool x32, int64_t kh, int64_t kw, int64_t c_in,
                           bool bn = false) {
  // from 32 patch features on and whenever its LDS tiles fit; the VALU kernel keeps the narrowest layers (9 / 16
  // features: the whole backward of the layer measured 0.37 / 0.55 ms against 0.53 / 0.67 on the MFMA kernel at
  // 2560 x 28 x 28 pixels, tools/stamp_qconv_train.py; from 32 features on the MFMA kernel wins) and the widest.
  // QIDDM_QCONV_VALU=1 / QIDDM_QCONV_MFMA=1: kernel experiments (A/B on the same box)
  static const bool env_mfma = std::getenv("QIDDM_QCONV_MFMA") != nullptr;
  static const bool env_valu = std::getenv("QIDDM_QCONV_VALU") != nullptr;
  TmChoice ch;
  // (below 32 features only the 1 x 1 up-convolution of unet_simple, whose shape is compiled in: 0.50 vs 0.55 ms)
  const bool narrow_ok = f == 16 && kh == 1 && kw == 1 && c_in == 16 && row_channels == 8;
  if (env_valu || !fits32 || (f < 32 && !env_mfma && !narrow_ok)) return ch;
  const int jbm = (int)((qiddm::tm_fcols((int)f) / 16 + qiddm::kTmWaves - 1) / qiddm::kTmWaves);
#define QIDDM_TM_CASE(CO, J)                                                                                       \
  if (!ch.kern && row_channels == CO && jbm <= J && qiddm::tm_lds_bytes_bn<CO>((int)f) <= kMaxLds) {                   \
    ch.smem = qiddm::tm_lds_bytes_bn<CO>((int)f);                                                                     \
    ch.kern = x32 ? reinterpret_cast<const void*>(qiddm::qconv_train_backward_mfma_kernel<CO, J, float>)           \
                  : (bn ? tm_bn_kernel<CO, J, 0, 0>()                                                              \
                        : reinterpret_cast<const void*>(qiddm::qconv_train_backward_mfma_kernel<CO, J, double>));  \
  }
  // the layer shapes of unet_simple with kernel extent and channel count compiled in (no control flow in the gather)
  static const bool env_generic = std::getenv("QIDDM_QCONV_GENERIC") != nullptr;
#define QIDDM_TM_SCASE(CO, J, SK, SC)                                                                              \
  if (!ch.kern && !env_generic && row_channels == CO && jbm <= J && kh == SK && kw == SK && c_in == SC &&          \
      qiddm::tm_lds_bytes_bn<CO>((int)f) <= kMaxLds) {                                                                \
    ch.smem = qiddm::tm_lds_bytes_bn<CO>((int)f);                                                                     \
    ch.kern = x32 ? reinterpret_cast<const void*>(qiddm::qconv_train_backward_mfma_kernel<CO, J, float, SK, SC>)   \
                  : (bn ? tm_bn_kernel<CO, J, SK, SC>()                                                            \
                        : reinterpret_cast<const void*>(qiddm::qconv_train_backward_mfma_kernel<CO, J, double, SK, SC>)); \
  }
  QIDDM_TM_SCASE(8, 2, 1, 16)
  QIDDM_TM_SCASE(8, 4, 3, 16)
  QIDDM_TM_SCASE(16, 2, 3, 8)
  QIDDM_TM_SCASE(16, 2, 1, 32)
  QIDDM_TM_SCASE(16, 8, 3, 32)
  QIDDM_TM_SCASE(32, 4, 3, 16)
#undef QIDDM_TM_SCASE
  QIDDM_TM_CASE(8, 2)
  QIDDM_TM_CASE(8, 4)
  QIDDM_TM_CASE(8, 8)
  QIDDM_TM_CASE(16, 2)
  QIDDM_TM_CASE(16, 4)
  QIDDM_TM_CASE(16, 8)
  QIDDM_TM_CASE(32, 2)
  QIDDM_TM_CASE(32, 4)
  QIDDM_TM_CASE(32, 8)
#undef QIDDM_TM_CASE
  return ch;
}
bool train_fits32(int64_t batch, int64_t in_channels, int64_t height, int64_t width, int64_t out_channels, int64_t ho,
                  int64_t wo) {
  // the MFMA kernel addresses x and grad_y with 32-bit element offsets
  return batch * in_channels * height * width < ((int64_t)1 << 32) && batch * out_channels * ho * wo < ((int64_t)1 << 32);
}
int train_backward(int32_t n_qubits, const void* x, bool x32, int64_t batch, int64_t in_channels, int64_t height,
                   int64_t width, int64_t kh, int64_t kw, int64_t pad_h, int64_t pad_w, const double* grad_y,
                   int64_t out_channels, const float* rows, int32_t row_channels, float* grad_features_t,
                   float* pixel_rows, float* h_partials, double* grad_x, void* stream, const double* bn_y = nullptr,
                   const double* bn_coef = nullptr, int64_t grad_y_batch_stride = 0);

// dL/dx from per-pixel rows (qsim_qconv_dx.h) instead of feature gradients + fold: the matrix-core kernel, a same-size
// convolution, at most 32 input channels, LDS of the dx kernel within the limit.  QIDDM_QCONV_FOLD=1 switches it off.
struct DxChoice {
  const void* kern = nullptr;
  size_t smem = 0;
};
DxChoice train_dx_choice(const qiddm::TrainConv& tc, int32_t row_channels, bool mfma) {
  static const bool env_fold = std::getenv("QIDDM_QCONV_FOLD") != nullptr;
  DxChoice ch;
  if (env_fold || !mfma || tc.Ho != tc.H || tc.Wo != tc.W || tc.C > 32 || tc.kh * tc.kw > 32) return ch;
  // (32-bit offsets inside the kernel)
  if (tc.M * tc.C >= ((int64_t)1 << 31) || tc.M * (2 * (int64_t)row_channels + 1) >= ((int64_t)1 << 31)) return ch;
  size_t smem = 0;
  const void* kern = nullptr;
  const bool one = tc.C <= 16;   // one block of 16 input channels, or two
  // float4 of a tile's source rows per thread (the kernel prefetches 2, 4 or 6 of them a tile ahead)
  const int64_t k2 = 2 * (int64_t)row_channels;
  const int64_t slots = ((qiddm::kDxTile + 2 * qiddm::dx_halo(tc)) * (k2 / 4) + qiddm::kDxThreads - 1) / qiddm::kDxThreads;
#define QIDDM_DX_PICK(K2)                                                                                         \
  do {                                                                                                            \
    smem = qiddm::dx_lds_bytes<K2>(tc);                                                                           \
    if (one)                                                                                                      \
      kern = slots <= 2 ? reinterpret_cast<const void*>(qiddm::qconv_dx_kernel<K2, 1, 2>)                          \
                        : (slots <= 4 ? reinterpret_cast<const void*>(qiddm::qconv_dx_kernel<K2, 1, 4>)            \
                                      : reinterpret_cast<const void*>(qiddm::qconv_dx_kernel<K2, 1, 6>));          \
    else                                                                                                          \
      kern = slots <= 2 ? reinterpret_cast<const void*>(qiddm::qconv_dx_kernel<K2, 2, 2>)                          \
                        : (slots <= 4 ? reinterpret_cast<const void*>(qiddm::qconv_dx_kernel<K2, 2, 4>)            \
                                      : reinterpret_cast<const void*>(qiddm::qconv_dx_kernel<K2, 2, 6>));          \
  } while (0)
  if (row_channels == 8)
    QIDDM_DX_PICK(16);
  else if (row_channels == 16)
    QIDDM_DX_PICK(32);
  else if (row_channels == 32)
    QIDDM_DX_PICK(64);
#undef QIDDM_DX_PICK
  if (kern && smem <= kMaxLds / 2) {   // (two or more workgroups per CU)
    ch.kern = kern;
    ch.smem = smem;
  }
  return ch;
}
qiddm::TrainConv train_geometry(int32_t n_qubits, int64_t batch, int64_t in_channels, int64_t height, int64_t width,
                                int64_t kh, int64_t kw, int64_t pad_h, int64_t pad_w, int64_t out_channels) {
  const int64_t d = (int64_t)1 << n_qubits, f = in_channels * kh * kw;
  const int64_t ho = height + 2 * pad_h - kh + 1, wo = width + 2 * pad_w - kw + 1;
  qiddm::TrainConv tc;
  std::memset(&tc, 0, sizeof(tc));
  tc.C = (int32_t)in_channels;
  tc.H = (int32_t)height;
  tc.W = (int32_t)width;
  tc.kh = (int32_t)kh;
  tc.kw = (int32_t)kw;
  tc.ph = (int32_t)pad_h;
  tc.pw = (int32_t)pad_w;
  tc.Ho = (int32_t)ho;
  tc.Wo = (int32_t)wo;
  tc.C_out = (int32_t)out_channels;
  tc.F = (int32_t)f;
  tc.M = batch * ho * wo;
  tc.gy_bstride = out_channels * ho * wo;
  tc.pad_norm2 = 0.25f * (float)(d - f);
  tc.post_scale = 0.5f * (float)d;
  return tc;
}
}  // namespace
}  // extern "C++"

int64_t qiddm_qconv_train_dx_elems(int32_t n_qubits, int64_t batch, int64_t in_channels, int64_t height, int64_t width,
                                   int64_t kh, int64_t kw, int64_t pad_h, int64_t pad_w, int64_t out_channels,
                                   int32_t row_channels) {
  if (n_qubits < 1 || n_qubits > 12 || batch < 1 || in_channels < 1 || height < 1 || width < 1 || kh < 1 || kw < 1 ||
      pad_h < 0 || pad_w < 0 || out_channels < 1 || kh > 15 || kw > 15)
    return 0;
  const int64_t ho = height + 2 * pad_h - kh + 1, wo = width + 2 * pad_w - kw + 1;
  if (ho < 1 || wo < 1) return 0;
  const int64_t f = in_channels * kh * kw;
  const qiddm::TrainConv tc = train_geometry(n_qubits, batch, in_channels, height, width, kh, kw, pad_h, pad_w, out_channels);
  const bool mfma = train_mfma_choice(f, row_channels, train_fits32(batch, in_channels, height, width, out_channels, ho, wo),
                                      false, kh, kw, in_channels).kern != nullptr;
  if (!train_dx_choice(tc, row_channels, mfma).kern) return 0;
  return tc.M * (2 * (int64_t)row_channels + 1);
}

int32_t qiddm_qconv_train_bn_ok(int64_t batch, int64_t in_channels, int64_t height, int64_t width, int64_t kh, int64_t kw,
                                int64_t pad_h, int64_t pad_w, int64_t out_channels, int32_t row_channels) {
  if (batch < 1 || in_channels < 1 || height < 1 || width < 1 || kh < 1 || kw < 1 || pad_h < 0 || pad_w < 0 ||
      out_channels < 1)
    return 0;
  const int64_t ho = height + 2 * pad_h - kh + 1, wo = width + 2 * pad_w - kw + 1;
  if (ho < 1 || wo < 1) return 0;
  const int64_t f = in_channels * kh * kw;
  const bool fits = train_fits32(batch, in_channels, height, width, out_channels, ho, wo);
  if (!train_mfma_choice(f, row_channels, fits, false, kh, kw, in_channels).kern) return 1;   // the VALU kernel folds it
  return train_mfma_choice(f, row_channels, fits, false, kh, kw, in_channels, true).kern != nullptr ? 1 : 0;
}

int qiddm_qconv_train_backward_bn(int32_t n_qubits, const double* x, int64_t batch, int64_t in_channels, int64_t height,
                                  int64_t width, int64_t kh, int64_t kw, int64_t pad_h, int64_t pad_w,
                                  const double* grad_out, const double* conv_y, const double* bn_coef,
                                  int64_t out_channels, const float* rows, int32_t row_channels,
                                  float* grad_features_t, float* pixel_rows, float* h_partials, double* grad_x,
                                  void* stream) {
  if (!conv_y || !bn_coef) return fail(QIDDM_ERR_INVALID, "conv_y/bn_coef is NULL");
  if (pixel_rows && (!grad_x || qiddm_qconv_train_dx_elems(n_qubits, batch, in_channels, height, width, kh, kw, pad_h,
                                                            pad_w, out_channels, row_channels) <= 0))
    return fail(QIDDM_ERR_UNSUPPORTED, "pixel_rows: this layer keeps the feature-gradient route");
  return train_backward(n_qubits, x, false, batch, in_channels, height, width, kh, kw, pad_h, pad_w, grad_out,
                        out_channels, rows, row_channels, pixel_rows ? nullptr : grad_features_t, pixel_rows, h_partials,
                        grad_x, stream, conv_y, bn_coef);
}

int qiddm_qconv_train_backward_dx(int32_t n_qubits, const double* x, int64_t batch, int64_t in_channels, int64_t height,
                                  int64_t width, int64_t kh, int64_t kw, int64_t pad_h, int64_t pad_w,
                                  const double* grad_y, int64_t grad_y_batch_stride, int64_t out_channels,
                                  const float* rows, int32_t row_channels, float* pixel_rows, float* h_partials,
                                  double* grad_x, void* stream) {
  if (!pixel_rows || !grad_x) return fail(QIDDM_ERR_INVALID, "pixel_rows/grad_x is NULL");
  if (qiddm_qconv_train_dx_elems(n_qubits, batch, in_channels, height, width, kh, kw, pad_h, pad_w, out_channels,
                                 row_channels) <= 0)
    return fail(QIDDM_ERR_UNSUPPORTED, "this layer keeps the feature-gradient route (qiddm_qconv_train_dx_elems() == 0)");
  return train_backward(n_qubits, x, false, batch, in_channels, height, width, kh, kw, pad_h, pad_w, grad_y,
                        out_channels, rows, row_channels, nullptr, pixel_rows, h_partials, grad_x, stream, nullptr, nullptr,
                        grad_y_batch_stride);
}

int32_t qiddm_qconv_train_x32_ok(int64_t batch, int64_t in_channels, int64_t height, int64_t width, int64_t kh,
                                 int64_t kw, int64_t pad_h, int64_t pad_w, int64_t out_channels, int32_t row_channels) {
  if (batch < 1 || in_channels < 1 || height < 1 || width < 1 || kh < 1 || kw < 1 || pad_h < 0 || pad_w < 0 ||
      out_channels < 1)
    return 0;
  const int64_t ho = height + 2 * pad_h - kh + 1, wo = width + 2 * pad_w - kw + 1;
  if (ho < 1 || wo < 1) return 0;
  const int64_t f = in_channels * kh * kw;
  return train_mfma_choice(f, row_channels, train_fits32(batch, in_channels, height, width, out_channels, ho, wo), true,
                           kh, kw, in_channels)
                 .kern != nullptr
             ? 1
             : 0;
}

int qiddm_qconv_train_backward(int32_t n_qubits, const double* x, int64_t batch, int64_t in_channels, int64_t height,
                               int64_t width, int64_t kh, int64_t kw, int64_t pad_h, int64_t pad_w,
                               const double* grad_y, int64_t out_channels, const float* rows, int32_t row_channels,
                               float* grad_features_t, float* h_partials, double* grad_x, void* stream) {
  return train_backward(n_qubits, x, false, batch, in_channels, height, width, kh, kw, pad_h, pad_w, grad_y,
                        out_channels, rows, row_channels, grad_features_t, nullptr, h_partials, grad_x, stream);
}

int qiddm_qconv_train_backward_x32(int32_t n_qubits, const float* x, int64_t batch, int64_t in_channels,
                                   int64_t height, int64_t width, int64_t kh, int64_t kw, int64_t pad_h, int64_t pad_w,
                                   const double* grad_y, int64_t out_channels, const float* rows,
                                   int32_t row_channels, float* grad_features_t, float* h_partials, double* grad_x,
                                   void* stream) {
  return train_backward(n_qubits, x, true, batch, in_channels, height, width, kh, kw, pad_h, pad_w, grad_y,
                        out_channels, rows, row_channels, grad_features_t, nullptr, h_partials, grad_x, stream);
}

namespace {
int train_backward(int32_t n_qubits, const void* x, bool x32, int64_t batch, int64_t in_channels, int64_t height,
                   int64_t width, int64_t kh, int64_t kw, int64_t pad_h, int64_t pad_w, const double* grad_y,
                   int64_t out_channels, const float* rows, int32_t row_channels, float* grad_features_t,
                   float* pixel_rows, float* h_partials, double* grad_x, void* stream, const double* bn_y,
                   const double* bn_coef, int64_t grad_y_batch_stride) {
  if ((bn_y == nullptr) != (bn_coef == nullptr)) return fail(QIDDM_ERR_INVALID, "bn_y and bn_coef go together");
  if (n_qubits < 1 || n_qubits > 12) return fail(QIDDM_ERR_UNSUPPORTED, "n_qubits=%d outside 1..12", n_qubits);
  if (batch < 1 || in_channels < 1 || height < 1 || width < 1 || kh < 1 || kw < 1 || pad_h < 0 || pad_w < 0 ||
      out_channels < 1)
    return fail(QIDDM_ERR_INVALID, "bad convolution geometry");
  if (kh > 15 || kw > 15 || in_channels * height * width >= (1 << 24))
    return fail(QIDDM_ERR_UNSUPPORTED, "kernel larger than 15 or image plane stack beyond 2^24 elements");
  const int64_t d = (int64_t)1 << n_qubits, f = in_channels * kh * kw;
  if (f > d) return fail(QIDDM_ERR_INVALID, "in_channels*kh*kw=%lld exceeds 2^n=%lld", (long long)f, (long long)d);
  if (2 * out_channels > d) return fail(QIDDM_ERR_INVALID, "out_channels beyond the even-index probabilities");
  if (out_channels > row_channels) return fail(QIDDM_ERR_INVALID, "out_channels > row_channels");
  const int64_t ho = height + 2 * pad_h - kh + 1, wo = width + 2 * pad_w - kw + 1;
  if (ho < 1 || wo < 1) return fail(QIDDM_ERR_INVALID, "kernel larger than the padded image");
  if (batch * ho * wo >= ((int64_t)1 << 40)) return fail(QIDDM_ERR_INVALID, "too many output pixels");
  if (!x || !grad_y || !rows || (!grad_features_t && !pixel_rows) || !h_partials)
    return fail(QIDDM_ERR_INVALID, "x/grad_y/rows/grad_features_t/h_partials is NULL");
  const int jch = (int)((f + 1 + qiddm::kTcThreads - 1) / qiddm::kTcThreads);
  qiddm::TrainConv tc = train_geometry(n_qubits, batch, in_channels, height, width, kh, kw, pad_h, pad_w, out_channels);
  tc.groups = train_groups(f);
  tc.bn_y = bn_y;
  tc.bn_coef = bn_coef;
  if (grad_y_batch_stride != 0) {
    if (grad_y_batch_stride < tc.gy_bstride) return fail(QIDDM_ERR_INVALID, "grad_y_batch_stride smaller than an image");
    if (batch * grad_y_batch_stride >= ((int64_t)1 << 32))
      return fail(QIDDM_ERR_UNSUPPORTED, "strided grad_y beyond 2^32 elements");
    tc.gy_bstride = grad_y_batch_stride;
  }
  tc.stamps = qiddm_capi::stamp_buffer(8);
  const unsigned grid = (unsigned)train_grid(batch * ho * wo, f);
  hipStream_t st = static_cast<hipStream_t>(stream);
  size_t smem = 0;
  const void* kern = nullptr;
  unsigned threads = qiddm::kTcThreads;
  // the three products on the f32 matrix cores (qsim_qconv_train_mfma.h) where train_mfma_choice() says so
  TmChoice tm = train_mfma_choice(f, row_channels,
                                  train_fits32(batch, in_channels, height, width, out_channels, ho, wo), x32, kh, kw,
                                  in_channels, bn_y != nullptr);
  if (bn_y && tm.smem && !tm.kern)
    return fail(QIDDM_ERR_UNSUPPORTED, "no BatchNorm-folding variant for this layer (qiddm_qconv_train_bn_ok() == 0)");
  if (tm.kern) {
    kern = tm.kern;
    smem = tm.smem;
    threads = qiddm::kTmThreads;
  } else if (x32) {
    return fail(QIDDM_ERR_UNSUPPORTED, "float32 x is taken by the matrix-core kernel only (qiddm_qconv_train_x32_ok)");
  }
#define QIDDM_TC_CASE(CO, J)                                                                      \
  if (!kern && row_channels == CO && jch == J) {                                                           \
    smem = qiddm::tc_lds_bytes<CO>((int)f);                                                       \
    kern = reinterpret_cast<const void*>(qiddm::qconv_train_backward_kernel<CO, J>);              \
  }
  QIDDM_TC_CASE(8, 1)
  QIDDM_TC_CASE(16, 1)
  QIDDM_TC_CASE(32, 1)
#undef QIDDM_TC_CASE
  if (!kern)
    return fail(QIDDM_ERR_UNSUPPORTED, "unitary-route backward: row_channels=%d with %lld features is outside "
                "{8,16,32} x 511", row_channels, (long long)f);
  if (smem > kMaxLds) return fail(QIDDM_ERR_UNSUPPORTED, "unitary-route backward needs %zu B of LDS", smem);
  if (smem > 48 * 1024) {
    const hipError_t ea = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    if (ea != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "hipFuncSetAttribute(LDS) failed: %s", hipGetErrorString(ea));
  }
  DxChoice dx;
  if (pixel_rows) {
    dx = train_dx_choice(tc, row_channels, tm.kern != nullptr);
    if (!dx.kern) return fail(QIDDM_ERR_UNSUPPORTED, "the per-pixel-row route does not take this layer");
    tc.wpix = pixel_rows;
  }
  void* args[] = {(void*)&x, (void*)&grad_y, (void*)&rows, (void*)&grad_features_t, (void*)&h_partials, (void*)&tc};
  hipError_t e = hipLaunchKernel(kern, dim3(grid), dim3(threads), args, smem, st);
  if (e != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "qconv_train_backward_kernel launch failed: %s", hipGetErrorString(e));
  if (pixel_rows) {
    if (dx.smem > 48 * 1024) {
      const hipError_t ea = hipFuncSetAttribute(dx.kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
      if (ea != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "hipFuncSetAttribute(LDS) failed: %s", hipGetErrorString(ea));
    }
    // resident workgroups only (each stages the rows table once and walks its tiles): up to eight per CU by threads
    const int64_t dtiles = (tc.M + qiddm::kDxTile - 1) / qiddm::kDxTile;
    const int64_t per_cu = std::min<int64_t>(8, (int64_t)(kMaxLds / (dx.smem + 1024)));
    const int64_t resident = 256 * std::max<int64_t>(1, per_cu);
    const unsigned dgrid = (unsigned)(dtiles < resident ? dtiles : resident);
    const double* xd = static_cast<const double*>(x);
    const float* wp = pixel_rows;
    void* dargs[] = {(void*)&xd, (void*)&wp, (void*)&rows, (void*)&grad_x, (void*)&tc};
    e = hipLaunchKernel(dx.kern, dim3(dgrid), dim3(qiddm::kDxThreads), dargs, dx.smem, st);
    if (e != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "qconv_dx_kernel launch failed: %s", hipGetErrorString(e));
    return QIDDM_OK;
  }
  if (grad_x) {
    e = qiddm::launch_fold_t(grad_features_t, grad_x, batch, tc, st);
    if (e != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "qconv_fold_t_kernel launch failed: %s", hipGetErrorString(e));
  }
  return QIDDM_OK;
}
}  // namespace

int qiddm_qconv_fold_features(const float* grad_features_t, int64_t batch, int64_t in_channels, int64_t height,
                              int64_t width, int64_t kh, int64_t kw, int64_t pad_h, int64_t pad_w, double* grad_x,
                              void* stream) {
  if (batch < 1 || in_channels < 1 || height < 1 || width < 1 || kh < 1 || kw < 1 || pad_h < 0 || pad_w < 0)
    return fail(QIDDM_ERR_INVALID, "bad convolution geometry");
  const int64_t ho = height + 2 * pad_h - kh + 1, wo = width + 2 * pad_w - kw + 1;
  if (ho < 1 || wo < 1) return fail(QIDDM_ERR_INVALID, "kernel larger than the padded image");
  if (!grad_features_t || !grad_x) return fail(QIDDM_ERR_INVALID, "grad_features_t/grad_x is NULL");
  const int64_t total = batch * in_channels * height * width;
  if (total >= ((int64_t)1 << 40)) return fail(QIDDM_ERR_INVALID, "image batch too large");
  qiddm::TrainConv tc;
  std::memset(&tc, 0, sizeof(tc));
  tc.C = (int32_t)in_channels;
  tc.H = (int32_t)height;
  tc.W = (int32_t)width;
  tc.kh = (int32_t)kh;
  tc.kw = (int32_t)kw;
  tc.ph = (int32_t)pad_h;
  tc.pw = (int32_t)pad_w;
  tc.Ho = (int32_t)ho;
  tc.Wo = (int32_t)wo;
  tc.F = (int32_t)(in_channels * kh * kw);
  tc.M = batch * ho * wo;
  const hipError_t e = qiddm::launch_fold_t(grad_features_t, grad_x, batch, tc, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "qconv_fold_t_kernel launch failed: %s", hipGetErrorString(e));
  return QIDDM_OK;
}

int qiddm_qconv_train_rows(int32_t n_qubits, const double* u, int32_t u_transposed, int64_t features,
                           int64_t out_channels, int32_t row_channels, float* rows, void* stream) {
  if (n_qubits < 1 || n_qubits > 12) return fail(QIDDM_ERR_UNSUPPORTED, "n_qubits=%d outside 1..12", n_qubits);
  const int64_t d = (int64_t)1 << n_qubits;
  if (features < 1 || features > d || out_channels < 1 || 2 * out_channels > d || out_channels > row_channels)
    return fail(QIDDM_ERR_INVALID, "features / out_channels do not fit 2^n or row_channels");
  if (!u || !rows) return fail(QIDDM_ERR_INVALID, "u/rows is NULL");
  hipLaunchKernelGGL(qiddm::qconv_rows_kernel, dim3((unsigned)row_channels), dim3(256), 0,
                     static_cast<hipStream_t>(stream), u, (int)u_transposed, (int)d, (int)features, (int)out_channels,
                     (int)row_channels, rows);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "qconv_rows_kernel launch failed: %s", hipGetErrorString(e));
  return QIDDM_OK;
}

int qiddm_qconv_train_vectors(int32_t n_qubits, const float* h_partials, int64_t n_partials, int64_t features,
                              int64_t out_channels, int32_t row_channels, double* psi0, double* lambda, void* stream) {
  if (n_qubits < 1 || n_qubits > 12) return fail(QIDDM_ERR_UNSUPPORTED, "n_qubits=%d outside 1..12", n_qubits);
  const int64_t d = (int64_t)1 << n_qubits;
  if (features < 1 || features > d || out_channels < 1 || 2 * out_channels > d || out_channels > row_channels)
    return fail(QIDDM_ERR_INVALID, "features / out_channels do not fit 2^n or row_channels");
  if (n_partials < 1 || n_partials > (1 << 20)) return fail(QIDDM_ERR_INVALID, "n_partials out of range");
  if (!h_partials || !psi0 || !lambda) return fail(QIDDM_ERR_INVALID, "h_partials/psi0/lambda is NULL");
  hipLaunchKernelGGL(qiddm::qconv_vectors_kernel, dim3((unsigned)((features + 1 + 7) / 8), (unsigned)out_channels),
                     dim3(256), 0,
                     static_cast<hipStream_t>(stream), h_partials, (int)n_partials, (int)d, (int)features,
                     (int)out_channels, (int)row_channels, psi0, lambda);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "qconv_vectors_kernel launch failed: %s", hipGetErrorString(e));
  return QIDDM_OK;
}

int qiddm_conv1x1_forward(const double* x, const double* weight, const double* bias, int64_t batch,
                          int64_t in_channels, int64_t out_channels, int64_t hw, double* y, void* stream) {
  if (batch < 0 || in_channels < 1 || out_channels < 1 || hw < 0)
    return fail(QIDDM_ERR_INVALID, "bad 1x1 convolution geometry");
  if (batch == 0 || hw == 0) return QIDDM_OK;
  if (!x || !weight || !y) return fail(QIDDM_ERR_INVALID, "x/weight/y is NULL");
  if (in_channels > (1 << 20) || out_channels > (1 << 20)) return fail(QIDDM_ERR_UNSUPPORTED, "too many channels");
  const int64_t total = batch * hw;
  const int64_t blocks = (total + 255) / 256;
  if (blocks > 0x7fffffff) return fail(QIDDM_ERR_UNSUPPORTED, "too many pixels for one launch");
  hipLaunchKernelGGL(qiddm::conv1x1_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     weight, bias, y, total, hw, (int)in_channels, (int)out_channels);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "conv1x1_kernel launch failed: %s", hipGetErrorString(e));
  return QIDDM_OK;
}

int64_t qiddm_conv1x1_head_partials(int64_t batch, int64_t hw) {
  if (batch < 0 || hw < 0) return -1;
  const int64_t blocks = (batch * hw + 255) / 256;
  return blocks < 1 ? 1 : (blocks < 2048 ? blocks : 2048);
}

int qiddm_conv1x1_head_backward(const double* x, const double* weight, const double* grad_y, int64_t batch,
                                int64_t in_channels, int64_t hw, double* grad_x, double* grad_weight,
                                double* grad_bias, double* partials, void* stream) {
  if (batch < 1 || in_channels < 1 || hw < 1) return fail(QIDDM_ERR_INVALID, "bad 1x1 convolution geometry");
  if (in_channels > qiddm::kHeadMaxC)
    return fail(QIDDM_ERR_UNSUPPORTED, "head backward: at most %d input channels", qiddm::kHeadMaxC);
  if (!x || !weight || !grad_y || !partials) return fail(QIDDM_ERR_INVALID, "x/weight/grad_y/partials is NULL");
  const int64_t total = batch * hw;
  const int64_t grid = qiddm_conv1x1_head_partials(batch, hw);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (in_channels <= 8)
    hipLaunchKernelGGL(qiddm::conv1x1_head_backward_kernel<8>, dim3((unsigned)grid), dim3(256), 0, st, x, weight, grad_y,
                       total, hw, (int)in_channels, grad_x, partials);
  else if (in_channels <= 16)
    hipLaunchKernelGGL(qiddm::conv1x1_head_backward_kernel<16>, dim3((unsigned)grid), dim3(256), 0, st, x, weight,
                       grad_y, total, hw, (int)in_channels, grad_x, partials);
  else
    hipLaunchKernelGGL(qiddm::conv1x1_head_backward_kernel<32>, dim3((unsigned)grid), dim3(256), 0, st, x, weight,
                       grad_y, total, hw, (int)in_channels, grad_x, partials);
  hipLaunchKernelGGL(qiddm::conv1x1_head_finalize_kernel, dim3(1), dim3(256), 0, st, partials, (int)grid,
                     (int)in_channels, grad_weight, grad_bias);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "conv1x1_head_backward launch failed: %s", hipGetErrorString(e));
  return QIDDM_OK;
}

}  // extern "C"
