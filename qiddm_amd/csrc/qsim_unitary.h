// qsim_unitary.h -- the eval-mode route of the quantum convolution.
//
// Reference nn/qconv.py:92-126: in eval mode `_QConv2d_FAST.train(False)` takes the D x D matrix of the
// weight-only sub-circuit (`qml.matrix` of StronglyEntanglingLayers(pi*tanh(w))) once and the per-patch circuit
// becomes AmplitudeEmbedding -> QubitUnitary(U) -> probs.  With the post-processing of :58-69 (scale by D/2,
// clamp, keep the even-index probabilities, first C_out of them) only rows k = 0, 2, ..., 2(C_out-1) of U matter,
// and the embedded state is real, so one output pixel is
//     a_f   = patch_f + 0.1  (f < F = C_in kh kw),  pad 0.5  (F <= f < D),  |a|^2 = sum a_f^2 + 0.25 (D - F)
//     y_c   = clamp( ((a . Re U[2c, :F] + 0.5 sum_{j>=F} Re U[2c, j])^2 + (same with Im)^2) / |a|^2 * D/2, 0, 1 )
// i.e. an implicit-im2col GEMM  (B Ho Wo) x F  times  F x 2 C_out  with a per-row normalisation epilogue: the
// one GEMM on the path.  It runs on the f32-input MFMA (v_mfma_f32_32x32x2_f32: exact f32 products, k-ordered
// fma chain -- cdna_hip_programming.md "FP32-input MFMA").
//
// Two kernels:
//   unitary_kernel      column j of U = the circuit applied to basis state |j>, one wavefront per column (n <= 10)
//   qconv_gemm_kernel   the GEMM + epilogue.  Workgroup = 128 output pixels x 32 channels (re and im tiles),
//                       4 wavefronts x (32 x 32 re, 32 x 32 im) accumulators, K staged through LDS 16 at a time.
#pragma once
#include "qsim_adjoint.h"

namespace qiddm {

struct OneHotSrc {
  int j;
  __device__ __forceinline__ float operator()(int k) const { return k == j ? 1.f : 0.f; }
};

// U[k * D + j] (complex float64, interleaved) for the (1, 1, S, n, 3) angles of a weight-only SEL circuit
template <typename T, int N>
__global__ __launch_bounds__(4 * kWave) void unitary_kernel(const double* __restrict__ angles,
                                                            double* __restrict__ u_out, const KScalars p) {
  using E = Engine<T, N>;
  using L = typename E::L;
  using C = V2<T>;
  constexpr int LB = L::LB, R = L::R, SPW = L::SPW, D = L::D;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int n_rot = p.n_blocks * p.sel_layers * N;
  AdjointEngine<T, N> adj;
  adj.fwd.carve(smem_raw, n_rot);
  adj.fwd.fill_gates_from_angles(angles, n_rot);
  adj.fwd.fill_rings(p.imprimitive == 0);
  __syncthreads();
  adj.dag = adj.fwd;
  const int wave = threadIdx.x >> 6;
  const int swave = adj.fwd.llane >> LB;
  const int waves = blockDim.x >> 6;
  constexpr int groups = (D + SPW - 1) / SPW;
  for (int grp = blockIdx.x * waves + wave; grp < groups; grp += gridDim.x * waves) {
    const int j_raw = grp * SPW + swave;
    const bool valid = j_raw < D;
    const int j = valid ? j_raw : D - 1;
    C psi[R], dx[R];
    T xs[N], cs[N], sn[N], amp_inv;
#pragma unroll
    for (int w = 0; w < N; ++w) xs[w] = (T)0;
    adj.forward_round(p, OneHotSrc{j}, xs, psi, dx, cs, sn, amp_inv);
    if (valid) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int k = (r << LB) | adj.fwd.sub;
        u_out[((size_t)k * D + j) * 2 + 0] = (double)psi[r].x;
        u_out[((size_t)k * D + j) * 2 + 1] = (double)psi[r].y;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// operand packing: W[f][col] float32, K_pad x N_pad, col = ct*64 + {0..31: Re, 32..63: Im} of channel ct*32 + (col&31)
//                  padv[col] = 0.5 * sum_{j >= F} (Re|Im) U[2c][j]
// ---------------------------------------------------------------------------
//                  bn[col] / bn[N_pad + col]: eval-mode BatchNorm folded to  y * scale + shift  per channel
__global__ void qconv_pack_kernel(const double* __restrict__ u, int D, int F, int C_out, int K_pad, int N_pad,
                                  float* __restrict__ w, float* __restrict__ padv,
                                  const double* __restrict__ bn_weight, const double* __restrict__ bn_bias,
                                  const double* __restrict__ bn_mean, const double* __restrict__ bn_var, double bn_eps,
                                  double* __restrict__ bn) {
  const int col = blockIdx.x;  // one workgroup per packed column
  const int c = (col >> 6) * 32 + (col & 31);
  const int part = (col >> 5) & 1;
  const bool live = c < C_out;
  for (int f = threadIdx.x; f < K_pad; f += blockDim.x)
    w[(size_t)f * N_pad + col] = (live && f < F) ? (float)u[((size_t)(2 * c) * D + f) * 2 + part] : 0.f;
  __shared__ double s_part[256];
  double acc = 0.0;
  if (live)
    for (int j = F + threadIdx.x; j < D; j += blockDim.x) acc += u[((size_t)(2 * c) * D + j) * 2 + part];
  s_part[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int i = 0; i < (int)blockDim.x; ++i) tot += s_part[i];
    padv[col] = (float)(0.5 * tot);
    if (bn != nullptr && part == 0) {
      double scale = 1.0, shift = 0.0;
      if (live && bn_mean != nullptr) {
        // torch batch_norm (eval): (y - mean) / sqrt(var + eps) * weight + bias
        const double inv = 1.0 / sqrt(bn_var[c] + bn_eps);
        scale = (bn_weight ? bn_weight[c] : 1.0) * inv;
        shift = (bn_bias ? bn_bias[c] : 0.0) - bn_mean[c] * scale;
      }
      bn[(col >> 6) * 32 + (col & 31)] = scale;
      bn[N_pad / 2 + (col >> 6) * 32 + (col & 31)] = shift;
    }
  }
}

struct GemmConv {
  int32_t C, H, W, kh, kw, ph, pw, Ho, Wo, C_out, F, K_pad, N_pad;
  int32_t upsample, Hs, Ws, has_bn;  // upsample: the (C, H, W) input is the bilinear x2 of a stored (C, Hs, Ws)
  int64_t M;           // batch * Ho * Wo
  double pad_norm2;    // 0.25 * (D - F)
  double post_scale;   // D / 2
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kGemmM = 128;  // output pixels per workgroup
constexpr int kGemmK = 16;   // K chunk

// torch.nn.Upsample(scale_factor=2, mode="bilinear") (align_corners=False) of one (Hs, Ws) plane at (ii, jj):
// ATen area_pixel_compute_source_index with scale 0.5, float64 accumulation
__device__ __forceinline__ double bilinear2x(const double* __restrict__ plane, int Hs, int Ws, int ii, int jj) {
  double sh = 0.5 * (ii + 0.5) - 0.5, sw = 0.5 * (jj + 0.5) - 0.5;
  sh = sh < 0.0 ? 0.0 : sh;
  sw = sw < 0.0 ? 0.0 : sw;
  const int h1 = (int)sh, w1 = (int)sw;
  const int h1p = h1 < Hs - 1 ? 1 : 0, w1p = w1 < Ws - 1 ? 1 : 0;
  const double h1l = sh - h1, h0l = 1.0 - h1l, w1l = sw - w1, w0l = 1.0 - w1l;
  const double* r0 = plane + (size_t)h1 * Ws + w1;
  const double* r1 = r0 + (size_t)h1p * Ws;
  return h0l * (w0l * r0[0] + w1l * r0[w1p]) + h1l * (w0l * r1[0] + w1l * r1[w1p]);
}

__global__ __launch_bounds__(4 * kWave) void qconv_gemm_kernel(const double* __restrict__ x,
                                                               const float* __restrict__ w,
                                                               const float* __restrict__ padv,
                                                               const double* __restrict__ bn,
                                                               double* __restrict__ y, const GemmConv g) {
  __shared__ float s_a[kGemmK][kGemmM];   // [k][m]: lanes 0-31 / 32-63 of an A read hit consecutive words
  __shared__ float s_b[kGemmK][64];       // [k][col]
  __shared__ float s_n2[2][kGemmM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t m0 = (int64_t)blockIdx.x * kGemmM;
  const int ct = blockIdx.y;  // 32-channel tile
  // ---- this thread's staging duty: row m_s, eight consecutive k of every chunk ------------------------
  const int m_s = tid & (kGemmM - 1), kh_s = (tid >> 7) * 8;
  const int64_t mg = m0 + m_s;
  const bool m_ok = mg < g.M;
  const int64_t pixels = (int64_t)g.Ho * g.Wo;
  const int64_t bi = m_ok ? mg / pixels : 0;
  const int pix = m_ok ? (int)(mg - bi * pixels) : 0;
  const int oi = pix / g.Wo - g.ph, oj = pix % g.Wo - g.pw;
  const size_t plane = g.upsample ? (size_t)g.Hs * g.Ws : (size_t)g.H * g.W;
  const double* __restrict__ img = x + (size_t)bi * g.C * plane;
  float n2 = 0.f;

  f32x16 acc_re, acc_im;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    acc_re[i] = 0.f;
    acc_im[i] = 0.f;
  }
  const int khw = g.kh * g.kw;
  for (int k0 = 0; k0 < g.K_pad; k0 += kGemmK) {
    // stage A (im2col gather, + 0.1) and B
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int f = k0 + kh_s + u;
      float v = 0.f;
      if (m_ok && f < g.F) {
        const int c = f / khw, rem = f - c * khw;
        const int di = rem / g.kw, dj = rem - di * g.kw;
        const int ii = oi + di, jj = oj + dj;
        double pv = 0.0;
        if (ii >= 0 && ii < g.H && jj >= 0 && jj < g.W)
          pv = g.upsample ? bilinear2x(img + (size_t)c * plane, g.Hs, g.Ws, ii, jj) : img[((size_t)c * g.H + ii) * g.W + jj];
        v = (float)(pv + 0.1);
        n2 = fmaf(v, v, n2);
      }
      s_a[kh_s + u][m_s] = v;
    }
    for (int i = tid; i < kGemmK * 64; i += 4 * kWave)
      s_b[i >> 6][i & 63] = w[(size_t)(k0 + (i >> 6)) * g.N_pad + ct * 64 + (i & 63)];
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < kGemmK; kk += 2) {
      const float a = s_a[kk + (lane >> 5)][wave * 32 + (lane & 31)];
      const float bre = s_b[kk + (lane >> 5)][lane & 31];
      const float bim = s_b[kk + (lane >> 5)][32 + (lane & 31)];
      acc_re = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bre, acc_re, 0, 0, 0);
      acc_im = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bim, acc_im, 0, 0, 0);
    }
    __syncthreads();
  }
  s_n2[tid >> 7][m_s] = n2;
  __syncthreads();
  // ---- epilogue: C/D map col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) ----------------
  const int c = ct * 32 + (lane & 31);
  if (c >= g.C_out) return;
  const float pre = padv[ct * 64 + (lane & 31)], pim = padv[ct * 64 + 32 + (lane & 31)];
  const double bn_scale = g.has_bn ? bn[c] : 1.0, bn_shift = g.has_bn ? bn[g.N_pad / 2 + c] : 0.0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int ml = wave * 32 + row;
    const int64_t m = m0 + ml;
    if (m >= g.M) continue;
    const double norm2 = (double)s_n2[0][ml] + (double)s_n2[1][ml] + g.pad_norm2;
    const double re = (double)(acc_re[r] + pre), im = (double)(acc_im[r] + pim);
    double v = (re * re + im * im) / norm2 * g.post_scale;
    v = fmin(fmax(v, 0.0), 1.0);
    if (g.has_bn) v = v * bn_scale + bn_shift;
    const int64_t b = m / pixels;
    const int64_t px = m - b * pixels;
    y[((size_t)b * g.C_out + c) * pixels + px] = v;
  }
}

// ---------------------------------------------------------------------------
// classical 1x1 convolution in float64 (the `final_conv` of the UNets, reference nn/unet.py:160-166): one thread
// per output pixel, all output channels; x (B, C_in, HW), w (C_out, C_in), y (B, C_out, HW)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv1x1_kernel(const double* __restrict__ x, const double* __restrict__ w,
                                                      const double* __restrict__ bias, double* __restrict__ y,
                                                      int64_t total, int64_t hw, int c_in, int c_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int64_t b = i / hw, px = i - b * hw;
  const double* __restrict__ xp = x + (size_t)b * c_in * hw + px;
  for (int o = 0; o < c_out; ++o) {
    double acc = bias ? bias[o] : 0.0;
    for (int c = 0; c < c_in; ++c) acc = fma(w[(size_t)o * c_in + c], xp[(size_t)c * hw], acc);
    y[((size_t)b * c_out + o) * hw + px] = acc;
  }
}

}  // namespace qiddm
