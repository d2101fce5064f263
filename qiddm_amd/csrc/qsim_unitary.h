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
#include "qsim_adjoint_wide.h"

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
// operand packing: W[f][col] float32, K_pad x N_pad, padv[col] = 0.5 * sum_{j >= F} (Re|Im) U[2c][j].
//   wide   (C_out > 16): 64 columns per 32-channel tile: col = ct*64 + {0..31: Re, 32..63: Im} of channel ct*32 + (col&31)
//   packed (C_out <= 16): 32 columns: {0..15: Re, 16..31: Im} of channel col & 15 -- one MFMA tile holds both parts
//   packed8 (C_out <= 8): 16 columns: {0..7: Re, 8..15: Im} -- one 16x16x4 MFMA tile, no padding columns at 8 channels
// bn[c] / bn[bn_stride + c]: eval-mode BatchNorm folded to  y * scale + shift  per channel
// ---------------------------------------------------------------------------
// u_transposed: u holds U^T (u[j][k] = <k|U|j>, what qiddm_circuit_unitary_wide writes)
__global__ void qconv_pack_kernel(const double* __restrict__ u, int u_transposed, int D, int F, int C_out, int K_pad,
                                  int N_pad, int packed, float* __restrict__ w, float* __restrict__ padv,
                                  const double* __restrict__ bn_weight, const double* __restrict__ bn_bias,
                                  const double* __restrict__ bn_mean, const double* __restrict__ bn_var, double bn_eps,
                                  double* __restrict__ bn, int bn_stride) {
  const int col = blockIdx.x;  // one workgroup per packed column
  // packed: 0 wide, 1 = 16 channels per 32-column tile, 2 = 8 channels per 16-column tile
  const int c = packed == 2 ? (col & 7) : packed == 1 ? (col & 15) : (col >> 6) * 32 + (col & 31);
  const int part = packed == 2 ? (col >> 3) & 1 : packed == 1 ? (col >> 4) & 1 : (col >> 5) & 1;
  const bool live = c < C_out;
  for (int f = threadIdx.x; f < K_pad; f += blockDim.x)
    w[(size_t)f * N_pad + col] = (live && f < F) ? (float)(u_transposed ? u[((size_t)f * D + 2 * c) * 2 + part]
                                                                         : u[((size_t)(2 * c) * D + f) * 2 + part])
                                                 : 0.f;
  __shared__ double s_part[256];
  double acc = 0.0;
  if (live)
    for (int j = F + threadIdx.x; j < D; j += blockDim.x)
      acc += u_transposed ? u[((size_t)j * D + 2 * c) * 2 + part] : u[((size_t)(2 * c) * D + j) * 2 + part];
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
      bn[c] = scale;
      bn[bn_stride + c] = shift;
    }
  }
}

struct GemmConv {
  int32_t C, H, W, kh, kw, ph, pw, Ho, Wo, C_out, F, K_pad, N_pad;
  int32_t bn_stride, pad_;
  int32_t upsample, Hs, Ws, has_bn;  // upsample: the (C, H, W) input is the bilinear x2 of a stored (C, Hs, Ws)
  int64_t M;           // batch * Ho * Wo
  double pad_norm2;    // 0.25 * (D - F)
  double post_scale;   // D / 2
  unsigned long long* stamps;  // diagnostics only (tools/stamp_qconv.py): s_memtime at phase ends of block 0
};

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kGemmM = 128;  // output pixels per workgroup
constexpr int kGemmK = 32;   // K chunk

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

constexpr int kGemmMaxK = 4096;  // F <= D <= 2^12 on this route

// MODE 0: 32 channels per workgroup, separate Re / Im tiles.  MODE 1 (C_out <= 16): Re and Im columns share one
// 32-wide tile (one accumulator, half the MFMAs).  MODE 2 (C_out <= 8): 16 columns on v_mfma_f32_16x16x4_f32
// (two 16-row tiles per wavefront, 32 cycles each): no padding columns at the 8-channel layers of unet_simple.
// UP (compile time = g.upsample): the bilinear x2 in front of an up_conv folded into the gather.  As a run-time branch
// its four-load taps shared registers with the plain path, and the wait-count pass then drained the plain path's gather
// before the chunk's MFMAs (vmcnt waits between the B loads)
template <int MODE, bool UP>
__global__ __launch_bounds__(4 * kWave) void qconv_gemm_kernel(const double* __restrict__ x,
                                                               const float* __restrict__ w,
                                                               const float* __restrict__ padv,
                                                               const double* __restrict__ bn,
                                                               double* __restrict__ y, const GemmConv g) {
  constexpr bool PACKED = MODE != 0;
  constexpr int NB = MODE == 2 ? 16 : MODE == 1 ? 32 : 64;   // B columns per workgroup
  constexpr int CT = MODE == 2 ? 8 : MODE == 1 ? 16 : 32;    // channels per workgroup
  constexpr int KT = kGemmK / 2;         // k values a staging thread owns per chunk
  // LDS: the staging tiles of the K loop and the output tile of the epilogue share one region
  constexpr int kStageBytes = (kGemmK * kGemmM + kGemmK * NB) * 4;
  constexpr int kOutBytes = CT * (kGemmM + 1) * 8;
  __shared__ __attribute__((aligned(16))) unsigned char s_raw[kOutBytes > kStageBytes ? kOutBytes : kStageBytes];
  __shared__ uint32_t s_tap[kGemmMaxK];       // feature f -> offset of its tap inside one image | di << 24 | dj << 28
  __shared__ float s_n2[2][kGemmM];
  float (*s_a)[kGemmM] = reinterpret_cast<float (*)[kGemmM]>(s_raw);                       // [k][m]
  float (*s_b)[NB] = reinterpret_cast<float (*)[NB]>(s_raw + kGemmK * kGemmM * 4);          // [k][col]
  double (*s_out)[kGemmM + 1] = reinterpret_cast<double (*)[kGemmM + 1]>(s_raw);           // [channel][m]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t m0 = (int64_t)blockIdx.x * kGemmM;
  const int ct = blockIdx.y;  // channel tile
  const bool stamp = g.stamps != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0;
  if (stamp) g.stamps[0] = __builtin_amdgcn_s_memtime();
  const int khw = g.kh * g.kw;
  const size_t plane = UP ? (size_t)g.Hs * g.Ws : (size_t)g.H * g.W;
  for (int f = tid; f < g.K_pad; f += 4 * kWave) {
    const int c = f / khw, rem = f - c * khw;
    const int di = rem / g.kw, dj = rem - di * g.kw;
    // (upsample: the offset field carries the channel; the taps are interpolated)
    const uint32_t off = UP ? (uint32_t)c : (uint32_t)(c * plane) + (uint32_t)(di * g.W + dj);
    s_tap[f] = off | ((uint32_t)di << 24) | ((uint32_t)dj << 28);
  }
  // ---- this thread's staging duty: row m_s, KT consecutive k of every chunk ------------------------------
  const int m_s = tid & (kGemmM - 1), kh_s = (tid >> 7) * KT;
  const int64_t mg = m0 + m_s;
  const bool m_ok = mg < g.M;
  const int64_t pixels = (int64_t)g.Ho * g.Wo;
  const int64_t bi = m_ok ? mg / pixels : 0;
  const int pix = m_ok ? (int)(mg - bi * pixels) : 0;
  const int oi = pix / g.Wo - g.ph, oj = pix % g.Wo - g.pw;
  const double* __restrict__ img = x + (size_t)bi * g.C * plane;
  const double* __restrict__ corner = img + (int64_t)oi * g.W + oj;  // tap (0, 0) of channel 0 (not dereferenced if outside)
  float n2 = 0.f;
  __syncthreads();
  if (stamp) g.stamps[1] = __builtin_amdgcn_s_memtime();

  // registers of the chunk in flight: the gather for chunk k+1 is issued before the MFMAs of chunk k
  double pv[KT];
  uint32_t in_mask = 0;   // bit u: tap u of the chunk in flight lies inside the image
  float bv[(kGemmK * NB) / (4 * kWave)];
  auto fetch = [&](int k0) {
    // branch-free: every load goes to a valid address (the image origin when the tap is outside) so that all KT
    // of them are in flight together; the select happens on the loaded value
    uint32_t tap[KT];
    in_mask = 0;
#pragma unroll
    for (int u = 0; u < KT; ++u) tap[u] = s_tap[k0 + kh_s + u];
#pragma unroll
    for (int u = 0; u < KT; ++u) {
      const int f = k0 + kh_s + u;
      const int ii = oi + (int)((tap[u] >> 24) & 15u), jj = oj + (int)(tap[u] >> 28);
      const uint32_t off = tap[u] & 0xFFFFFFu;
      const bool in = m_ok && f < g.F && ii >= 0 && ii < g.H && jj >= 0 && jj < g.W;
      // the select on the loaded value is deferred to stage(): done here it put a wait for the gather in front of
      // the chunk's MFMAs (s_waitcnt vmcnt right after the issue), i.e. the load latency on the critical path
      if constexpr (UP) {
        const double v = bilinear2x(img + (in ? (size_t)off * plane : 0), g.Hs, g.Ws, in ? ii : 0, in ? jj : 0);
        pv[u] = in ? v : 0.0;
        in_mask |= 1u << u;
      } else {
        pv[u] = *(in ? corner + off : img);
        in_mask |= (uint32_t)in << u;
      }
    }
#pragma unroll
    for (int i = 0; i < (kGemmK * NB) / (4 * kWave); ++i) {
      const int e = tid + i * 4 * kWave;
      bv[i] = w[(size_t)(k0 + e / NB) * g.N_pad + ct * NB + (e % NB)];
    }
  };
  auto stage = [&](int k0) {
#pragma unroll
    for (int u = 0; u < KT; ++u) {
      const int f = k0 + kh_s + u;
      float v = 0.f;
      if (m_ok && f < g.F) {
        v = (float)((((in_mask >> u) & 1u) ? pv[u] : 0.0) + 0.1);
        n2 = fmaf(v, v, n2);
      }
      s_a[kh_s + u][m_s] = v;
    }
#pragma unroll
    for (int i = 0; i < (kGemmK * NB) / (4 * kWave); ++i) {
      const int e = tid + i * 4 * kWave;
      s_b[e / NB][e % NB] = bv[i];
    }
  };

  f32x16 acc_re, acc_im;
  f32x4 acc4[2];   // MODE 2: rows 0..15 and 16..31 of the wavefront's strip
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    acc_re[i] = 0.f;
    acc_im[i] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    acc4[0][i] = 0.f;
    acc4[1][i] = 0.f;
  }
  fetch(0);
  for (int k0 = 0; k0 < g.K_pad; k0 += kGemmK) {
    stage(k0);
    __syncthreads();
    if (k0 + kGemmK < g.K_pad) fetch(k0 + kGemmK);
    if constexpr (MODE == 2) {
      // A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15]; C/D col = l & 15, row = (l >> 4) * 4 + reg
#pragma unroll
      for (int kk = 0; kk < kGemmK; kk += 4) {
        const float b = s_b[kk + (lane >> 4)][lane & 15];
        const float a0 = s_a[kk + (lane >> 4)][wave * 32 + (lane & 15)];
        const float a1 = s_a[kk + (lane >> 4)][wave * 32 + 16 + (lane & 15)];
        acc4[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, acc4[0], 0, 0, 0);
        acc4[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, acc4[1], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < kGemmK; kk += 2) {
        const float a = s_a[kk + (lane >> 5)][wave * 32 + (lane & 31)];
        const float bre = s_b[kk + (lane >> 5)][lane & 31];
        acc_re = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bre, acc_re, 0, 0, 0);
        if constexpr (!PACKED) {
          const float bim = s_b[kk + (lane >> 5)][32 + (lane & 31)];
          acc_im = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bim, acc_im, 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  if (stamp) g.stamps[2] = __builtin_amdgcn_s_memtime();
  s_n2[tid >> 7][m_s] = n2;
  __syncthreads();
  // one division per pixel: D/2 over |a|^2
  __shared__ float s_inv[kGemmM];
  if (tid < kGemmM) s_inv[tid] = (float)(g.post_scale / ((double)s_n2[0][tid] + (double)s_n2[1][tid] + g.pad_norm2));
  __syncthreads();
  // ---- epilogue: C/D map col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) ----------------
  // |.|^2 / |a|^2 * D/2, clamp, BatchNorm -> the [channel][pixel] tile in LDS, then rows of consecutive pixels out
  if constexpr (MODE == 2) {
    const int col = lane & 15, cl = col & 7, c = cl;
    const bool live = c < g.C_out;
    const float pre = padv[cl], pim = padv[8 + cl];
    const double bn_scale = (g.has_bn && live) ? bn[c] : 1.0, bn_shift = (g.has_bn && live) ? bn[g.bn_stride + c] : 0.0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ml = wave * 32 + t * 16 + (lane >> 4) * 4 + r;
        // lanes col < 8 hold Re of channel col, lanes col >= 8 hold Im of channel col - 8
        const float other = __shfl_xor(acc4[t][r], 8, kWave);
        const float fre = col < 8 ? acc4[t][r] : other, fim = col < 8 ? other : acc4[t][r];
        const float re = fre + pre, im = fim + pim;
        const float v = fminf(fmaxf((re * re + im * im) * s_inv[ml], 0.f), 1.f);
        if (col < 8) s_out[cl][ml] = (double)v * bn_scale + bn_shift;
      }
    }
  } else {
    const int col = lane & 31;
    const int cl = PACKED ? (col & 15) : col, c = ct * CT + cl;
    const bool live = c < g.C_out;
    const float pre = padv[ct * NB + cl], pim = padv[ct * NB + CT + cl];
    const double bn_scale = (g.has_bn && live) ? bn[c] : 1.0, bn_shift = (g.has_bn && live) ? bn[g.bn_stride + c] : 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ml = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      float fre, fim;
      if constexpr (PACKED) {
        // lanes col < 16 hold Re of channel col, lanes col >= 16 hold Im of channel col - 16
        const float other = __shfl_xor(acc_re[r], 16, kWave);
        fre = col < 16 ? acc_re[r] : other;
        fim = col < 16 ? other : acc_re[r];
      } else {
        fre = acc_re[r];
        fim = acc_im[r];
      }
      // float32 until the clamp (the accumulators are float32), float64 for the normalisation that follows
      const float re = fre + pre, im = fim + pim;
      const float v = fminf(fmaxf((re * re + im * im) * s_inv[ml], 0.f), 1.f);
      if (!PACKED || col < 16) s_out[cl][ml] = (double)v * bn_scale + bn_shift;
    }
  }
  __syncthreads();
  if (stamp) g.stamps[3] = __builtin_amdgcn_s_memtime();
  if (!m_ok) return;
  const int c_live = g.C_out - ct * CT < CT ? g.C_out - ct * CT : CT;
  double* __restrict__ dst = y + ((size_t)bi * g.C_out + (size_t)ct * CT) * pixels + pix;
  for (int cl = tid >> 7; cl < c_live; cl += 2) dst[(size_t)cl * pixels] = s_out[cl][m_s];
  if (stamp) g.stamps[4] = __builtin_amdgcn_s_memtime();
}

// ---------------------------------------------------------------------------
// many output channels (C4: 256): the gather of a K chunk is the expensive part of qconv_gemm_kernel<0>, and with one
// 32-channel tile per workgroup it is repeated by every channel tile.  Here a workgroup keeps kWideSub = 4 channel tiles
// (128 channels: 8 accumulator tiles per wavefront) per staged A chunk, so four times the MFMA work rides on every
// gather; K is staged 16 at a time to keep the LDS footprint.  Same operand layout (wide packing), same epilogue.
// ---------------------------------------------------------------------------
constexpr int kWideSub = 4;
constexpr int kWideK = 16;

template <bool UP>
__global__ __launch_bounds__(4 * kWave, UP ? 1 : 2) void qconv_gemm_wide_kernel(const double* __restrict__ x,
                                                                    const float* __restrict__ w,
                                                                    const float* __restrict__ padv,
                                                                    const double* __restrict__ bn,
                                                                    double* __restrict__ y, const GemmConv g) {
  constexpr int NBW = 64 * kWideSub;  // B columns per workgroup
  constexpr int KT = kWideK / 2;
  // two staging buffers: chunk k + 1 is written while chunk k feeds the matrix cores, ONE barrier per chunk
  constexpr int kStageBytes = (kWideK * kGemmM + kWideK * NBW) * 4;
  constexpr int kOutBytes = 32 * (kGemmM + 1) * 8;
  __shared__ __attribute__((aligned(16))) unsigned char s_raw[kOutBytes > 2 * kStageBytes ? kOutBytes : 2 * kStageBytes];
  __shared__ uint32_t s_tap[kGemmMaxK];
  __shared__ float s_n2[2][kGemmM];
  __shared__ float s_inv[kGemmM];
  double (*s_out)[kGemmM + 1] = reinterpret_cast<double (*)[kGemmM + 1]>(s_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t m0 = (int64_t)blockIdx.x * kGemmM;
  const int ct0 = blockIdx.y * kWideSub;  // first 32-channel tile of this workgroup
  const int khw = g.kh * g.kw;
  const size_t plane = UP ? (size_t)g.Hs * g.Ws : (size_t)g.H * g.W;
  for (int f = tid; f < g.K_pad; f += 4 * kWave) {
    const int c = f / khw, rem = f - c * khw;
    const int di = rem / g.kw, dj = rem - di * g.kw;
    const uint32_t off = UP ? (uint32_t)c : (uint32_t)(c * plane) + (uint32_t)(di * g.W + dj);
    s_tap[f] = off | ((uint32_t)di << 24) | ((uint32_t)dj << 28);
  }
  const int m_s = tid & (kGemmM - 1), kh_s = (tid >> 7) * KT;
  const int64_t mg = m0 + m_s;
  const bool m_ok = mg < g.M;
  const int64_t pixels = (int64_t)g.Ho * g.Wo;
  const int64_t bi = m_ok ? mg / pixels : 0;
  const int pix = m_ok ? (int)(mg - bi * pixels) : 0;
  const int oi = pix / g.Wo - g.ph, oj = pix % g.Wo - g.pw;
  const double* __restrict__ img = x + (size_t)bi * g.C * plane;
  const double* __restrict__ corner = img + (int64_t)oi * g.W + oj;
  float n2 = 0.f;
  __syncthreads();

  // The gather of an A chunk (strided float64 loads) is what the matrix cores used to wait for: with one chunk of
  // distance the loads of chunk c + 1 had ONE MFMA phase (~1.7 us) to land, and under the load of 512 workgroups they did
  // not (63 % of wave cycles waiting, matrix cores 58 % busy).  Now two register sets alternate: chunk c + 2 is requested
  // before the MFMAs of chunk c (two phases of flight).  B (L2-resident, contiguous rows) keeps one chunk of distance
  // but moves as 16-byte pieces -- 4 loads + 4 LDS stores per thread and chunk instead of 16 + 16 -- and is requested
  // BEFORE the far gather, so that waiting for it (the load counter retires in order) leaves the gather in flight.
  // (B by LDS-DMA was tried: the wait-count pass then drains vmcnt to zero in front of every LDS read behind a DMA.)
  using F4 = float __attribute__((ext_vector_type(4)));
  constexpr int kBV = (kWideK * NBW) / (4 * kWave) / 4;   // float4 pieces of B per thread and chunk
  double pv0[KT], pv1[KT];
  uint32_t mask0 = 0, mask1 = 0;   // bit u: tap u of that chunk lies inside the image
  F4 bv[kBV];
  auto fetch = [&](int k0, double (&pv)[KT], uint32_t& in_mask) {
    uint32_t tap[KT];
    in_mask = 0;
#pragma unroll
    for (int u = 0; u < KT; ++u) tap[u] = s_tap[k0 + kh_s + u];
#pragma unroll
    for (int u = 0; u < KT; ++u) {
      const int f = k0 + kh_s + u;
      const int ii = oi + (int)((tap[u] >> 24) & 15u), jj = oj + (int)(tap[u] >> 28);
      const uint32_t off = tap[u] & 0xFFFFFFu;
      const bool in = m_ok && f < g.F && ii >= 0 && ii < g.H && jj >= 0 && jj < g.W;
      // the select on the loaded value is deferred to stage(): done here it put a wait for the gather in front of
      // the chunk's MFMAs (s_waitcnt vmcnt right after the issue), i.e. the load latency on the critical path
      if constexpr (UP) {
        const double v = bilinear2x(img + (in ? (size_t)off * plane : 0), g.Hs, g.Ws, in ? ii : 0, in ? jj : 0);
        pv[u] = in ? v : 0.0;
        in_mask |= 1u << u;
      } else {
        pv[u] = *(in ? corner + off : img);
        in_mask |= (uint32_t)in << u;
      }
    }
  };
  // B chunk k0: wave w takes rows w, w + 4, ... -- one whole 1 KB row per wave instruction
  auto fetch_b = [&](int k0) {
#pragma unroll
    for (int i = 0; i < kBV; ++i)
      bv[i] = *reinterpret_cast<const F4*>(w + (size_t)(k0 + wave + 4 * i) * g.N_pad + ct0 * 64 + lane * 4);
  };
  auto stage = [&](int k0, int buf, const double (&pv)[KT], uint32_t in_mask) {
    float (*s_a)[kGemmM] = reinterpret_cast<float (*)[kGemmM]>(s_raw + buf * kStageBytes);
    float (*s_b)[NBW] = reinterpret_cast<float (*)[NBW]>(s_raw + buf * kStageBytes + kWideK * kGemmM * 4);
#pragma unroll
    for (int i = 0; i < kBV; ++i) *reinterpret_cast<F4*>(&s_b[wave + 4 * i][lane * 4]) = bv[i];
#pragma unroll
    for (int u = 0; u < KT; ++u) {
      const int f = k0 + kh_s + u;
      float v = 0.f;
      if (m_ok && f < g.F) {
        v = (float)((((in_mask >> u) & 1u) ? pv[u] : 0.0) + 0.1);
        n2 = fmaf(v, v, n2);
      }
      s_a[kh_s + u][m_s] = v;
    }
  };

  f32x16 acc_re[kWideSub], acc_im[kWideSub];
#pragma unroll
  for (int sub = 0; sub < kWideSub; ++sub)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      acc_re[sub][i] = 0.f;
      acc_im[sub][i] = 0.f;
    }
  int cur = 0;
  // one K chunk: request B of chunk c + 1 and A of chunk c + 2, run the MFMAs of chunk c, stage chunk c + 1
  auto chunk = [&](int k0, double (&pv_next)[KT], uint32_t& mask_next, double (&pv_far)[KT], uint32_t& mask_far) {
    const bool more = k0 + kWideK < g.K_pad;
    const int k_last = g.K_pad - kWideK;
    // (both requests unconditional -- past the end they repeat the last chunk and are never staged -- so that the wait
    //  counters behind them are exact on every path: a conditional request makes the pass assume the shorter queue)
    fetch_b(k0 + kWideK < k_last ? k0 + kWideK : k_last);
    __builtin_amdgcn_sched_barrier(0);   // B's loads are issued first: "B has landed" must not imply "the far gather has"
    fetch(k0 + 2 * kWideK < k_last ? k0 + 2 * kWideK : k_last, pv_far, mask_far);
    const float (*s_a)[kGemmM] = reinterpret_cast<const float (*)[kGemmM]>(s_raw + cur * kStageBytes);
    const float (*s_b)[NBW] = reinterpret_cast<const float (*)[NBW]>(s_raw + cur * kStageBytes + kWideK * kGemmM * 4);
    // operands of k-step i + 1 are read from LDS while the eight MFMAs of step i issue (two register sets): as one
    // read-then-use sequence the schedule was  wait - 2 MFMAs - read - wait - ...  and every pair of MFMAs (128 cycles
    // of matrix-core time) waited out an LDS round trip of about that length
    float fa[2], fb[2][2 * kWideSub];
    auto read_step = [&](int buf, int kk) {
      fa[buf] = s_a[kk + (lane >> 5)][wave * 32 + (lane & 31)];
#pragma unroll
      for (int sub = 0; sub < kWideSub; ++sub) {
        fb[buf][2 * sub] = s_b[kk + (lane >> 5)][sub * 64 + (lane & 31)];
        fb[buf][2 * sub + 1] = s_b[kk + (lane >> 5)][sub * 64 + 32 + (lane & 31)];
      }
    };
    read_step(0, 0);
#pragma unroll
    for (int kk = 0; kk < kWideK; kk += 2) {
      const int buf = (kk >> 1) & 1;
      if (kk + 2 < kWideK) read_step(buf ^ 1, kk + 2);
#pragma unroll
      for (int sub = 0; sub < kWideSub; ++sub) {
        acc_re[sub] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf], fb[buf][2 * sub], acc_re[sub], 0, 0, 0);
        acc_im[sub] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf], fb[buf][2 * sub + 1], acc_im[sub], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the reads of step i + 1 in front of the MFMAs of step i
    }
    if (more) stage(k0 + kWideK, cur ^ 1, pv_next, mask_next);
    __syncthreads();
    cur ^= 1;
  };
  fetch_b(0);
  fetch(0, pv0, mask0);
  fetch(kWideK < g.K_pad ? kWideK : 0, pv1, mask1);
  stage(0, 0, pv0, mask0);
  __syncthreads();
  int kc = 0;
  for (; kc + kWideK < g.K_pad; kc += 2 * kWideK) {
    chunk(kc, pv1, mask1, pv0, mask0);            // chunk c even: c + 1 waits in set 1, c + 2 goes to set 0
    chunk(kc + kWideK, pv0, mask0, pv1, mask1);
  }
  if (kc < g.K_pad) chunk(kc, pv1, mask1, pv0, mask0);
  s_n2[tid >> 7][m_s] = n2;
  __syncthreads();
  if (tid < kGemmM) s_inv[tid] = (float)(g.post_scale / ((double)s_n2[0][tid] + (double)s_n2[1][tid] + g.pad_norm2));
  __syncthreads();
#pragma unroll
  for (int sub = 0; sub < kWideSub; ++sub) {
    const int ct = ct0 + sub;
    const int col = lane & 31, c = ct * 32 + col;
    const bool live = c < g.C_out;
    const float pre = padv[ct * 64 + col], pim = padv[ct * 64 + 32 + col];
    const double bn_scale = (g.has_bn && live) ? bn[c] : 1.0, bn_shift = (g.has_bn && live) ? bn[g.bn_stride + c] : 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ml = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const float re = acc_re[sub][r] + pre, im = acc_im[sub][r] + pim;
      const float v = fminf(fmaxf((re * re + im * im) * s_inv[ml], 0.f), 1.f);
      s_out[col][ml] = (double)v * bn_scale + bn_shift;
    }
    __syncthreads();
    if (m_ok) {
      const int c_live = g.C_out - ct * 32 < 32 ? g.C_out - ct * 32 : 32;
      double* __restrict__ dst = y + ((size_t)bi * g.C_out + (size_t)ct * 32) * pixels + pix;
      for (int cl = tid >> 7; cl < c_live; cl += 2) dst[(size_t)cl * pixels] = s_out[cl][m_s];
    }
    __syncthreads();
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

// backward of the one-output-channel head in ONE pass over x and grad_y: gx[b, c, p] = w[c] gy[b, p] written while
// gw[c] += gy x and gb += gy accumulate in registers; per-workgroup partial sums [grid][c_in + 1] are summed in a fixed
// order by conv1x1_head_finalize_kernel (deterministic, no atomics).  (torch autograd ran five launches over the
// same tensors, one of them materialising x * w.)
constexpr int kHeadMaxC = 32;
// (CMAX: register arrays of 8 / 16 / 32 channels -- 8 keeps four workgroups per CU resident)
template <int CMAX>
__global__ __launch_bounds__(256) void conv1x1_head_backward_kernel(const double* __restrict__ x,
                                                                    const double* __restrict__ w,
                                                                    const double* __restrict__ gy, int64_t total,
                                                                    int64_t hw, int c_in, double* __restrict__ gx,
                                                                    double* __restrict__ partial) {
  __shared__ double s_red[4];
  double wv[CMAX], acc[CMAX + 1];   // acc[CMAX]: the bias gradient
#pragma unroll
  for (int c = 0; c < CMAX; ++c) {
    wv[c] = c < c_in ? w[c] : 0.0;
    acc[c] = 0.0;
  }
  acc[CMAX] = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / hw, px = i - b * hw;
    const size_t base = (size_t)b * c_in * hw + px;
    const double g = gy[i];
    acc[CMAX] += g;
    double xv[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) xv[c] = c < c_in ? x[base + (size_t)c * hw] : 0.0;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      if (c < c_in) {
        if (gx) gx[base + (size_t)c * hw] = wv[c] * g;
        acc[c] = fma(g, xv[c], acc[c]);
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* __restrict__ out = partial + (size_t)blockIdx.x * (c_in + 1);
#pragma unroll
  for (int c = 0; c <= CMAX; ++c) {
    if (c < c_in || c == CMAX) {   // (uniform)
      double v = acc[c];
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      __syncthreads();
      if (lane == 0) s_red[wave] = v;
      __syncthreads();
      if (threadIdx.x == 0) out[c == CMAX ? c_in : c] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    }
  }
}

__global__ __launch_bounds__(256) void conv1x1_head_finalize_kernel(const double* __restrict__ partial, int n_partials,
                                                                    int c_in, double* __restrict__ gw,
                                                                    double* __restrict__ gb) {
  // one workgroup: thread t adds partials t, t + 256, ... of a column, then the 256 sums are added in a fixed tree
  __shared__ double s_red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = 0; c <= c_in; ++c) {
    double v = 0.0;
    for (int p = threadIdx.x; p < n_partials; p += 256) v += partial[(size_t)p * (c_in + 1) + c];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      const double s = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
      if (c < c_in) {
        if (gw) gw[c] = s;
      } else if (gb) {
        gb[0] = s;
      }
    }
  }
}

}  // namespace qiddm
