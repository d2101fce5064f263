// qsim_qconv_train.h -- backward of the quantum convolution through the circuit's unitary.
//
// QConv2d's circuit does not depend on the data (reference nn/qconv.py:51-56: AmplitudeEmbedding, then
// StronglyEntanglingLayers(weights)), so with U = U(weights) and the embedded patch v^ = v / |v| (real) every output
// pixel m is  a_mc = sum_j U[2c, j] v^_mj,  y_mc = clamp(|a_mc|^2 D/2)  -- the GEMM the eval-mode route already runs.
// Its derivative needs no per-pixel circuit sweep either.  With t_mc = dL/dy_mc * D/2 where the clamp passes:
//
//   dL/dv^_mj = 2 Re sum_c t_mc conj(a_mc) U[2c, j]                 (a second product with the same rows of U)
//   dL/dv_mj  = (dL/dv^_mj - v^_mj * 2 sum_c t_mc |a_mc|^2) / |v_m|   (through the normalisation)
//   dL/dtheta = 2 Re sum_c  e_2c^T (dU/dtheta) h_c,    h_c[j] = sum_m t_mc conj(a_mc) v^_mj
//
// i.e. the weight gradient is the adjoint sweep of C_out vectors h_c (a third product, reduced over the pixels)
// instead of one sweep per output pixel.  This kernel computes all three products per tile of 64 pixels on the
// VALU (2 C_out <= 64 rows: the operand is thin); the h_c sweep is `wide_adjoint_kernel` in its raw mode.
//
// Tables: rt[(F + 1)][2 CO] float32 -- rt[j][c] = Re U[2c, j], rt[j][CO + c] = Im U[2c, j], row F = the pad
// columns' contribution 0.5 * sum_{j >= F} U[2c, j]; channels c >= C_out are zero.
// Outputs: gfeat_t[F][M] (feature gradients, TRANSPOSED so that both this kernel's stores and the fold's loads are
// coalesced) and hpart[grid][2 CO][F + 1] per-workgroup sums of W2_cc v^_j with W2 = (t Re a, t Im a)
// (h_c = hsum[c] - i hsum[CO + c]; column F = the value every pad column shares).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace qiddm {

struct TrainConv {
  int32_t C, H, W, kh, kw, ph, pw, Ho, Wo, C_out, F;
  int32_t groups;     // thread groups of the h product: each owns all F + 1 columns and 64 / groups pixels of a tile
  int64_t M;          // batch * Ho * Wo
  int64_t gy_bstride; // elements between two images of grad_y (C_out * Ho * Wo when dense; larger for a channel slice of a
                      // wider tensor, e.g. one half of a concatenation's gradient: read in place, no copy)
  float pad_norm2;    // 0.25 * (D - F)
  float post_scale;   // D / 2
  unsigned long long* stamps;  // diagnostics (QIDDM_STAMP_PTR): per-phase s_memtime sums of workgroup 0, else null
  // dL/dy handed in BEFORE the BatchNorm2d that follows the convolution (training mode): with bn_y = the convolution's
  // own output and bn_coef = [3][C_out] (qiddm_batchnorm_backward_stats), the kernels form
  //     dL/dy = coef[0][c] * grad_y + coef[1][c] * bn_y + coef[2][c]
  // while they load it -- the BatchNorm backward's transform pass (read 2, write 1 tensor) is never run.  Null: grad_y as is.
  const double* bn_y;
  const double* bn_coef;
  float* wpix;        // matrix-core kernel: per-pixel rows [M][2 CO] of 2 W2 / |v|, then [M] of dot / |v|^2, INSTEAD of
                      // gfeat_t (qsim_qconv_dx.h turns them into dL/dx); null: feature gradients + fold
};

constexpr int kTcWaves = 8;                  // wavefronts per workgroup: they split the feature columns of a tile
constexpr int kTcThreads = 64 * kTcWaves;
constexpr int kTcTile = 64;  // output pixels per tile: lane = pixel

__host__ __device__ inline int tc_v_stride(int F) { return (F + 1) | 1; }  // odd: conflict-free row writes
template <int CO>
__host__ __device__ inline size_t tc_lds_bytes(int F) {
  // rows table, v^ tile, W2 tile, cross-wave partials [waves][64][CO + 1], per-pixel scalars [2][waves][64], taps
  return ((size_t)(F + 1) * 2 * CO + (size_t)kTcTile * tc_v_stride(F) + (size_t)kTcTile * (2 * CO + 1) +
          (size_t)kTcWaves * kTcTile * (CO + 1) + (size_t)2 * kTcWaves * kTcTile) * sizeof(float) +
         (size_t)F * sizeof(uint32_t);
}

template <int CO, int JCH>
__global__ __launch_bounds__(kTcThreads, 1) void qconv_train_backward_kernel(const double* __restrict__ x,
                                                                          const double* __restrict__ gy,
                                                                          const float* __restrict__ rt,
                                                                          float* __restrict__ gfeat_t,
                                                                          float* __restrict__ hpart,
                                                                          const TrainConv tc) {
  constexpr int K2 = 2 * CO, WS = K2 + 1, PS = CO + 1, NW = kTcWaves, CQ = CO / NW;
  static_assert(CO % NW == 0, "every wavefront owns CO / waves output channels");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int F = tc.F, FS = tc_v_stride(F);
  float* s_rt = reinterpret_cast<float*>(smem_raw);  // [(F + 1)][K2]
  float* s_v = s_rt + (size_t)(F + 1) * K2;          // [kTcTile][FS]
  float* s_w = s_v + (size_t)kTcTile * FS;           // [kTcTile][WS]
  float* s_part = s_w + (size_t)kTcTile * WS;        // [NW][kTcTile][PS]
  float* s_sc = s_part + (size_t)NW * kTcTile * PS;  // [2][NW][kTcTile]: norm^2 partials, dot partials
  uint32_t* s_tap = reinterpret_cast<uint32_t*>(s_sc + 2 * NW * kTcTile);  // [F]: offset | di << 24 | dj << 28
  const int tid = threadIdx.x, q = tid >> 6, lane = tid & 63;
  for (int i = tid; i < (F + 1) * K2; i += kTcThreads) s_rt[i] = rt[i];
  for (int f = tid; f < F; f += kTcThreads) {
    const int dj = f % tc.kw, t = f / tc.kw;
    const int di = t % tc.kh, c = t / tc.kh;
    s_tap[f] = (uint32_t)((c * tc.H + di) * tc.W + dj) | ((uint32_t)di << 24) | ((uint32_t)dj << 28);
  }
  float acc[JCH][K2];
#pragma unroll
  for (int jc = 0; jc < JCH; ++jc)
#pragma unroll
    for (int cc = 0; cc < K2; ++cc) acc[jc][cc] = 0.f;
  // the h product: group hg of F + 1 consecutive threads owns pixels [hp_lo, hp_hi) of every tile, thread hj its column
  const int hg = tid / (F + 1), hj = tid - hg * (F + 1);
  const int hp_lo = kTcTile * hg / tc.groups, hp_hi = kTcTile * (hg + 1) / tc.groups;
  // this wave's share of the feature columns
  const int jq = (F + NW - 1) / NW;
  const int j_lo = q * jq < F ? q * jq : F, j_hi = (j_lo + jq < F) ? j_lo + jq : F;

  // BatchNorm coefficients of this thread's channels (TrainConv::bn_coef), once
  double bn_a[CQ], bn_b[CQ], bn_c[CQ];
#pragma unroll
  for (int i = 0; i < CQ; ++i) {
    const int c = q * CQ + i;
    const bool on = tc.bn_y != nullptr && c < tc.C_out;
    bn_a[i] = on ? tc.bn_coef[c] : 1.0;
    bn_b[i] = on ? tc.bn_coef[tc.C_out + c] : 0.0;
    bn_c[i] = on ? tc.bn_coef[2 * tc.C_out + c] : 0.0;
  }

  const int64_t pixels = (int64_t)tc.Ho * tc.Wo;
  const int64_t tiles = (tc.M + kTcTile - 1) / kTcTile;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t m_raw = tile * kTcTile + lane;
    const bool valid = m_raw < tc.M;
    const int64_t m = valid ? m_raw : tc.M - 1;
    const int64_t b = m / pixels;
    const int pix = (int)(m - b * pixels);
    const int oi = pix / tc.Wo, oj = pix - oi * tc.Wo;
    const int i0 = oi - tc.ph, j0 = oj - tc.pw;
    const double* __restrict__ img = x + (size_t)b * tc.C * tc.H * tc.W + (ptrdiff_t)i0 * tc.W + j0;
    auto feature = [&](int j) -> float {  // v_j = patch value + 0.1 (zero padding outside the image)
      const uint32_t tap = s_tap[j];
      const int ii = i0 + (int)((tap >> 24) & 15u), jj = j0 + (int)(tap >> 28);
      float v = 0.f;
      if (ii >= 0 && ii < tc.H && jj >= 0 && jj < tc.W) v = (float)img[tap & 0xffffffu];
      return v + 0.1f;
    };
    __syncthreads();  // the previous tile's readers of s_v / s_w / s_part are done (and the tables are filled)
    // ---- partial a = U_rows v over this wave's columns; the features are parked (unnormalised) in s_v ---------------
    {
      float a[K2];
#pragma unroll
      for (int cc = 0; cc < K2; ++cc) a[cc] = 0.f;
      float n2 = 0.f;
      for (int j = j_lo; j < j_hi; ++j) {
        const float v = feature(j);
        s_v[lane * FS + j] = v;
        n2 = fmaf(v, v, n2);
        const float* __restrict__ r = s_rt + (size_t)j * K2;
#pragma unroll
        for (int cc = 0; cc < K2; ++cc) a[cc] = fmaf(r[cc], v, a[cc]);
      }
      s_sc[q * kTcTile + lane] = n2;
      // cross-wave sum in two rounds (real parts, then imaginary parts) through [4][64][CO + 1]
      float mine[2 * CQ];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int c = 0; c < CO; ++c) s_part[((size_t)q * kTcTile + lane) * PS + c] = a[half * CO + c];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < CQ; ++i) {
          float t = 0.f;
#pragma unroll
          for (int w = 0; w < NW; ++w) t += s_part[((size_t)w * kTcTile + lane) * PS + q * CQ + i];
          mine[half * CQ + i] = t;
        }
        __syncthreads();
      }
      float nrm2 = tc.pad_norm2;
#pragma unroll
      for (int w = 0; w < NW; ++w) nrm2 += s_sc[w * kTcTile + lane];
      const float inv = 1.0f / sqrtf(nrm2);
      // ---- this thread's channels: a, t, W2 = t (Re a, Im a) -> s_w; its share of dot = 2 sum t |a|^2 ------------------
      const double* __restrict__ gpix = gy + (size_t)b * tc.gy_bstride + pix;
      const float* __restrict__ rp = s_rt + (size_t)F * K2;
      float dotp = 0.f;
#pragma unroll
      for (int i = 0; i < CQ; ++i) {
        const int c = q * CQ + i;
        const float ar = (mine[i] + rp[c]) * inv, ai = (mine[CQ + i] + rp[CO + c]) * inv;
        const float p2 = ar * ar + ai * ai;
        float t = 0.f;
        if (valid && c < tc.C_out && p2 * tc.post_scale <= 1.0f) {
          double g = gpix[(size_t)c * pixels];
          if (tc.bn_y)
            g = fma(bn_a[i], g, fma(bn_b[i], tc.bn_y[(size_t)b * tc.C_out * pixels + pix + (size_t)c * pixels], bn_c[i]));
          t = (float)g * tc.post_scale;
        }
        dotp = fmaf(2.0f * t, p2, dotp);
        s_w[lane * WS + c] = t * ar;
        s_w[lane * WS + CO + c] = t * ai;
      }
      s_sc[(NW + q) * kTcTile + lane] = dotp;
      __syncthreads();
      // ---- feature gradients over this wave's columns; s_v becomes v^ ------------------------------------------------
      float dot = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) dot += s_sc[(NW + w) * kTcTile + lane];
#pragma unroll
      for (int cc = 0; cc < K2; ++cc) a[cc] = s_w[lane * WS + cc];
      for (int j = j_lo; j < j_hi; ++j) {
        const float vh = s_v[lane * FS + j] * inv;
        s_v[lane * FS + j] = vh;
        const float* __restrict__ r = s_rt + (size_t)j * K2;
        float sacc = 0.f;
#pragma unroll
        for (int cc = 0; cc < K2; ++cc) sacc = fmaf(a[cc], r[cc], sacc);
        if (valid) gfeat_t[(size_t)j * tc.M + m] = (2.0f * sacc - vh * dot) * inv;
      }
      if (q == NW - 1) s_v[lane * FS + F] = 0.5f * inv;  // the value every pad column holds
    }
    __syncthreads();
    // ---- h += W2^T v^ over the tile: thread = (pixel group, feature column) ------------------------------------------
    if (hg < tc.groups) {
      for (int p = hp_lo; p < hp_hi; ++p) {
        const float v = s_v[p * FS + hj];
        const float* __restrict__ w = s_w + p * WS;
#pragma unroll
        for (int cc = 0; cc < K2; ++cc) acc[0][cc] = fmaf(w[cc], v, acc[0][cc]);
      }
    }
  }
  // the groups' sums meet in LDS (eight rows at a time through the v^ tile's space): one slab per workgroup
  float* __restrict__ hp = hpart + (size_t)blockIdx.x * K2 * (F + 1);
#pragma unroll
  for (int c0 = 0; c0 < K2; c0 += 8) {
    __syncthreads();
    if (hg < tc.groups) {
#pragma unroll
      for (int i = 0; i < 8; ++i) s_v[((size_t)hg * 8 + i) * (F + 1) + hj] = acc[0][c0 + i];
    }
    __syncthreads();
    if (hg == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float t = 0.f;
        for (int g = 0; g < tc.groups; ++g) t += s_v[((size_t)g * 8 + i) * (F + 1) + hj];
        hp[(size_t)(c0 + i) * (F + 1) + hj] = t;
      }
    }
  }
}

// rows table of the backward from the circuit unitary u (D x D complex128; u[k][j] = <k|U|j>, or its transpose):
// rt[j][c] = Re U[2c, j], rt[j][CO + c] = Im U[2c, j] for j < F, row F = 0.5 * sum_{j >= F} U[2c, j]
__global__ __launch_bounds__(256) void qconv_rows_kernel(const double* __restrict__ u, int transposed, int D, int F,
                                                         int C_out, int CO, float* __restrict__ rt) {
  // one workgroup per channel c (< CO): columns j < F copied, then the pad columns summed by the whole workgroup
  __shared__ double s_re[4], s_im[4];
  const int c = blockIdx.x, tid = threadIdx.x;
  auto elem = [&](int col, double& r, double& i) {
    const size_t at = transposed ? ((size_t)col * D + 2 * c) : ((size_t)2 * c * D + col);
    r = u[2 * at];
    i = u[2 * at + 1];
  };
  for (int j = tid; j < F; j += 256) {
    double re = 0, im = 0;
    if (c < C_out) elem(j, re, im);
    rt[(size_t)j * 2 * CO + c] = (float)re;
    rt[(size_t)j * 2 * CO + CO + c] = (float)im;
  }
  double re = 0, im = 0;
  if (c < C_out) {
    for (int col = F + tid; col < D; col += 256) {
      double r, i;
      elem(col, r, i);
      re += r;
      im += i;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    re += __shfl_down(re, off, 64);
    im += __shfl_down(im, off, 64);
  }
  if ((tid & 63) == 0) {
    s_re[tid >> 6] = re;
    s_im[tid >> 6] = im;
  }
  __syncthreads();
  if (tid == 0) {
    rt[(size_t)F * 2 * CO + c] = (float)(0.5 * (s_re[0] + s_re[1] + s_re[2] + s_re[3]));
    rt[(size_t)F * 2 * CO + CO + c] = (float)(0.5 * (s_im[0] + s_im[1] + s_im[2] + s_im[3]));
  }
}

// h_c from the per-workgroup partial sums, laid out as the start vectors of the matrix-element sweep, and the
// matching lambda = e_2c:  psi0[c][j] = sum_p hp[p][c][min(j, F)] - i sum_p hp[p][CO + c][min(j, F)]
__global__ __launch_bounds__(256) void qconv_vectors_kernel(const float* __restrict__ hpart, int n_partials, int D,
                                                            int F, int C_out, int CO, double* __restrict__ psi0,
                                                            double* __restrict__ lambda) {
  // workgroup = 8 columns x 32 slices of the partial slabs; blockIdx.y = channel
  __shared__ double s_re[32][9], s_im[32][9];
  const int jl = threadIdx.x & 7, pg = threadIdx.x >> 3;
  const int j = blockIdx.x * 8 + jl, c = blockIdx.y;
  const size_t slab = (size_t)2 * CO * (F + 1);
  double re = 0, im = 0;
  if (j <= F) {
    for (int p = pg; p < n_partials; p += 32) {
      re += (double)hpart[p * slab + (size_t)c * (F + 1) + j];
      im -= (double)hpart[p * slab + (size_t)(CO + c) * (F + 1) + j];
    }
  }
  s_re[pg][jl] = re;
  s_im[pg][jl] = im;
  __syncthreads();
  if (pg != 0 || j > F) return;
  for (int g = 1; g < 32; ++g) {
    re += s_re[g][jl];
    im += s_im[g][jl];
  }
  double* row = psi0 + (size_t)c * 2 * D;
  double* lrow = lambda + (size_t)c * 2 * D;
  const int j_end = j < F ? j + 1 : D;  // column F stands for every pad column
  for (int col = j; col < j_end; ++col) {
    row[2 * col] = re;
    row[2 * col + 1] = im;
    lrow[2 * col] = col == 2 * c ? 1.0 : 0.0;
    lrow[2 * col + 1] = 0.0;
  }
}

// dL/dx from the transposed feature gradients: every input element gathers the kh*kw patch entries it appeared in
// (fixed order; neighbouring threads read neighbouring pixels of the same feature row: coalesced).  One 64-bit
// division per workgroup, 32-bit ones per element.  KH, KW > 0: the taps unrolled,
// every load issued (to a clamped address) before the first add -- out-of-range taps add 0.0, which leaves the sum as the
// branchy loop has it; 0: run-time extents.
template <int KH, int KW>
__global__ __launch_bounds__(256) void qconv_fold_t_kernel(const float* __restrict__ gfeat_t, double* __restrict__ gx,
                                                           int64_t total, const TrainConv tc) {
  const int hw = tc.H * tc.W;
  const int kh = KH > 0 ? KH : tc.kh, kw = KW > 0 ? KW : tc.kw;
  // element -> (plane, i, j): one 64-bit division per workgroup (uniform), 32-bit ones per thread
  const int64_t base = (int64_t)blockIdx.x * blockDim.x;
  const int64_t pl0 = base / hw;
  const uint32_t q = (uint32_t)(base - pl0 * hw) + threadIdx.x;
  {
    const int64_t pl = pl0 + q / (uint32_t)hw;
    const int p = (int)(q % (uint32_t)hw);
    if (pl * hw + p >= total) return;
    const int i = p / tc.W, j = p - i * tc.W;
    const int64_t b = pl / tc.C;
    const int c = (int)(pl - b * tc.C);
    const float* __restrict__ src = gfeat_t + (size_t)c * kh * kw * tc.M + (size_t)b * tc.Ho * tc.Wo;
    double accv = 0;
    if constexpr (KH > 0 && KW > 0) {
      float v[KH * KW];
#pragma unroll
      for (int di = 0; di < KH; ++di) {
        const int oi = i - di + tc.ph;
        const int oic = min(max(oi, 0), tc.Ho - 1);
#pragma unroll
        for (int dj = 0; dj < KW; ++dj) {
          const int oj = j - dj + tc.pw;
          const int ojc = min(max(oj, 0), tc.Wo - 1);
          const float ld = src[(size_t)(di * KW + dj) * tc.M + oic * tc.Wo + ojc];
          v[di * KW + dj] = (oi == oic && oj == ojc) ? ld : 0.f;
        }
      }
#pragma unroll
      for (int t = 0; t < KH * KW; ++t) accv += (double)v[t];
    } else {
      for (int di = 0; di < kh; ++di) {
        const int oi = i - di + tc.ph;
        if (oi < 0 || oi >= tc.Ho) continue;
        for (int dj = 0; dj < kw; ++dj) {
          const int oj = j - dj + tc.pw;
          if (oj < 0 || oj >= tc.Wo) continue;
          accv += (double)src[(size_t)(di * kw + dj) * tc.M + oi * tc.Wo + oj];
        }
      }
    }
    gx[pl * hw + p] = accv;
  }
}

// host side of the launch (both entry points that fold)
inline hipError_t launch_fold_t(const float* gfeat_t, double* gx, int64_t batch, const TrainConv& tc, hipStream_t st) {
  const int64_t total = batch * tc.C * tc.H * tc.W;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (tc.kh == 3 && tc.kw == 3)
    hipLaunchKernelGGL((qconv_fold_t_kernel<3, 3>), grid, dim3(256), 0, st, gfeat_t, gx, total, tc);
  else if (tc.kh == 1 && tc.kw == 1)
    hipLaunchKernelGGL((qconv_fold_t_kernel<1, 1>), grid, dim3(256), 0, st, gfeat_t, gx, total, tc);
  else
    hipLaunchKernelGGL((qconv_fold_t_kernel<0, 0>), grid, dim3(256), 0, st, gfeat_t, gx, total, tc);
  return hipGetLastError();
}

}  // namespace qiddm
