// C ABI of the training-mode BatchNorm2d that follows every quantum convolution of `unet_simple`
// (reference nn/unet_simple.py:9-18, 30-39: `net = [QConv2d, BatchNorm2d]`).
//
// torch has no float64 library path for it on ROCm and runs three generic launches forward (statistics,
// running-stat update, transform) and a slow one backward; here each direction is two launches over a
// (channel, batch-slice) grid: partial sums, then every workgroup re-reduces its channel's partials in a fixed
// order (deterministic, no atomics) and transforms its slice.
#include "capi_common.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

using qiddm_capi::fail;

namespace qiddm {

constexpr int kNormThreads = 256;
constexpr int kNormMaxSlices = 64;

struct NormGeom {
  int64_t batch, hw;
  int32_t channels, slices;
};

__device__ __forceinline__ double block_sum(double v, double* s_red) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  double t = 0;
  for (int w = 0; w < kNormThreads / 64; ++w) t += s_red[w];
  return t;
}

// A slice's elements -- rows b0 .. b1 of one channel, hw contiguous values each -- are dealt to the threads as ONE flat
// range (thread t takes elements t, t + 256, ...), eight loads issued before the first use: a 7 x 7 plane keeps all
// 256 threads busy, and a workgroup has ~16 KB in flight instead of ~4 (the row-by-row walk read at 1.6 TB/s).
constexpr int kNormUnroll = 8;
struct SliceWalk {
  int64_t off;       // element offset of the current element in the (batch, channels, hw) tensor
  int i, di, hw;     // position in the row; per-step advance
  int64_t drow, dstep;
  __device__ __forceinline__ SliceWalk(const int64_t b0, const int c, const int channels, const int64_t hw_) {
    hw = (int)hw_;
    const int t = threadIdx.x;
    i = t % hw;
    off = ((b0 + t / hw) * channels + c) * hw_ + i;
    di = kNormThreads % hw;
    drow = (int64_t)channels * hw_;                       // next batch row of the same channel
    dstep = (int64_t)(kNormThreads / hw) * drow + di;     // kNormThreads elements further, without a wrap
  }
  __device__ __forceinline__ void step() {
    off += dstep;
    i += di;
    if (i >= hw) {       // wrapped past the row's end: one more row, hw back
      i -= hw;
      off += drow - hw;
    }
  }
};

// batch rows [b0, b1) of slice s
__device__ __forceinline__ void slice_range(const NormGeom& g, int s, int64_t& b0, int64_t& b1) {
  b0 = g.batch * s / g.slices;
  b1 = g.batch * (s + 1) / g.slices;
}

// forward pass 1: per (channel, slice) sums of d = x - pivot and d^2, pivot = the channel's first element
// (keeps the variance free of the E[x^2] - E[x]^2 cancellation)
__global__ __launch_bounds__(kNormThreads) void bn_stats_kernel(const double* __restrict__ x, const NormGeom g,
                                                                double* __restrict__ partial) {
  __shared__ double s_red[kNormThreads / 64];
  const int c = blockIdx.x, s = blockIdx.y;
  int64_t b0, b1;
  slice_range(g, s, b0, b1);
  const double pivot = x[(int64_t)c * g.hw];
  double sum = 0, sq = 0;
  const int64_t count = (b1 - b0) * g.hw;
  SliceWalk w(b0, c, g.channels, g.hw);
  for (int64_t e = threadIdx.x; e < count; e += (int64_t)kNormThreads * kNormUnroll) {
    double v[kNormUnroll];
#pragma unroll
    for (int u = 0; u < kNormUnroll; ++u) {
      v[u] = pivot;
      if (e + (int64_t)u * kNormThreads < count) v[u] = x[w.off];
      w.step();
    }
#pragma unroll
    for (int u = 0; u < kNormUnroll; ++u) {
      const double d = v[u] - pivot;
      sum += d;
      sq = fma(d, d, sq);
    }
  }
  sum = block_sum(sum, s_red);
  sq = block_sum(sq, s_red);
  if (threadIdx.x == 0) {
    partial[((int64_t)c * g.slices + s) * 2 + 0] = sum;
    partial[((int64_t)c * g.slices + s) * 2 + 1] = sq;
  }
}

// forward pass 2: y = (x - mean) * invstd * weight + bias; slice 0 of a channel also stores mean / invstd and
// moves the running statistics (unbiased variance, as torch does)
__global__ __launch_bounds__(kNormThreads) void bn_apply_kernel(const double* __restrict__ x, const NormGeom g,
                                                                const double* __restrict__ partial,
                                                                const double* __restrict__ weight,
                                                                const double* __restrict__ bias,
                                                                double* __restrict__ running_mean,
                                                                double* __restrict__ running_var, double momentum,
                                                                double eps, double* __restrict__ y,
                                                                double* __restrict__ save_mean,
                                                                double* __restrict__ save_invstd) {
  const int c = blockIdx.x, s = blockIdx.y;
  double sum = 0, sq = 0;
  for (int i = 0; i < g.slices; ++i) {
    sum += partial[((int64_t)c * g.slices + i) * 2 + 0];
    sq += partial[((int64_t)c * g.slices + i) * 2 + 1];
  }
  const double n = (double)(g.batch * g.hw);
  const double pivot = x[(int64_t)c * g.hw];
  const double dmean = sum / n;
  const double mean = pivot + dmean;
  double var = (sq - sum * dmean) / n;
  var = var < 0 ? 0 : var;
  const double invstd = 1.0 / sqrt(var + eps);
  if (s == 0 && threadIdx.x == 0) {
    save_mean[c] = mean;
    save_invstd[c] = invstd;
    if (running_mean) running_mean[c] = (1.0 - momentum) * running_mean[c] + momentum * mean;
    if (running_var) {
      const double unbiased = n > 1 ? var * n / (n - 1.0) : var;
      running_var[c] = (1.0 - momentum) * running_var[c] + momentum * unbiased;
    }
  }
  const double scale = invstd * (weight ? weight[c] : 1.0);
  const double shift = bias ? bias[c] : 0.0;
  int64_t b0, b1;
  slice_range(g, s, b0, b1);
  const int64_t count = (b1 - b0) * g.hw;
  SliceWalk w(b0, c, g.channels, g.hw);
  for (int64_t e = threadIdx.x; e < count; e += (int64_t)kNormThreads * kNormUnroll) {
    double v[kNormUnroll];
    int64_t at[kNormUnroll];
#pragma unroll
    for (int u = 0; u < kNormUnroll; ++u) {
      at[u] = e + (int64_t)u * kNormThreads < count ? w.off : -1;
      v[u] = at[u] >= 0 ? x[at[u]] : 0.0;
      w.step();
    }
#pragma unroll
    for (int u = 0; u < kNormUnroll; ++u)
      if (at[u] >= 0) y[at[u]] = fma(v[u] - mean, scale, shift);
  }
}

// backward pass 1: per (channel, slice) sums of g and g * xhat
__global__ __launch_bounds__(kNormThreads) void bn_back_stats_kernel(const double* __restrict__ x,
                                                                     const double* __restrict__ gy, const NormGeom g,
                                                                     const double* __restrict__ save_mean,
                                                                     const double* __restrict__ save_invstd,
                                                                     double* __restrict__ partial) {
  __shared__ double s_red[kNormThreads / 64];
  const int c = blockIdx.x, s = blockIdx.y;
  int64_t b0, b1;
  slice_range(g, s, b0, b1);
  const double mean = save_mean[c], invstd = save_invstd[c];
  double sg = 0, sgx = 0;
  const int64_t count = (b1 - b0) * g.hw;
  SliceWalk w(b0, c, g.channels, g.hw);
  for (int64_t e = threadIdx.x; e < count; e += (int64_t)kNormThreads * kNormUnroll) {
    double gv[kNormUnroll], xv[kNormUnroll];
#pragma unroll
    for (int u = 0; u < kNormUnroll; ++u) {
      const bool in = e + (int64_t)u * kNormThreads < count;
      gv[u] = in ? gy[w.off] : 0.0;
      xv[u] = in ? x[w.off] : mean;
      w.step();
    }
#pragma unroll
    for (int u = 0; u < kNormUnroll; ++u) {
      sg += gv[u];
      sgx = fma(gv[u], (xv[u] - mean) * invstd, sgx);
    }
  }
  sg = block_sum(sg, s_red);
  sgx = block_sum(sgx, s_red);
  if (threadIdx.x == 0) {
    partial[((int64_t)c * g.slices + s) * 2 + 0] = sg;
    partial[((int64_t)c * g.slices + s) * 2 + 1] = sgx;
  }
}

// backward pass 2: dx = weight * invstd * (g - mean(g) - xhat * mean(g * xhat)); dweight = sum g xhat, dbias = sum g
__global__ __launch_bounds__(kNormThreads) void bn_back_apply_kernel(const double* __restrict__ x,
                                                                     const double* __restrict__ gy, const NormGeom g,
                                                                     const double* __restrict__ partial,
                                                                     const double* __restrict__ weight,
                                                                     const double* __restrict__ save_mean,
                                                                     const double* __restrict__ save_invstd,
                                                                     double* __restrict__ gx,
                                                                     double* __restrict__ gweight,
                                                                     double* __restrict__ gbias) {
  const int c = blockIdx.x, s = blockIdx.y;
  double sg = 0, sgx = 0;
  for (int i = 0; i < g.slices; ++i) {
    sg += partial[((int64_t)c * g.slices + i) * 2 + 0];
    sgx += partial[((int64_t)c * g.slices + i) * 2 + 1];
  }
  if (s == 0 && threadIdx.x == 0) {
    if (gweight) gweight[c] = sgx;
    if (gbias) gbias[c] = sg;
  }
  if (!gx) return;
  const double n = (double)(g.batch * g.hw);
  const double mean = save_mean[c], invstd = save_invstd[c];
  const double k = invstd * (weight ? weight[c] : 1.0);
  const double mg = sg / n, mgx = sgx / n;
  int64_t b0, b1;
  slice_range(g, s, b0, b1);
  const int64_t count = (b1 - b0) * g.hw;
  SliceWalk w(b0, c, g.channels, g.hw);
  for (int64_t e = threadIdx.x; e < count; e += (int64_t)kNormThreads * kNormUnroll) {
    double gv[kNormUnroll], xv[kNormUnroll];
    int64_t at[kNormUnroll];
#pragma unroll
    for (int u = 0; u < kNormUnroll; ++u) {
      at[u] = e + (int64_t)u * kNormThreads < count ? w.off : -1;
      gv[u] = at[u] >= 0 ? gy[at[u]] : 0.0;
      xv[u] = at[u] >= 0 ? x[at[u]] : 0.0;
      w.step();
    }
#pragma unroll
    for (int u = 0; u < kNormUnroll; ++u) {
      const double xh = (xv[u] - mean) * invstd;
      if (at[u] >= 0) gx[at[u]] = k * (gv[u] - mg - xh * mgx);
    }
  }
}

// backward pass 2 without the transform: dweight, dbias and the three per-channel coefficients with which a consumer
// forms dL/dx itself,  dx = coef[0][c] gy + coef[1][c] x + coef[2][c]  (the same  weight * invstd * (g - mean(g) -
// xhat * mean(g * xhat))  multiplied out) -- the quantum convolution in front of the BatchNorm applies it while it loads
// its dL/dy (qsim_qconv_train.h: TrainConv::bn_coef)
__global__ __launch_bounds__(64) void bn_back_coef_kernel(const NormGeom g, const double* __restrict__ partial,
                                                          const double* __restrict__ weight,
                                                          const double* __restrict__ save_mean,
                                                          const double* __restrict__ save_invstd,
                                                          double* __restrict__ gweight, double* __restrict__ gbias,
                                                          double* __restrict__ coef) {
  for (int c = threadIdx.x; c < g.channels; c += 64) {
    double sg = 0, sgx = 0;
    for (int i = 0; i < g.slices; ++i) {
      sg += partial[((int64_t)c * g.slices + i) * 2 + 0];
      sgx += partial[((int64_t)c * g.slices + i) * 2 + 1];
    }
    if (gweight) gweight[c] = sgx;
    if (gbias) gbias[c] = sg;
    const double n = (double)(g.batch * g.hw);
    const double mean = save_mean[c], invstd = save_invstd[c];
    const double k = invstd * (weight ? weight[c] : 1.0);
    const double mg = sg / n, mgx = sgx / n;
    coef[c] = k;
    coef[g.channels + c] = -k * invstd * mgx;
    coef[2 * g.channels + c] = k * (invstd * mgx * mean - mg);
  }
}

// ---- bilinear x2 (the `Upsample(scale_factor=2, mode="bilinear")` in front of every `up_conv`, reference
// nn/unet_simple.py:40-49) with the interpolation weights handed in as the two dense 1-D matrices ah (2H x H) and
// aw (2W x W) -- the caller takes them from torch's own operator, so the numbers are torch's; row o of such a matrix
// is non-zero only in columns o/2 - 1 .. o/2 + 1, column i only in rows 2i - 1 .. 2i + 2.
__global__ __launch_bounds__(256) void upsample2x_forward_kernel(const double* __restrict__ x,
                                                                 const double* __restrict__ ah,
                                                                 const double* __restrict__ aw, int64_t planes, int H,
                                                                 int W, double* __restrict__ y) {
  // one thread per INPUT pixel (i, j): the 2 x 2 output block (2i.., 2j..) from the 3 x 3 patch around it
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= planes * H * W) return;
  const int j = (int)(idx % W);
  const int i = (int)((idx / W) % H);
  const int64_t plane = idx / ((int64_t)W * H);
  const double* __restrict__ src = x + plane * H * W;
  const int Wo = 2 * W;
  // rows 2i (columns i-1, i) and 2i+1 (columns i, i+1) of ah; out-of-range neighbours carry weight 0
  const int im = i > 0 ? i - 1 : i, ip = i < H - 1 ? i + 1 : i;
  const int jm = j > 0 ? j - 1 : j, jp = j < W - 1 ? j + 1 : j;
  const double h0m = i > 0 ? ah[(size_t)(2 * i) * H + im] : 0.0, h0c = ah[(size_t)(2 * i) * H + i];
  const double h1c = ah[(size_t)(2 * i + 1) * H + i], h1p = i < H - 1 ? ah[(size_t)(2 * i + 1) * H + ip] : 0.0;
  const double w0m = j > 0 ? aw[(size_t)(2 * j) * W + jm] : 0.0, w0c = aw[(size_t)(2 * j) * W + j];
  const double w1c = aw[(size_t)(2 * j + 1) * W + j], w1p = j < W - 1 ? aw[(size_t)(2 * j + 1) * W + jp] : 0.0;
  double r0[3], r1[3], r2[3];  // patch rows i-1, i, i+1 at columns j-1, j, j+1
  r0[0] = src[(size_t)im * W + jm]; r0[1] = src[(size_t)im * W + j]; r0[2] = src[(size_t)im * W + jp];
  r1[0] = src[(size_t)i * W + jm];  r1[1] = src[(size_t)i * W + j];  r1[2] = src[(size_t)i * W + jp];
  r2[0] = src[(size_t)ip * W + jm]; r2[1] = src[(size_t)ip * W + j]; r2[2] = src[(size_t)ip * W + jp];
  // horizontal pass per patch row: output columns 2j and 2j+1
  const double a0 = fma(w0m, r0[0], w0c * r0[1]), b0 = fma(w1p, r0[2], w1c * r0[1]);
  const double a1 = fma(w0m, r1[0], w0c * r1[1]), b1 = fma(w1p, r1[2], w1c * r1[1]);
  const double a2 = fma(w0m, r2[0], w0c * r2[1]), b2 = fma(w1p, r2[2], w1c * r2[1]);
  double* __restrict__ dst = y + plane * 4 * H * W + (size_t)(2 * i) * Wo + 2 * j;
  dst[0] = fma(h0m, a0, h0c * a1);
  dst[1] = fma(h0m, b0, h0c * b1);
  dst[Wo] = fma(h1p, a2, h1c * a1);
  dst[Wo + 1] = fma(h1p, b2, h1c * b1);
}

__global__ __launch_bounds__(256) void upsample2x_backward_kernel(const double* __restrict__ gy,
                                                                  const double* __restrict__ ah,
                                                                  const double* __restrict__ aw, int64_t planes, int H,
                                                                  int W, double* __restrict__ gx) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= planes * H * W) return;
  const int j = (int)(idx % W);
  const int i = (int)((idx / W) % H);
  const int64_t plane = idx / ((int64_t)W * H);
  const int Ho = 2 * H, Wo = 2 * W;
  const double* __restrict__ src = gy + plane * Ho * Wo;
  double acc = 0;
  for (int o = 2 * i - 1; o <= 2 * i + 2; ++o) {
    if (o < 0 || o >= Ho) continue;
    const double wh = ah[(size_t)o * H + i];
    if (wh == 0.0) continue;
    double row = 0;
    for (int p = 2 * j - 1; p <= 2 * j + 2; ++p) {
      if (p < 0 || p >= Wo) continue;
      row = fma(aw[(size_t)p * W + j], src[(size_t)o * Wo + p], row);
    }
    acc = fma(wh, row, acc);
  }
  gx[idx] = acc;
}

// the same gather with the 2H x 2W plane staged in LDS first: every gy element is read from memory once, coalesced
// (the one-thread-per-output kernel above fetches each one four times, strided by two: 1.3 TB/s at 28 x 28), then each
// thread makes outputs of the plane from the staged copy.  One plane per trip; planes up to 4096 elements.
constexpr int kUpsMaxPlane = 4096;
__host__ __device__ inline int ups_planes_per_trip(int big) { return big >= 2048 ? 1 : 2048 / big; }
__global__ __launch_bounds__(256) void upsample2x_backward_lds_kernel(const double* __restrict__ gy,
                                                                      const double* __restrict__ ah,
                                                                      const double* __restrict__ aw, int64_t planes,
                                                                      int H, int W, double* __restrict__ gx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ups_smem[];
  double* s_p = reinterpret_cast<double*>(ups_smem);   // [planes per trip][2H][2W]
  const int Ho = 2 * H, Wo = 2 * W, big = Ho * Wo, small = H * W;
  const int pp = ups_planes_per_trip(big);             // consecutive planes of a trip: one contiguous range of gy
  // the four interpolation weights of every output row / column (rows 2i - 1 .. 2i + 2 of ah's column i; zero outside
  // the plane), staged once: the gather below then touches LDS only (read from ah / aw in place, every output waited
  // for ~20 dependent global loads)
  double* s_wh = s_p + (size_t)pp * big;               // [H][4]
  double* s_ww = s_wh + 4 * H;                         // [W][4]
  const int tid = threadIdx.x;
  for (int e = tid; e < 4 * H; e += 256) {
    const int i = e >> 2, o = 2 * i - 1 + (e & 3);
    s_wh[e] = (o >= 0 && o < Ho) ? ah[(size_t)o * H + i] : 0.0;
  }
  for (int e = tid; e < 4 * W; e += 256) {
    const int j = e >> 2, p = 2 * j - 1 + (e & 3);
    s_ww[e] = (p >= 0 && p < Wo) ? aw[(size_t)p * W + j] : 0.0;
  }
  const int64_t trips = (planes + pp - 1) / pp;
  for (int64_t trip = blockIdx.x; trip < trips; trip += gridDim.x) {
    const int64_t first = trip * pp;
    const int n = (int)(planes - first < pp ? planes - first : pp);
    const double* __restrict__ src = gy + first * big;
    __syncthreads();   // the previous trip's readers are done (and the weights are staged)
    for (int e = tid; e < n * big; e += 256) s_p[e] = src[e];
    __syncthreads();
    for (int e = tid; e < n * small; e += 256) {
      const int pl = e / small, r = e - pl * small;
      const int i = r / W, j = r - i * W;
      const double* __restrict__ sp = s_p + pl * big;
      double ww[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) ww[k] = s_ww[4 * j + k];
      double acc = 0;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int o = 2 * i - 1 + a;
        const double wh = s_wh[4 * i + a];
        if (o < 0 || o >= Ho || wh == 0.0) continue;
        double row = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int p = 2 * j - 1 + k;
          if (p < 0 || p >= Wo) continue;
          row = fma(ww[k], sp[o * Wo + p], row);
        }
        acc = fma(wh, row, acc);
      }
      gx[first * small + e] = acc;
    }
  }
}

// ---- MaxPool2d(kernel 2, stride 2) of the down blocks (reference nn/unet.py:93-95), float64, floor mode -----------------
// forward: one thread per output element; backward: one thread per output element recomputes which of its four inputs
// won (the first maximum in row-major window order, as torch's kernel picks it; a NaN wins) and writes all four input
// gradients -- no index tensor (torch keeps one int64 per output), every element of grad_x written exactly once.
__global__ __launch_bounds__(256) void maxpool2_forward_kernel(const double* __restrict__ x, int64_t planes, int H, int W,
                                                               double* __restrict__ y) {
  const int Ho = H / 2, Wo = W / 2;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= planes * Ho * Wo) return;
  const int j = (int)(idx % Wo);
  const int i = (int)((idx / Wo) % Ho);
  const int64_t plane = idx / ((int64_t)Wo * Ho);
  const double* __restrict__ src = x + plane * H * W + (size_t)(2 * i) * W + 2 * j;
  const double v[4] = {src[0], src[1], src[W], src[W + 1]};
  double m = v[0];
#pragma unroll
  for (int k = 1; k < 4; ++k) m = (v[k] > m || v[k] != v[k]) ? v[k] : m;
  y[idx] = m;
}

__global__ __launch_bounds__(256) void maxpool2_backward_kernel(const double* __restrict__ x, const double* __restrict__ gy,
                                                                int64_t planes, int H, int W, double* __restrict__ gx) {
  const int Ho = H / 2, Wo = W / 2;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= planes * Ho * Wo) return;
  const int j = (int)(idx % Wo);
  const int i = (int)((idx / Wo) % Ho);
  const int64_t plane = idx / ((int64_t)Wo * Ho);
  const size_t at = (size_t)plane * H * W + (size_t)(2 * i) * W + 2 * j;
  const double v[4] = {x[at], x[at + 1], x[at + W], x[at + W + 1]};
  double m = v[0];
  int win = 0;
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    if (v[k] > m || v[k] != v[k]) {
      m = v[k];
      win = k;
    }
  }
  const double g = gy[idx];
  gx[at] = win == 0 ? g : 0.0;
  gx[at + 1] = win == 1 ? g : 0.0;
  gx[at + W] = win == 2 ? g : 0.0;
  gx[at + W + 1] = win == 3 ? g : 0.0;
  // an odd last row / column belongs to no window
  if (j == Wo - 1 && (W & 1)) {
    gx[at + 2] = 0.0;
    gx[at + W + 2] = 0.0;
  }
  if (i == Ho - 1 && (H & 1)) {
    gx[at + 2 * W] = 0.0;
    gx[at + 2 * W + 1] = 0.0;
    if (j == Wo - 1 && (W & 1)) gx[at + 2 * W + 2] = 0.0;
  }
}

// ---- the two elementwise ends of the dense UNITARY route (A4 nets at inference: AmplitudeEmbedding -> weight-only layers
// -> probs, reference nn/qdense.py:40-47, 56-57; the circuit does not depend on the data, so a batch is one product with
// the cached circuit unitary -- the same move as the reference's own eval-mode QConv2d, nn/qconv.py:96-113) -----------------
// rows of AmplitudeEmbedding(normalize=True, pad_with): (B, F) float64 -> (B, D) float32, v / |v| with D - F pad entries
__global__ __launch_bounds__(256) void amp_embed_rows_kernel(const double* __restrict__ x, int64_t x_ld, int64_t batch,
                                                             int F, int D, double pad_with, double offset,
                                                             float* __restrict__ v) {
  __shared__ double s_red[4];
  const int64_t b = blockIdx.x;
  if (b >= batch) return;
  const double* __restrict__ row = x + b * x_ld;
  double n2 = 0.0;
  for (int j = threadIdx.x; j < F; j += 256) {
    const double t = row[j] + offset;
    n2 = fma(t, t, n2);
  }
  for (int off = 32; off > 0; off >>= 1) n2 += __shfl_down(n2, off, 64);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = n2;
  __syncthreads();
  const double tot = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]) + pad_with * pad_with * (double)(D - F);
  const double inv = 1.0 / sqrt(tot);
  float* __restrict__ out = v + b * (int64_t)D;
  for (int j = threadIdx.x; j < D; j += 256) out[j] = (float)((j < F ? row[j] + offset : pad_with) * inv);
}

// probabilities of the first `cols` outcomes from (B, 2 cols) float32 amplitudes [Re | Im], post-processed:
// out (B, cols) float64 = clamp((Re^2 + Im^2) * scale, 0, 1)
__global__ __launch_bounds__(256) void prob_post_kernel(const float* __restrict__ a, int64_t total, int cols, double scale,
                                                        double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int64_t b = i / cols;
  const int c = (int)(i - b * cols);
  const float re = a[b * 2 * cols + c], im = a[b * 2 * cols + cols + c];
  const double p = (double)(re * re + im * im) * scale;
  out[i] = p < 0.0 ? 0.0 : (p > 1.0 ? 1.0 : p);
}

}  // namespace qiddm

namespace {

int make_geom(int64_t batch, int64_t channels, int64_t hw, qiddm::NormGeom* g) {
  if (batch < 1 || channels < 1 || hw < 1) return fail(QIDDM_ERR_INVALID, "BatchNorm needs batch, channels, hw >= 1");
  if (channels > 65535 || batch * channels * hw >= ((int64_t)1 << 40))
    return fail(QIDDM_ERR_UNSUPPORTED, "BatchNorm tensor too large for one launch");
  g->batch = batch;
  g->hw = hw;
  g->channels = (int32_t)channels;
  g->slices = (int32_t)(batch < qiddm::kNormMaxSlices ? batch : qiddm::kNormMaxSlices);
  return QIDDM_OK;
}

int launched(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(QIDDM_ERR_LAUNCH, "%s launch failed: %s", what, hipGetErrorString(e));
  return QIDDM_OK;
}

}  // namespace

extern "C" {

int64_t qiddm_batchnorm_workspace_bytes(int64_t batch, int64_t channels, int64_t hw) {
  qiddm::NormGeom g;
  if (make_geom(batch, channels, hw, &g) != QIDDM_OK) return -1;
  return (int64_t)g.channels * g.slices * 2 * (int64_t)sizeof(double);
}

int qiddm_batchnorm_train_forward(const double* x, int64_t batch, int64_t channels, int64_t hw,
                                  const double* weight, const double* bias, double* running_mean,
                                  double* running_var, double momentum, double eps, double* y, double* save_mean,
                                  double* save_invstd, void* workspace, int64_t workspace_bytes, void* stream) {
  qiddm::NormGeom g;
  const int rc = make_geom(batch, channels, hw, &g);
  if (rc != QIDDM_OK) return rc;
  if (!x || !y || !save_mean || !save_invstd || !workspace)
    return fail(QIDDM_ERR_INVALID, "x/y/save_mean/save_invstd/workspace is NULL");
  if (workspace_bytes < qiddm_batchnorm_workspace_bytes(batch, channels, hw))
    return fail(QIDDM_ERR_INVALID, "workspace too small");
  if (!(eps >= 0.0)) return fail(QIDDM_ERR_INVALID, "eps < 0");
  if (batch * hw == 1)   // torch.nn.BatchNorm2d in training mode raises for this shape (no variance to estimate)
    return fail(QIDDM_ERR_INVALID, "Expected more than 1 value per channel when training, got input size (%lld, %lld, %lld)",
                (long long)batch, (long long)channels, (long long)hw);
  hipStream_t st = static_cast<hipStream_t>(stream);
  double* partial = static_cast<double*>(workspace);
  const dim3 grid((unsigned)g.channels, (unsigned)g.slices);
  hipLaunchKernelGGL(qiddm::bn_stats_kernel, grid, dim3(qiddm::kNormThreads), 0, st, x, g, partial);
  hipLaunchKernelGGL(qiddm::bn_apply_kernel, grid, dim3(qiddm::kNormThreads), 0, st, x, g, partial, weight, bias,
                     running_mean, running_var, momentum, eps, y, save_mean, save_invstd);
  return launched("bn_apply_kernel");
}

int qiddm_batchnorm_backward(const double* x, const double* grad_y, int64_t batch, int64_t channels, int64_t hw,
                             const double* weight, const double* save_mean, const double* save_invstd,
                             double* grad_x, double* grad_weight, double* grad_bias, void* workspace,
                             int64_t workspace_bytes, void* stream) {
  qiddm::NormGeom g;
  const int rc = make_geom(batch, channels, hw, &g);
  if (rc != QIDDM_OK) return rc;
  if (!x || !grad_y || !save_mean || !save_invstd || !workspace)
    return fail(QIDDM_ERR_INVALID, "x/grad_y/save_mean/save_invstd/workspace is NULL");
  if (workspace_bytes < qiddm_batchnorm_workspace_bytes(batch, channels, hw))
    return fail(QIDDM_ERR_INVALID, "workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  double* partial = static_cast<double*>(workspace);
  const dim3 grid((unsigned)g.channels, (unsigned)g.slices);
  hipLaunchKernelGGL(qiddm::bn_back_stats_kernel, grid, dim3(qiddm::kNormThreads), 0, st, x, grad_y, g, save_mean,
                     save_invstd, partial);
  hipLaunchKernelGGL(qiddm::bn_back_apply_kernel, grid, dim3(qiddm::kNormThreads), 0, st, x, grad_y, g, partial,
                     weight, save_mean, save_invstd, grad_x, grad_weight, grad_bias);
  return launched("bn_back_apply_kernel");
}

int qiddm_batchnorm_backward_stats(const double* x, const double* grad_y, int64_t batch, int64_t channels, int64_t hw,
                                   const double* weight, const double* save_mean, const double* save_invstd,
                                   double* grad_weight, double* grad_bias, double* coef, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
  qiddm::NormGeom g;
  const int rc = make_geom(batch, channels, hw, &g);
  if (rc != QIDDM_OK) return rc;
  if (!x || !grad_y || !save_mean || !save_invstd || !coef || !workspace)
    return fail(QIDDM_ERR_INVALID, "x/grad_y/save_mean/save_invstd/coef/workspace is NULL");
  if (workspace_bytes < qiddm_batchnorm_workspace_bytes(batch, channels, hw))
    return fail(QIDDM_ERR_INVALID, "workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  double* partial = static_cast<double*>(workspace);
  const dim3 grid((unsigned)g.channels, (unsigned)g.slices);
  hipLaunchKernelGGL(qiddm::bn_back_stats_kernel, grid, dim3(qiddm::kNormThreads), 0, st, x, grad_y, g, save_mean,
                     save_invstd, partial);
  hipLaunchKernelGGL(qiddm::bn_back_coef_kernel, dim3(1), dim3(64), 0, st, g, partial, weight, save_mean, save_invstd,
                     grad_weight, grad_bias, coef);
  return launched("bn_back_coef_kernel");
}

int qiddm_upsample2x_forward(const double* x, int64_t planes, int64_t height, int64_t width, const double* ah,
                             const double* aw, double* y, void* stream) {
  if (planes < 1 || height < 1 || width < 1 || height > (1 << 14) || width > (1 << 14))
    return fail(QIDDM_ERR_INVALID, "bad upsample geometry");
  if (!x || !ah || !aw || !y) return fail(QIDDM_ERR_INVALID, "x/ah/aw/y is NULL");
  const int64_t total = planes * height * width;
  if (total >= ((int64_t)1 << 37)) return fail(QIDDM_ERR_UNSUPPORTED, "tensor too large for one launch");
  hipLaunchKernelGGL(qiddm::upsample2x_forward_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, ah, aw, planes, (int)height, (int)width, y);
  return launched("upsample2x_forward_kernel");
}

int qiddm_upsample2x_backward(const double* grad_y, int64_t planes, int64_t height, int64_t width, const double* ah,
                              const double* aw, double* grad_x, void* stream) {
  if (planes < 1 || height < 1 || width < 1 || height > (1 << 14) || width > (1 << 14))
    return fail(QIDDM_ERR_INVALID, "bad upsample geometry");
  if (!grad_y || !ah || !aw || !grad_x) return fail(QIDDM_ERR_INVALID, "grad_y/ah/aw/grad_x is NULL");
  const int64_t total = planes * height * width;
  if (total >= ((int64_t)1 << 39)) return fail(QIDDM_ERR_UNSUPPORTED, "tensor too large for one launch");
  const int64_t big = 4 * height * width;
  if (big <= qiddm::kUpsMaxPlane) {
    // (same arithmetic in the same order as the kernel below: identical results)
    const int pp = qiddm::ups_planes_per_trip((int)big);
    const size_t smem = ((size_t)pp * big + 4 * (size_t)(height + width)) * sizeof(double);
    const int64_t per_cu = std::min<int64_t>(8, (int64_t)(160 * 1024) / (int64_t)(smem + 256));
    const int64_t resident = 256 * std::max<int64_t>(1, per_cu);
    const int64_t trips = (planes + pp - 1) / pp;
    hipLaunchKernelGGL(qiddm::upsample2x_backward_lds_kernel, dim3((unsigned)std::min<int64_t>(trips, resident)),
                       dim3(256), smem, static_cast<hipStream_t>(stream), grad_y, ah, aw, planes, (int)height,
                       (int)width, grad_x);
    return launched("upsample2x_backward_lds_kernel");
  }
  hipLaunchKernelGGL(qiddm::upsample2x_backward_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), grad_y, ah, aw, planes, (int)height, (int)width, grad_x);
  return launched("upsample2x_backward_kernel");
}

int qiddm_maxpool2_forward(const double* x, int64_t planes, int64_t height, int64_t width, double* y, void* stream) {
  if (planes < 1 || height < 2 || width < 2 || height > (1 << 14) || width > (1 << 14))
    return fail(QIDDM_ERR_INVALID, "bad max-pool geometry");
  if (!x || !y) return fail(QIDDM_ERR_INVALID, "x/y is NULL");
  const int64_t total = planes * (height / 2) * (width / 2);
  if (total >= ((int64_t)1 << 39)) return fail(QIDDM_ERR_UNSUPPORTED, "tensor too large for one launch");
  hipLaunchKernelGGL(qiddm::maxpool2_forward_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, planes, (int)height, (int)width, y);
  return launched("maxpool2_forward_kernel");
}

int qiddm_maxpool2_backward(const double* x, const double* grad_y, int64_t planes, int64_t height, int64_t width,
                            double* grad_x, void* stream) {
  if (planes < 1 || height < 2 || width < 2 || height > (1 << 14) || width > (1 << 14))
    return fail(QIDDM_ERR_INVALID, "bad max-pool geometry");
  if (!x || !grad_y || !grad_x) return fail(QIDDM_ERR_INVALID, "x/grad_y/grad_x is NULL");
  const int64_t total = planes * (height / 2) * (width / 2);
  if (total >= ((int64_t)1 << 39)) return fail(QIDDM_ERR_UNSUPPORTED, "tensor too large for one launch");
  hipLaunchKernelGGL(qiddm::maxpool2_backward_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, grad_y, planes, (int)height, (int)width, grad_x);
  return launched("maxpool2_backward_kernel");
}

int qiddm_amp_embed_rows(const double* x, int64_t batch, int64_t x_ld, int64_t features, int32_t n_qubits, double pad_with,
                         double offset, float* v, void* stream) {
  if (batch < 0 || features < 1 || n_qubits < 1 || n_qubits > 14 || features > ((int64_t)1 << n_qubits) || x_ld < features)
    return fail(QIDDM_ERR_INVALID, "bad amplitude-embedding geometry");
  if (batch == 0) return QIDDM_OK;
  if (!x || !v) return fail(QIDDM_ERR_INVALID, "x/v is NULL");
  if (batch > 0x7fffffff) return fail(QIDDM_ERR_UNSUPPORTED, "too many rows for one launch");
  hipLaunchKernelGGL(qiddm::amp_embed_rows_kernel, dim3((unsigned)batch), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     x_ld, batch, (int)features, 1 << n_qubits, pad_with, offset, v);
  return launched("amp_embed_rows_kernel");
}

int qiddm_prob_post(const float* amplitudes, int64_t batch, int64_t cols, double scale, double* out, void* stream) {
  if (batch < 0 || cols < 1 || cols > (1 << 20)) return fail(QIDDM_ERR_INVALID, "bad geometry");
  if (batch == 0) return QIDDM_OK;
  if (!amplitudes || !out) return fail(QIDDM_ERR_INVALID, "amplitudes/out is NULL");
  const int64_t total = batch * cols;
  if (total >= ((int64_t)1 << 39)) return fail(QIDDM_ERR_UNSUPPORTED, "tensor too large for one launch");
  hipLaunchKernelGGL(qiddm::prob_post_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), amplitudes, total, (int)cols, scale, out);
  return launched("prob_post_kernel");
}

}  // extern "C"
