// qsim_qconv_dx.h -- dL/dx of the quantum convolution's unitary-route backward WITHOUT the feature-gradient matrix.
//
// qsim_qconv_train_mfma.h's product (2) makes g[j][m] for every patch feature j = (c, tap) of every output pixel m and
// qconv_fold_t_kernel sums the kh kw entries every input element appeared in: F x M float32 written and read back
// (1.16 GB for the last up-convolution of unet_simple at 2560 x 28 x 28 pixels -- the step's largest stream).  Folding
// commutes with the product.  With W3 = W2 / |v| and k = dot / |v|^2 per output pixel (what product (1)'s epilogue
// already holds; reference: the same autograd path, nn/qconv.py:46, 58-87),
//
//   g[(c, tap)][m] = 2 sum_cc rt[(c, tap)][cc] W3[m][cc]  -  v[m][(c, tap)] k[m]
//
// and v[m][(c, tap)] = x[c][p] + 0.1 for the ONE input element p = m + off(tap) it was gathered from, so
//
//   dL/dx[c][p] = sum_{tap valid} sum_cc rt[(c, tap)][cc] 2 W3[p - off(tap)][cc]  -  (x[c][p] + 0.1) sum_{tap valid} k[p - off(tap)]
//
// -- a transposed convolution of the per-pixel rows (2 C_out + 1 floats per pixel instead of F) with the rows table:
// per tile of 64 input pixels one 64 x (taps * 2 C_out) by (taps * 2 C_out) x C product on v_mfma_f32_16x16x4_f32 and
// a 3 x 3 box sum.  The thin-product kernel writes `wpix` = [M][2 CO] rows of 2 W3, then [M] values of k
// (TrainConv::wpix) and skips product (2) altogether.  Same-size convolutions only (Ho = H, Wo = W: every layer of the
// reference's UNets); anything else keeps the fold.
#pragma once
#include "qsim_qconv_train.h"

namespace qiddm {

constexpr int kDxTile = 64;
constexpr int kDxThreads = 256;

__host__ __device__ inline int dx_halo(const TrainConv& tc) {   // source pixels in front of / behind a tile's range
  const int lo = tc.ph * tc.W + tc.pw, hi = (tc.kh - 1 - tc.ph) * tc.W + (tc.kw - 1 - tc.pw);
  return lo > hi ? lo : hi;
}
// channel columns of the staged rows table: 16, or 32 with the columns of rows 2, 3 (mod 4) rotated by 16 -- the four
// rows a B fragment reads then sit in four different quarters of the LDS banks
__host__ __device__ inline int dx_cpad(int C) { return C <= 16 ? 16 : 32; }
__host__ __device__ inline int dx_bcol(int row, int c, int cp) { return cp == 16 ? c : ((c + 16 * ((row >> 1) & 1)) & 31); }
template <int K2>
__host__ __device__ inline size_t dx_lds_bytes(const TrainConv& tc) {
  const int n_src = kDxTile + 2 * dx_halo(tc), kk = tc.kh * tc.kw;
  // source rows, k of the source pixels, B = rows table as [(tap, cc)][channel], the output tile [channel][64 + 1],
  // per-pixel tap masks and box sums
  return ((size_t)n_src * (K2 + 1) + n_src + (size_t)kk * K2 * dx_cpad(tc.C) + (size_t)dx_cpad(tc.C) * (kDxTile + 1) +
          2 * kDxTile) * sizeof(float);
}

using dx_f32x4 = float __attribute__((ext_vector_type(4)));

constexpr int kDxPrefetch = 6;   // most float4 per thread of the next tile's source rows held in registers
constexpr int kDxMaxC = 32;

// Per tile the kernel has two global streams -- the source rows (contiguous in wpix) and x / gx -- and three barriers.
// Both streams are issued a phase ahead: the NEXT tile's source rows go into registers right after this tile's were
// written to LDS, this tile's x values at the top of the tile; what is left on the critical path is LDS and the MFMAs.
// CB: 16-channel blocks compiled in (1: C <= 16, 2: C <= 32) -- sizes the accumulators and the x / gx slots of a thread
// PF: float4 per thread of the next tile's source rows held in registers (2, 4 or 6: the host picks the smallest that
//     covers 64 + 2 halo rows; more rows than that are fetched when the tile starts)
template <int K2, int CB, int PF>
__global__ __launch_bounds__(kDxThreads) void qconv_dx_kernel(const double* __restrict__ x, const float* __restrict__ wpix,
                                                               const float* __restrict__ rt, double* __restrict__ gx,
                                                               const TrainConv tc) {
  constexpr int SS = K2 + 1, K4 = K2 / 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int KK = tc.kh * tc.kw, C = tc.C, CP = dx_cpad(C);
  const int halo = dx_halo(tc), n_src = kDxTile + 2 * halo;
  float* s_src = reinterpret_cast<float*>(smem_raw);        // [n_src][SS]: 2 W3 of source pixel tile_start - halo + r
  float* s_k = s_src + (size_t)n_src * SS;                   // [n_src]
  float* s_b = s_k + n_src;                                  // [KK * K2][CP]
  float* s_out = s_b + (size_t)KK * K2 * CP;                 // [CP][65]
  uint32_t* s_mask = reinterpret_cast<uint32_t*>(s_out + (size_t)CP * (kDxTile + 1));   // [64] valid taps of the pixel
  float* s_ks = reinterpret_cast<float*>(s_mask + kDxTile);  // [64] box sum of k
  const int tid = threadIdx.x, lane = tid & 63;
  const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int64_t hw = (int64_t)tc.H * tc.W;
  const int64_t total = tc.M;                                // input pixels = output pixels (same-size convolution)
  const float* __restrict__ kvec = wpix + (size_t)total * K2;
  // (the host sends tensors below 2^31 elements: 32-bit offsets and divisions in the tile loop)
  const uint32_t hw32 = (uint32_t)hw, total32 = (uint32_t)total;
  const dx_f32x4* __restrict__ wpix4 = reinterpret_cast<const dx_f32x4*>(wpix);

  // B[(tap, cc)][c] = rt[c KK + tap][cc] (read in rt's own order: coalesced); columns c >= C are zero
  for (int i = tid; i < KK * K2 * CP; i += kDxThreads) s_b[i] = 0.f;
  __syncthreads();
  for (int i = tid; i < C * KK * K2; i += kDxThreads) {
    const int cc = i % K2, j = i / K2, tap = j % KK, c = j / KK;
    const int r = tap * K2 + cc;
    s_b[(size_t)r * CP + dx_bcol(r, c, CP)] = rt[i];
  }

  const int64_t tiles = (total + kDxTile - 1) / kDxTile;
  const int n4 = n_src * K4;   // float4 elements of a tile's source rows
  dx_f32x4 pre[PF];
  float pre_k = 0.f;
  auto issue_rows = [&](int64_t tile) {
    const int32_t first = (int32_t)(tile * kDxTile) - halo;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int i = tid + u * kDxThreads;
      const int32_t m = first + i / K4;          // (K4 is a power of two)
      pre[u] = dx_f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < n4 && m >= 0 && (uint32_t)m < total32) pre[u] = wpix4[(uint32_t)m * (uint32_t)K4 + (uint32_t)(i % K4)];
    }
    const int32_t mk = first + tid;
    pre_k = (tid < n_src && mk >= 0 && (uint32_t)mk < total32) ? kvec[(uint32_t)mk] : 0.f;
  };
  if ((int64_t)blockIdx.x < tiles) issue_rows(blockIdx.x);

  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t p0 = tile * kDxTile;
    // (image of the tile's first pixel: one 64-bit division per tile, uniform; 32-bit ones per thread from there)
    const uint32_t b0 = (uint32_t)p0 / hw32;
    const uint32_t rem0 = (uint32_t)p0 - b0 * hw32;
    // ---- this tile's x values: thread = (channel, pixel), pixels fastest (512-byte rows of x and gx) ------------------
    constexpr int XS = 4 * CB;   // (channel, pixel) slots of a thread: 16 CB channels x 64 pixels / 256 threads
    double xv[XS];
    uint32_t xe[XS];   // element offset, 0xffffffff: none
#pragma unroll
    for (int u = 0; u < XS; ++u) {
      const int i = tid + u * kDxThreads;
      const int c = i >> 6, pl = i & 63;
      xe[u] = 0xffffffffu;
      xv[u] = 0.0;
      if (c < C && p0 + pl < total) {
        const uint32_t rem = rem0 + (uint32_t)pl, db = rem / hw32;
        xe[u] = ((b0 + db) * (uint32_t)C + (uint32_t)c) * hw32 + (rem - db * hw32);
        xv[u] = x[xe[u]];
      }
    }
    __syncthreads();   // the previous tile's readers are done (and B is staged)
    // ---- source rows [p0 - halo, p0 + 64 + halo) from the registers (zero outside [0, M)) ---------------------------
    {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int i = tid + u * kDxThreads;
        if (i < n4) {
          float* dst = s_src + (size_t)(i / K4) * SS + 4 * (i % K4);
          dst[0] = pre[u].x;
          dst[1] = pre[u].y;
          dst[2] = pre[u].z;
          dst[3] = pre[u].w;
        }
      }
      if (tid < n_src) s_k[tid] = pre_k;
      // wide images: what the registers do not hold
      const int64_t first = p0 - halo;
      for (int i = tid + PF * kDxThreads; i < n4; i += kDxThreads) {
        const int64_t m = first + i / K4;
        dx_f32x4 v{0.f, 0.f, 0.f, 0.f};
        if (m >= 0 && m < total) v = wpix4[(size_t)m * K4 + (i % K4)];
        float* dst = s_src + (size_t)(i / K4) * SS + 4 * (i % K4);
        dst[0] = v.x;
        dst[1] = v.y;
        dst[2] = v.z;
        dst[3] = v.w;
      }
      for (int r = tid + kDxThreads; r < n_src; r += kDxThreads) {
        const int64_t m = first + r;
        s_k[r] = (m >= 0 && m < total) ? kvec[m] : 0.f;
      }
    }
    // ---- which taps of this input pixel have a source output pixel (thread = pixel) -----------------------------------
    if (tid < kDxTile) {
      const int64_t p = p0 + tid;
      uint32_t mask = 0;
      if (p < total) {
        const uint32_t pix = (rem0 + (uint32_t)tid) % hw32;
        const int i = (int)(pix / (uint32_t)tc.W), j = (int)pix - i * tc.W;
        uint32_t cols = 0;   // taps of one kernel row whose source column exists
        for (int dj = 0; dj < tc.kw; ++dj) {
          const int oj = j - dj + tc.pw;
          cols |= (uint32_t)(oj >= 0 && oj < tc.W) << dj;
        }
        for (int di = 0; di < tc.kh; ++di) {
          const int oi = i - di + tc.ph;
          mask |= (oi >= 0 && oi < tc.H) ? cols << (di * tc.kw) : 0u;
        }
      }
      s_mask[tid] = mask;
    }
    __syncthreads();
    if (tile + gridDim.x < tiles) issue_rows(tile + gridDim.x);   // lands while the product and the output run
    if (tid < kDxTile) {
      const uint32_t mask = s_mask[tid];
      float ks = 0.f;
      for (int di = 0; di < tc.kh; ++di)
        for (int dj = 0; dj < tc.kw; ++dj)
          if ((mask >> (di * tc.kw + dj)) & 1u) ks += s_k[tid + halo - ((di - tc.ph) * tc.W + (dj - tc.pw))];
      s_ks[tid] = ks;
    }
    // ---- the product: wave q owns input pixels 16 q .. 16 q + 15, all channels ---------------------------------------
    {
      dx_f32x4 acc[CB];
#pragma unroll
      for (int nb = 0; nb < CB; ++nb) acc[nb] = dx_f32x4{0.f, 0.f, 0.f, 0.f};
      const uint32_t mask = s_mask[16 * q + l15];
      const int row0 = 16 * q + l15 + halo;
      const int second = CP == 16 ? 0 : (dx_bcol(l4, 16 + l15, CP) - dx_bcol(l4, l15, CP));
      int tap = 0;
      for (int di = 0; di < tc.kh; ++di) {
        for (int dj = 0; dj < tc.kw; ++dj, ++tap) {
          const bool ok = (mask >> tap) & 1u;
          const float* __restrict__ arow = s_src + (size_t)(row0 - ((di - tc.ph) * tc.W + (dj - tc.pw))) * SS + l4;
          const float* __restrict__ brow = s_b + (size_t)(tap * K2 + l4) * CP + dx_bcol(l4, l15, CP);   // (K2 % 4 == 0)
          float a[K4], b0v[K4], b1v[K4];
#pragma unroll
          for (int s4 = 0; s4 < K4; ++s4) {
            a[s4] = arow[4 * s4];
            b0v[s4] = brow[(size_t)4 * s4 * CP];
            b1v[s4] = CB > 1 ? brow[(size_t)4 * s4 * CP + second] : 0.f;
          }
#pragma unroll
          for (int s4 = 0; s4 < K4; ++s4) {
            const float av = ok ? a[s4] : 0.f;
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0v[s4], acc[0], 0, 0, 0);
            if constexpr (CB > 1) acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1v[s4], acc[1], 0, 0, 0);
          }
        }
      }
      // C/D layout: row (pixel) = 4 (lane >> 4) + reg, column (channel) = lane & 15
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s_out[(size_t)l15 * (kDxTile + 1) + 16 * q + 4 * l4 + r] = acc[0][r];
        if constexpr (CB > 1) s_out[(size_t)(16 + l15) * (kDxTile + 1) + 16 * q + 4 * l4 + r] = acc[1][r];
      }
    }
    __syncthreads();
    // ---- out ---------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int u = 0; u < XS; ++u) {
      const int i = tid + u * kDxThreads;
      const int c = i >> 6, pl = i & 63;
      if (xe[u] != 0xffffffffu) {
        const float v = (float)xv[u] + 0.1f;
        gx[xe[u]] = (double)(s_out[(size_t)c * (kDxTile + 1) + pl] - v * s_ks[pl]);
      }
    }
  }
}

}  // namespace qiddm
