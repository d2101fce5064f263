// qsim_qconv_fwd.h -- the quantum convolution's GEMM with the patch matrix read from an LDS copy of the image.
//
// qconv_gemm_kernel (qsim_unitary.h; reference: eval-mode QConv2d.forward, nn/qconv.py:105-113, 58-69) gathers every
// patch element from memory: kh kw loads per input element, 2.3 GB through the L2 for the last up-convolution of
// unet_simple at 2560 x 28 x 28 pixels, which is what bounds it (24 TFLOP/s of the 157 the f32 matrix cores have).  For a
// same-size convolution a tile of T consecutive output pixels only touches input pixels T0 - halo .. T0 + T + halo of every
// channel (halo = one image row + 1 for 3 x 3), so here a workgroup
//   * copies that range of x once, coalesced, as float32 into LDS (C x (T + 2 halo) values; the NEXT tile's range is
//     already in registers, issued a tile ahead),
//   * keeps the whole packed operand (K_pad x 2 C_out, from qconv_pack_kernel) in LDS for all its tiles,
//   * and runs the K loop on v_mfma_f32_16x16x4_f32 with A fragments read straight from the image copy: patch column
//     f = (c, tap) of pixel m is s_x[c][m + halo + off(tap)] (+ 0.1, or 0.1 alone where the tap is outside the image),
//     one table lookup per k -- no staged A tile, no barrier inside the K loop.
// The normalisation |v|^2 accumulates from the same fragments.  Epilogue as qconv_gemm_kernel: (|Re|^2 + |Im|^2) D / 2 /
// |v|^2 clamped to [0, 1], optional eval-mode BatchNorm, [channel][pixel] tile in LDS, 1 KB rows out.
// Same-size convolutions (Ho = H, Wo = W) with C_out <= 32, no fused upsampling; everything else keeps qconv_gemm_kernel.
#pragma once
#include "qsim_unitary.h"

namespace qiddm {

constexpr int kFwdTile = 128;     // output pixels per tile: 4 wavefronts x 2 row blocks of 16
constexpr int kFwdThreads = 256;

__host__ __device__ inline int fwd_halo(const GemmConv& g) {
  const int lo = g.ph * g.W + g.pw, hi = (g.kh - 1 - g.ph) * g.W + (g.kw - 1 - g.pw);
  return lo > hi ? lo : hi;
}
__host__ __device__ inline int fwd_xs(const GemmConv& g) { return (kFwdTile + 2 * fwd_halo(g)) | 1; }   // odd row stride
// column of packed-operand element (row, col) inside its LDS row: the four rows a B fragment reads are rotated into four
// different quarters of the banks (NCOL = 16: the rows are 16 floats long and consecutive already)
template <int NCOL>
__host__ __device__ inline int fwd_bcol(int row, int col) {
  if constexpr (NCOL == 16) return col;
  if constexpr (NCOL == 32) return (col + 16 * ((row >> 1) & 1)) & 31;
  return (col + 16 * (row & 3)) & 63;
}
template <int NCOL>
__host__ __device__ inline size_t fwd_bn_offset(const GemmConv& g) {   // the float regions, then [2][C_out tile] float64
  return (((size_t)g.C * fwd_xs(g) + (size_t)g.K_pad * NCOL + (size_t)g.K_pad + (size_t)(NCOL / 2) * (kFwdTile + 1) +
           kFwdTile) * sizeof(float) + 7) / 8 * 8;
}
template <int NCOL>
__host__ __device__ inline size_t fwd_lds_bytes(const GemmConv& g) {
  // image copy, packed operand, k table, output tile [C_out tile][T + 1] float, 1 / |v|^2 per pixel
  return fwd_bn_offset<NCOL>(g) + (size_t)2 * (NCOL / 2) * sizeof(double);
}

// XPT: float64 values of the next tile's image range a thread holds in registers (C (T + 2 halo) / 256, rounded up to
// one of 1, 4, 12, 20 by the host)
template <int NCOL, int XPT>
__global__ __launch_bounds__(kFwdThreads) void qconv_fwd_halo_kernel(const double* __restrict__ x,
                                                                     const float* __restrict__ w,
                                                                     const float* __restrict__ padv,
                                                                     const double* __restrict__ bn,
                                                                     double* __restrict__ y, const GemmConv g) {
  constexpr int NCB = NCOL / 16, CT = NCOL / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int halo = fwd_halo(g), XS = fwd_xs(g), KK = g.kh * g.kw;
  float* s_x = reinterpret_cast<float*>(smem_raw);                  // [C][XS]: (float)x of flat pixel tile_start - halo + r
  float* s_b = s_x + (size_t)g.C * XS;                              // [K_pad][NCOL], columns rotated (fwd_bcol)
  uint32_t* s_k = reinterpret_cast<uint32_t*>(s_b + (size_t)g.K_pad * NCOL);   // [K_pad]: offset into s_x | tap << 24; bit 31: padding k
  float* s_out = reinterpret_cast<float*>(s_k + g.K_pad);           // [CT][T + 1]
  float* s_inv = s_out + (size_t)CT * (kFwdTile + 1);               // [T]
  double* s_bn = reinterpret_cast<double*>(smem_raw + fwd_bn_offset<NCOL>(g));   // [2][CT]: eval-mode BatchNorm scale, shift
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int64_t hw = (int64_t)g.H * g.W;
  const int64_t total = g.M;

  for (int i = tid; i < g.K_pad * NCOL; i += kFwdThreads) {
    const int row = i / NCOL, col = i - row * NCOL;
    s_b[row * NCOL + fwd_bcol<NCOL>(row, col)] = w[(size_t)row * g.N_pad + col];
  }
  for (int i = tid; i < 2 * CT; i += kFwdThreads) {
    const int k = i / CT, ch = i - k * CT;
    s_bn[i] = (g.has_bn && ch < g.C_out) ? bn[k * g.bn_stride + ch] : (k == 0 ? 1.0 : 0.0);
  }
  for (int f = tid; f < g.K_pad; f += kFwdThreads) {
    uint32_t info = 0x80000000u | (31u << 24);   // k beyond the patch (offset 0, a tap no mask has): the value must be 0
    if (f < g.F) {
      const int c = f / KK, tap = f - c * KK;
      const int di = tap / g.kw, dj = tap - di * g.kw;
      info = (uint32_t)(c * XS + halo + (di - g.ph) * g.W + (dj - g.pw)) | ((uint32_t)tap << 24);
    }
    s_k[f] = info;
  }
  // per-lane constants of the epilogue (column = l15 of every 16-column block)
  float pre[NCB > 1 ? NCB / 2 : 1], pim[NCB > 1 ? NCB / 2 : 1];
  if constexpr (NCB == 1) {
    pre[0] = padv[l15 & 7];
    pim[0] = padv[8 + (l15 & 7)];
  } else {
#pragma unroll
    for (int h = 0; h < NCB / 2; ++h) {
      pre[h] = padv[16 * h + l15];
      pim[h] = padv[CT + 16 * h + l15];
    }
  }

  const int n_x = g.C * XS;   // elements of a tile's image copy (the pad column of a row is written but never read)
  const int64_t tiles = (total + kFwdTile - 1) / kFwdTile;
  double xr[XPT];
  // which element of the image copy a register slot holds does not change from tile to tile: (channel, position) once
  uint32_t xslot[XPT];
#pragma unroll
  for (int u = 0; u < XPT; ++u) {
    const int e = tid + u * kFwdThreads;
    const int c = e < n_x ? e / XS : 0xffff, r = e < n_x ? e - c * XS : 0;
    xslot[u] = ((uint32_t)c << 16) | (uint32_t)r;
  }
  // (the host sends only tensors below 2^31 elements here: 32-bit offsets and divisions throughout the tile loop, which
  //  is otherwise as long as the products it feeds)
  const uint32_t hw32 = (uint32_t)hw, C32 = (uint32_t)g.C;
  const bool one_step = (int64_t)XS <= hw;   // a tile's range crosses at most one image boundary
  auto issue_x = [&](int64_t tile) {
    // flat input pixel q = q0 + r -> (image, pixel): one division per tile, compares (or 32-bit divisions) per slot
    const int32_t q0 = (int32_t)(tile * kFwdTile) - halo;
    const int32_t bq0 = q0 >= 0 ? (int32_t)((uint32_t)q0 / hw32) : -(int32_t)(((uint32_t)(-q0) + hw32 - 1) / hw32);
    const uint32_t rq0 = (uint32_t)(q0 - bq0 * (int32_t)hw32);
    auto slot = [&](int u, uint32_t rem, uint32_t db) {
      const uint32_t c = xslot[u] >> 16, r = xslot[u] & 0xffffu;
      const int32_t b = bq0 + (int32_t)db;
      xr[u] = 0.0;
      if (c != 0xffffu && b >= 0 && q0 + (int32_t)r < (int32_t)total)
        xr[u] = x[((uint32_t)b * C32 + c) * hw32 + (rem - db * hw32)];
    };
    if (one_step) {   // (uniform: the common case carries no division)
#pragma unroll
      for (int u = 0; u < XPT; ++u) {
        const uint32_t rem = rq0 + (xslot[u] & 0xffffu);
        slot(u, rem, (uint32_t)(rem >= hw32));
      }
    } else {
#pragma unroll
      for (int u = 0; u < XPT; ++u) {
        const uint32_t rem = rq0 + (xslot[u] & 0xffffu);
        slot(u, rem, rem / hw32);
      }
    }
  };
  if ((int64_t)blockIdx.x < tiles) issue_x(blockIdx.x);

  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t m0 = tile * kFwdTile;
    __syncthreads();   // the previous tile's readers of s_x / s_out are done (and the tables are staged)
#pragma unroll
    for (int u = 0; u < XPT; ++u) {
      const int e = tid + u * kFwdThreads;
      if (e < n_x) s_x[e] = (float)xr[u];
    }
    for (int e = tid + XPT * kFwdThreads; e < n_x; e += kFwdThreads) {   // (images wider than the registers cover)
      const int c = e / XS, r = e - c * XS;
      const int32_t q = (int32_t)m0 - halo + r;
      float v = 0.f;
      if (q >= 0 && q < (int32_t)total) {
        const uint32_t b = (uint32_t)q / hw32;
        v = (float)x[(b * C32 + (uint32_t)c) * hw32 + ((uint32_t)q - b * hw32)];
      }
      s_x[e] = v;
    }
    __syncthreads();
    if (tile + gridDim.x < tiles) issue_x(tile + gridDim.x);   // lands while this tile's products run

    // ---- this lane's two pixels (row blocks 0 / 1 of the wave's 32): which taps lie inside the image ------------------
    const uint32_t b0 = (uint32_t)m0 / hw32;
    const uint32_t rem0 = (uint32_t)m0 - b0 * hw32;
    uint32_t mask[2];
    int prow[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int pl = wave * 32 + rb * 16 + l15;
      prow[rb] = pl;
      uint32_t mk = 0;
      if (m0 + pl < total) {
        const uint32_t pix = (rem0 + (uint32_t)pl) % hw32;
        const int i = (int)(pix / (uint32_t)g.W), j = (int)pix - i * g.W;
        uint32_t cols = 0;   // taps of one kernel row whose column lies inside the image
        for (int dj = 0; dj < g.kw; ++dj) {
          const int jj = j + dj - g.pw;
          cols |= (uint32_t)(jj >= 0 && jj < g.W) << dj;
        }
        for (int di = 0; di < g.kh; ++di) {
          const int ii = i + di - g.ph;
          mk |= (ii >= 0 && ii < g.H) ? cols << (di * g.kw) : 0u;
        }
      }
      mask[rb] = mk;
    }
    // ---- K loop: A[i = pixel][k] from the image copy, B[k][j = column] from the packed operand ---------------------------
    f32x4 acc[2][NCB];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float n2[2] = {0.f, 0.f};
    const bool live[2] = {m0 + prow[0] < total, m0 + prow[1] < total};
    // K_pad is a multiple of 32: four k steps per trip, written out so that the table entries, then the eight image
    // values and the operand values of the trip are in flight together (as a plain loop every k step waited for three
    // dependent LDS round trips), and without control flow: a padding k reads element 0 and is zeroed by a select
    const float lv0 = live[0] ? 1.f : 0.f, lv1 = live[1] ? 1.f : 0.f;
    for (int kk = 0; kk < g.K_pad; kk += 16) {
      uint32_t info[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) info[s4] = s_k[kk + 4 * s4 + l4];
      float xv[4][2], bv[4][NCB];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const uint32_t off = info[s4] & 0xFFFFFFu;     // (0 for a padding k)
        xv[s4][0] = s_x[off + prow[0]];
        xv[s4][1] = s_x[off + prow[1]];
        const int k = kk + 4 * s4 + l4;
        const float* __restrict__ brow = s_b + (size_t)k * NCOL;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) bv[s4][cb] = brow[fwd_bcol<NCOL>(k, cb * 16 + l15)];
      }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const uint32_t tap = (info[s4] >> 24) & 31u;
        const float keep = (info[s4] >> 31) ? 0.f : 1.f;   // padding k: the value is 0, not 0.1
        const float a0 = ((((mask[0] >> tap) & 1u) ? xv[s4][0] : 0.f) + 0.1f) * (keep * lv0);
        const float a1 = ((((mask[1] >> tap) & 1u) ? xv[s4][1] : 0.f) + 0.1f) * (keep * lv1);
        n2[0] = fmaf(a0, a0, n2[0]);
        n2[1] = fmaf(a1, a1, n2[1]);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          acc[0][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv[s4][cb], acc[0][cb], 0, 0, 0);
          acc[1][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv[s4][cb], acc[1][cb], 0, 0, 0);
        }
      }
    }
    // |v|^2 of a pixel: the four k groups of its lanes
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      float t = n2[rb];
      t += __shfl_xor(t, 16, kWave);
      t += __shfl_xor(t, 32, kWave);
      if (l4 == 0) s_inv[prow[rb]] = (float)(g.post_scale / ((double)t + g.pad_norm2));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- epilogue in the C/D layout: row (pixel) = 4 (lane >> 4) + reg, column = lane & 15 ------------------------------
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int pl = wave * 32 + rb * 16 + 4 * l4 + r;
        const float inv = s_inv[pl];
        if constexpr (NCB == 1) {
          // lanes with column < 8 hold Re of channel column, the others Im of channel column - 8
          const float own = acc[rb][0][r], other = __shfl_xor(own, 8, kWave);
          const float re = (l15 < 8 ? own : other) + pre[0], im = (l15 < 8 ? other : own) + pim[0];
          const float v = fminf(fmaxf((re * re + im * im) * inv, 0.f), 1.f);
          if (l15 < 8) s_out[(size_t)l15 * (kFwdTile + 1) + pl] = v;
        } else {
#pragma unroll
          for (int h = 0; h < NCB / 2; ++h) {
            const float re = acc[rb][h][r] + pre[h], im = acc[rb][NCB / 2 + h][r] + pim[h];
            const float v = fminf(fmaxf((re * re + im * im) * inv, 0.f), 1.f);
            s_out[(size_t)(16 * h + l15) * (kFwdTile + 1) + pl] = v;
          }
        }
      }
    }
    __syncthreads();
    // ---- out: thread = (channel, pixel), pixels fastest.  A compile-time trip count: the wait in front of the next
    //      tile's staging then counts these stores exactly instead of draining them (vmcnt(0) cost a store round trip
    //      per tile)
    {
      const int pl = tid & (kFwdTile - 1);
      const uint32_t rem = rem0 + (uint32_t)pl, db = rem / hw32;
      double* __restrict__ dst = y + ((b0 + db) * (uint32_t)g.C_out * hw32 + (rem - db * hw32));
      const bool in = m0 + pl < total;
#pragma unroll
      for (int u = 0; u < CT / 2; ++u) {
        const int ch = 2 * u + (tid >> 7);
        if (in && ch < g.C_out) {
          dst[(uint32_t)ch * hw32] = (double)s_out[(size_t)ch * (kFwdTile + 1) + pl] * s_bn[ch] + s_bn[CT + ch];
        }
      }
    }
  }
}

}  // namespace qiddm
