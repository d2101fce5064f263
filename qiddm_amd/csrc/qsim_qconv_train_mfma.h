// qsim_qconv_train_mfma.h -- the thin-product backward of the quantum convolution on the f32 matrix cores.
//
// Same mathematics, tables and outputs as qconv_train_backward_kernel (qsim_qconv_train.h; reference: torch autograd
// through Unfold, default.qubit.torch and the post-processing slices of a training QConv2d, nn/qconv.py:46, 58-87):
// per tile of 64 output pixels, with the rows table rt[j][cc] (cc < 2 CO: Re / Im of U[2c, j]) and the gathered,
// un-normalised patch v,
//
//   (1)  a[m][cc]  = sum_j v[m][j] rt[j][cc]                    64 x (F+1) by (F+1) x 2CO
//   (2)  g[j][m]   = sum_cc rt[j][cc] W2[m][cc]                 F x 2CO by 2CO x 64        -> gfeat_t (feature gradients)
//   (3)  h[cc][j] += sum_m W2[m][cc] inv_m v[m][j]              2CO x 64 by 64 x (F+1)     -> hpart (per-workgroup sums)
//
// W2 = t (Re a, Im a) / |v|, t = dL/dy * D/2 where the clamp passes.  The VALU version spends one FMA instruction
// per 64 multiply-adds and half a wavefront per SIMD of occupancy; here the three products run on
// v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 (exact f32 products, the same 157 TFLOP/s peak as the vector ALU but
// 1024 / 2048 multiply-adds per instruction and no cross-wave reduction):
//   (1) wave q owns pixels 16q..16q+15 (all channels, all features): the result stays in its accumulators, the
//       per-pixel epilogue (clamp rule, t, W2, the normalisation's dot term) runs in registers in the C/D layout;
//   (2) wave q owns blocks of 32 features; N = 32 pixels, so a store instruction writes 128-byte rows of gfeat_t;
//   (3) wave q owns blocks of 16 feature columns; the 2CO x 16 accumulators persist across the tiles of the workgroup.
// LDS operands are read in the MFMA fragment layouts directly (odd row strides: conflict-free).
#pragma once
#include "qsim_fused.h"
#include "qsim_qconv_train.h"

namespace qiddm {

constexpr int kTmWaves = 4;
constexpr int kTmThreads = 64 * kTmWaves;
constexpr int kTmTile = 64;  // output pixels per tile

using f32x4 = float __attribute__((ext_vector_type(4)));
using f32x16 = float __attribute__((ext_vector_type(16)));

__host__ __device__ inline int tm_fcols(int F) { return ((F + 1 + 15) / 16) * 16; }   // patch columns + the pad column
__host__ __device__ inline int tm_frows(int F) { return ((F + 1 + 31) / 32) * 32; }   // rows of the LDS rows table
__host__ __device__ inline int tm_v_stride(int F) { return tm_fcols(F) + 1; }         // odd
template <int CO>
__host__ __device__ inline size_t tm_lds_bytes(int F) {
  // rows table, v tile, W2, W2 / |v|, staged dL/dy, norm partials, inv, dot, taps
  return ((size_t)tm_frows(F) * (2 * CO + 1) + (size_t)kTmTile * tm_v_stride(F) + (size_t)2 * kTmTile * (2 * CO + 1) +
          (size_t)kTmTile * (CO + 1) + (size_t)kTmWaves * kTmTile + 2 * kTmTile) * sizeof(float) +
         (size_t)F * sizeof(uint32_t);
}
// ... and, behind those, the three BatchNorm coefficients per channel (float64; TrainConv::bn_coef) when the layer is
// followed by one
template <int CO>
__host__ __device__ inline size_t tm_coef_offset(int F) { return (tm_lds_bytes<CO>(F) + 7) / 8 * 8; }
template <int CO>
__host__ __device__ inline size_t tm_lds_bytes_bn(int F) { return tm_coef_offset<CO>(F) + (size_t)3 * CO * sizeof(double); }

// XT: element type of x -- the float64 activations as the reference holds them, or a float32 copy (same values after
// the (float) conversion every patch element goes through here; half the registers per load in flight)
// SK, SC: kernel extent (SK x SK) and input channels as compile-time constants (0: run-time).  With both known the walk
// over a wave's patch columns has no control flow at all: three instructions per load instead of ~25 and three branches
// BN: the layer is followed by a training-mode BatchNorm2d whose backward transform is applied to dL/dy here
//     (TrainConv::bn_y / bn_coef; the convolution's own output rides in registers next to dL/dy, a tile ahead).  A
//     template parameter, not a run-time test: the extra registers cost the CO = 32 variants their second wave per SIMD
//     (spills), so those layers keep the separate BatchNorm pass (qiddm_qconv_train_bn_ok) and every other variant is
//     compiled without the code.
template <int CO, int JBMAX, typename XT, int SK = 0, int SC = 0, bool BN = false>
__global__ __launch_bounds__(kTmThreads, (JBMAX >= 8 ? 1 : 2)) void qconv_train_backward_mfma_kernel(const XT* __restrict__ x,
                                                                               const double* __restrict__ gy,
                                                                               const float* __restrict__ rt,
                                                                               float* __restrict__ gfeat_t,
                                                                               float* __restrict__ hpart,
                                                                               const TrainConv tc) {
  constexpr int K2 = 2 * CO, RS = K2 + 1, WS = K2 + 1, TS = CO + 1;
  constexpr int NBC = K2 / 16;                  // 16-column blocks of product (1) / 16-row blocks of product (3)
  constexpr int NBRE = CO >= 16 ? CO / 16 : 1;  // blocks holding real parts (CO = 8: one block, Re | Im halves)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int F = tc.F, FC = tm_fcols(F), FS = FC + 1, FR = tm_frows(F);
  float* s_rt = reinterpret_cast<float*>(smem_raw);   // [FR][RS]   row F = 2 x (pad columns' row), rows > F = 0
  float* s_v = s_rt + (size_t)FR * RS;                // [64][FS]   column F = 0.5 (what a pad column holds), > F = 0
  float* s_w = s_v + (size_t)kTmTile * FS;            // [64][WS]   W2
  float* s_w3 = s_w + (size_t)kTmTile * WS;           // [64][WS]   W2 / |v|
  float* s_t = s_w3 + (size_t)kTmTile * WS;           // [64][TS]   dL/dy * D/2 (0 outside the batch / channels)
  float* s_n2 = s_t + (size_t)kTmTile * TS;           // [waves][64]
  float* s_inv = s_n2 + kTmWaves * kTmTile;           // [64]
  float* s_dot = s_inv + kTmTile;                     // [64]
  uint32_t* s_tap = reinterpret_cast<uint32_t*>(s_dot + kTmTile);  // [F]: offset | di << 24 | dj << 28
  double* s_coef = reinterpret_cast<double*>(smem_raw + tm_coef_offset<CO>(tc.F));   // [3][CO], only with tc.bn_y
  const int tid = threadIdx.x, lane = tid & 63;
  const int q = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index, provably uniform: scalar addressing and branches
  const int l15 = lane & 15, l4 = lane >> 4, l31 = lane & 31, l5 = lane >> 5;

  for (int i = tid; i < FR * RS; i += kTmThreads) {
    const int j = i / RS, cc = i - j * RS;
    float v = 0.f;
    if (cc < K2 && j <= F) v = rt[(size_t)j * K2 + cc] * (j == F ? 2.0f : 1.0f);
    s_rt[i] = v;
  }
  for (int i = tid; i < kTmTile * FS; i += kTmThreads) s_v[i] = 0.f;
  if constexpr (BN) {   // (read per tile from LDS: scalar loads from memory sat on every tile's critical path)
    for (int i = tid; i < 3 * CO; i += kTmThreads) {
      const int k = i / CO, c = i - k * CO;
      s_coef[i] = c < tc.C_out ? tc.bn_coef[k * tc.C_out + c] : 0.0;
    }
  }
  // persistent accumulators of product (3): this wave's feature-column blocks jb = q, q + 4, ...
  f32x4 acch[JBMAX][NBC];
#pragma unroll
  for (int u = 0; u < JBMAX; ++u)
#pragma unroll
    for (int mb = 0; mb < NBC; ++mb) acch[u][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int n_jb = FC / 16;

  const int64_t pixels = (int64_t)tc.Ho * tc.Wo;
  const int64_t tiles = (tc.M + kTmTile - 1) / kTmTile;
  const bool stamp = tc.stamps != nullptr && blockIdx.x == 0 && tid == 0;
  unsigned long long st_sum[5] = {0, 0, 0, 0, 0}, st_prev = 0;
  auto mark = [&](int phase) {
    if (stamp) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      st_sum[phase] += now - st_prev;
      st_prev = now;
    }
  };
  if (stamp) st_prev = __builtin_amdgcn_s_memtime();

  // ---- gather state (lane = pixel): the global loads of a tile are ISSUED one tile ahead -- right after product (1) of
  // the previous tile -- and stay in flight in registers while products (2) and (3) run.  PRE patch columns per thread are
  // covered that way (all of them up to 4 PRE features); the rest is fetched, 16 at a time, when the tile starts.
  // register budget: 2 waves per SIMD; the JBMAX = 8 variants (>= 256 patch features: > 80 KB of LDS, one workgroup per
  // CU anyway) are compiled for one and prefetch deeper
  constexpr bool X32 = sizeof(XT) == 4;
  constexpr int PRE = X32 ? (JBMAX <= 2 ? 32 : (JBMAX <= 4 ? (CO <= 16 ? 48 : 16) : 64))
                          : (JBMAX <= 2 ? (CO <= 16 ? 32 : 24) : (JBMAX <= 4 ? (CO == 8 ? 28 : (CO == 16 ? 20 : 8)) : 40));
  constexpr int GC = CO / kTmWaves;
  static_assert(PRE <= 64, "one in-image bit per prefetched column in a 64-bit mask");
  XT raw[PRE];
  double graw[GC], yraw[BN ? GC : 1];   // dL/dy of the tile in flight (and, BN, the convolution's own output)
  uint64_t inb_mask = 0;
  uint32_t glive_mask = 0;
  int g_i0 = 0, g_j0 = 0;
  // This wave's patch columns are walked in (tap, channel) order -- tap t = di kw + dj, then its channels c = q, q + 4,
  // ... -- by scalar counters: the clamped element offset of a tap (and whether the tap lies inside the image) is
  // per-lane work done once per TAP, and a column costs one add, the address and the load.  (A tap table in LDS made
  // every load wait for its own LDS round trip; per-column tap arithmetic cost ~30 instructions a load: issuing a
  // tile's loads was 31 % of the tile either way.)  Column index of (t, c): j = c kh kw + t.
  constexpr bool STATIC = SK > 0 && SC > 0;
  static_assert(!STATIC || SC % kTmWaves == 0, "the static walk deals whole channels to the four waves");
  constexpr int SKK = SK * SK, SCQ = SC / kTmWaves, SCOLS = SKK * SCQ;   // taps, channels and columns of a wave
  const int KK = STATIC ? SKK : tc.kh * tc.kw;
  const uint32_t plane = (uint32_t)(tc.H * tc.W);
  uint32_t g_base = 0;  // element offset of this pixel's image in x (the host checks numel(x), numel(gy) < 2^32): 32-bit
                        // offsets from the kernel-argument base keep a load's address in ONE register
  struct Walk {
    int t, di, dj, c;   // wave-uniform
  };
  Walk rest{0, 0, 0, 0};  // where the prefetched columns of the tile in flight end
  // per-lane: offset of tap (di, dj) of this lane's pixel clamped into the image (zero padding reads the nearest in-image
  // element and is zeroed afterwards: NOT one shared dummy address, which serialises the grid on one L2 channel)
  auto tap_offset = [&](const Walk& w, bool& inside) -> uint32_t {
    const int ii = g_i0 + w.di, jj = g_j0 + w.dj;
    const int iic = min(max(ii, 0), tc.H - 1), jjc = min(max(jj, 0), tc.W - 1);
    inside = ii == iic && jj == jjc;
    return g_base + __umul24((uint32_t)iic, (uint32_t)tc.W) + (uint32_t)jjc;
  };
  auto advance = [&](Walk& w) -> bool {   // next column; true when the tap changed
    w.c += kTmWaves;
    if (w.c < tc.C) return false;
    w.c = q;
    ++w.t;
    if (++w.dj == tc.kw) {
      w.dj = 0;
      ++w.di;
    }
    return true;
  };
  auto issue_gather = [&](int64_t tile) {
    const int64_t m_raw = tile * kTmTile + lane;
    const bool valid = m_raw < tc.M;
    const int64_t m = valid ? m_raw : tc.M - 1;
    // (batch, pixel) of the output position: 32-bit division whenever the pixel count allows it
    const int64_t b = tc.M < ((int64_t)1 << 31) ? (int64_t)((uint32_t)m / (uint32_t)pixels) : m / pixels;
    const int pix = (int)(m - b * pixels);
    const int oi = pix / tc.Wo, oj = pix - oi * tc.Wo;
    g_i0 = oi - tc.ph;
    g_j0 = oj - tc.pw;
    const uint32_t gpix = (uint32_t)(b * tc.gy_bstride + pix);
    const uint32_t ypix = (uint32_t)(b * tc.C_out * pixels + pix);   // (the convolution's own output is dense)
    glive_mask = 0;
#pragma unroll
    for (int cu = 0; cu < GC; ++cu) {
      const int c = q + kTmWaves * cu;
      const bool live = valid && c < tc.C_out;
      glive_mask |= (uint32_t)live << cu;
      graw[cu] = 0.0;
      if constexpr (BN) yraw[cu] = 0.0;
      if (c < tc.C_out) {   // wave-uniform; tail pixels read pixel M - 1
        graw[cu] = gy[gpix + (uint32_t)c * (uint32_t)pixels];
        if constexpr (BN) yraw[cu] = tc.bn_y[ypix + (uint32_t)c * (uint32_t)pixels];
      }
    }
    inb_mask = 0;
    g_base = (uint32_t)((int64_t)b * tc.C * tc.H * tc.W);
    if constexpr (STATIC) {   // column u = t * SCQ + ci: tap t, channel q + 4 ci
      const uint32_t qplane = (uint32_t)q * plane;
#pragma unroll
      for (int t = 0; t < SKK; ++t) {
        if (t * SCQ < PRE) {
          bool inside = false;
          const uint32_t off = tap_offset(Walk{t, t / SK, t % SK, 0}, inside) + qplane;
#pragma unroll
          for (int ci = 0; ci < SCQ; ++ci) {
            const int u = t * SCQ + ci;
            if (u < PRE) {
              raw[u] = x[off + (uint32_t)(kTmWaves * ci) * plane];
              inb_mask |= (uint64_t)inside << u;
            }
          }
        }
      }
    } else {
      Walk w{q < tc.C ? 0 : KK, 0, 0, q};   // a wave beyond the channel count owns no column
      bool inside = false;
      uint32_t off = w.t < KK ? tap_offset(w, inside) : 0u;
#pragma unroll
      for (int u = 0; u < PRE; ++u) {
        if (w.t < KK) {
          raw[u] = x[off + (uint32_t)w.c * plane];
          inb_mask |= (uint64_t)inside << u;
          if (advance(w) && w.t < KK) off = tap_offset(w, inside);
        }
      }
      rest = w;
    }
  };
  __syncthreads();  // s_tap is staged
  if ((int64_t)blockIdx.x < tiles) issue_gather(blockIdx.x);

  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t m_base = tile * kTmTile;
    __syncthreads();  // the previous tile's readers of s_v / s_w / s_w3 are done (and the tables are staged)
    {
      float n2 = 0.f;
      if constexpr (STATIC) {
        float* __restrict__ vrow = s_v + (size_t)lane * FS + q * SKK;   // column of (t, ci): (q + 4 ci) SKK + t
#pragma unroll
        for (int u = 0; u < (PRE < SCOLS ? PRE : SCOLS); ++u) {
          const int t = u / SCQ, ci = u % SCQ;
          const float v = (((inb_mask >> u) & 1) ? (float)raw[u] : 0.f) + 0.1f;
          vrow[kTmWaves * ci * SKK + t] = v;
          n2 = fmaf(v, v, n2);
        }
        // wide layers: the columns past the prefetch, sixteen loads in flight at a time
        constexpr int GU = 16;
        const uint32_t qplane = (uint32_t)q * plane;
#pragma unroll
        for (int u0 = PRE; u0 < SCOLS; u0 += GU) {
          XT more[GU];
          uint32_t inb = 0;
#pragma unroll
          for (int k = 0; k < GU; ++k) {
            const int u = u0 + k;
            if (u < SCOLS) {
              const int t = u / SCQ, ci = u % SCQ;
              bool inside = false;
              const uint32_t off = tap_offset(Walk{t, t / SK, t % SK, 0}, inside) + qplane;
              more[k] = x[off + (uint32_t)(kTmWaves * ci) * plane];
              inb |= (uint32_t)inside << k;
            }
          }
#pragma unroll
          for (int k = 0; k < GU; ++k) {
            const int u = u0 + k;
            if (u < SCOLS) {
              const int t = u / SCQ, ci = u % SCQ;
              const float v = (((inb >> k) & 1) ? (float)more[k] : 0.f) + 0.1f;
              vrow[kTmWaves * ci * SKK + t] = v;
              n2 = fmaf(v, v, n2);
            }
          }
        }
      } else {
        Walk w{q < tc.C ? 0 : KK, 0, 0, q};
#pragma unroll
        for (int u = 0; u < PRE; ++u) {
          if (w.t < KK) {
            const float v = (((inb_mask >> u) & 1) ? (float)raw[u] : 0.f) + 0.1f;
            s_v[lane * FS + w.c * KK + w.t] = v;
            n2 = fmaf(v, v, n2);
            advance(w);
          }
        }
        // wide layers (more than 4 PRE patch columns): the remaining columns, sixteen loads in flight at a time
        constexpr int GU = 16;
        w = rest;
        while (w.t < KK) {
          XT more[GU];
          uint32_t inb = 0;
          Walk wl = w;
          bool inside = false;
          uint32_t off = tap_offset(w, inside);
#pragma unroll
          for (int u = 0; u < GU; ++u) {
            if (w.t < KK) {
              more[u] = x[off + (uint32_t)w.c * plane];
              inb |= (uint32_t)inside << u;
              if (advance(w) && w.t < KK) off = tap_offset(w, inside);
            }
          }
#pragma unroll
          for (int u = 0; u < GU; ++u) {
            if (wl.t < KK) {
              const float v = (((inb >> u) & 1) ? (float)more[u] : 0.f) + 0.1f;
              s_v[lane * FS + wl.c * KK + wl.t] = v;
              n2 = fmaf(v, v, n2);
              advance(wl);
            }
          }
        }
      }
      if (q == 0) s_v[lane * FS + F] = 0.5f;
      s_n2[q * kTmTile + lane] = n2;
#pragma unroll
      for (int cu = 0; cu < GC; ++cu) {
        double g = graw[cu];
        if constexpr (BN) {   // through the BatchNorm behind the convolution: three per-channel coefficients
          const int c = q + kTmWaves * cu;   // (< CO)
          g = fma(s_coef[c], g, fma(s_coef[CO + c], yraw[cu], s_coef[2 * CO + c]));
        }
        s_t[lane * TS + q + kTmWaves * cu] = ((glive_mask >> cu) & 1) ? (float)g * tc.post_scale : 0.f;
      }
    }
    __syncthreads();
    mark(0);
    __builtin_amdgcn_sched_barrier(0);
    // ---- (1) a = v rt for this wave's 16 pixels; epilogue in the C/D layout: row = 4 (lane >> 4) + reg, col = lane & 15
    {
      f32x4 acc1[NBC];
#pragma unroll
      for (int nb = 0; nb < NBC; ++nb) acc1[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* __restrict__ arow = s_v + (size_t)(16 * q + l15) * FS + l4;
      const float* __restrict__ brow = s_rt + (size_t)l4 * RS + l15;
      // chunks of four k-steps (16 patch columns; the columns / rows up to the next multiple of 16 exist and are zero):
      // the four fragments are fetched together, then the MFMAs issue back to back
      for (int k0 = 0; k0 < FC; k0 += 16) {
        float a[4], bfr[4][NBC];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          a[s4] = arow[k0 + 4 * s4];
#pragma unroll
          for (int nb = 0; nb < NBC; ++nb) bfr[s4][nb] = brow[(size_t)(k0 + 4 * s4) * RS + nb * 16];
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
          for (int nb = 0; nb < NBC; ++nb)
            acc1[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s4], bfr[s4][nb], acc1[nb], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = 16 * q + 4 * l4 + i;
        float nrm2 = tc.pad_norm2;
#pragma unroll
        for (int w = 0; w < kTmWaves; ++w) nrm2 += s_n2[w * kTmTile + m];
        const float inv = 1.0f / sqrtf(nrm2);
        float dotp = 0.f;
#pragma unroll
        for (int rb = 0; rb < NBRE; ++rb) {
          float ar, ai;
          int c;
          bool re_lane = true;
          if constexpr (CO >= 16) {
            c = rb * 16 + l15;
            ar = acc1[rb][i] * inv;
            ai = acc1[NBRE + rb][i] * inv;
          } else {  // CO == 8: columns 0..7 real parts, 8..15 imaginary parts of the same channels
            c = l15 & 7;
            re_lane = l15 < 8;
            const float own = acc1[0][i];
            const float other = xlane<8>(own, lane);
            ar = (re_lane ? own : other) * inv;
            ai = (re_lane ? other : own) * inv;
          }
          const float p2 = ar * ar + ai * ai;
          const float t = (p2 * tc.post_scale <= 1.0f) ? s_t[m * TS + c] : 0.f;
          if (re_lane) dotp = fmaf(2.0f * t, p2, dotp);
          if constexpr (CO >= 16) {
            s_w[m * WS + c] = t * ar;
            s_w[m * WS + CO + c] = t * ai;
            s_w3[m * WS + c] = t * ar * inv;
            s_w3[m * WS + CO + c] = t * ai * inv;
          } else {
            const float wv = re_lane ? t * ar : t * ai;
            s_w[m * WS + l15] = wv;        // column l15 = c (real) or CO + c (imaginary)
            s_w3[m * WS + l15] = wv * inv;
          }
        }
        const float dot = group_sum<float, 4>(dotp, lane);   // over the 16 lanes of the row group
        if (l15 == 0) {
          s_dot[m] = dot;
          s_inv[m] = inv;
        }
      }
    }
    __syncthreads();
    mark(1);
    __builtin_amdgcn_sched_barrier(0);
    // ---- (2) g = W2 rt^T: units of 32 pixels x 32 features dealt round-robin to the waves.  Pixels are the M
    //      dimension, so in the C/D layout (col = lane & 31 = feature, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) =
    //      pixel) a lane holds four CONSECUTIVE pixels per register quad: one 16-byte store per quad into the
    //      pixel-contiguous gfeat_t (a quarter of the store instructions of a dword-per-lane epilogue; the four quads
    //      of the two lane halves complete every 128-byte row segment)
    if (tc.wpix != nullptr) {
      // the per-pixel rows of qconv_dx_kernel instead (qsim_qconv_dx.h): the fold commutes with this product, so dL/dx
      // is made from 2 CO + 1 floats per pixel and the F x M feature gradients are never written
      float* __restrict__ wrow = tc.wpix + (size_t)m_base * K2;
      for (int i = tid; i < kTmTile * K2; i += kTmThreads) {
        const int m = i / K2, cc = i - m * K2;
        if (m_base + m < tc.M) wrow[i] = 2.0f * s_w3[m * WS + cc];
      }
      if (tid < kTmTile && m_base + tid < tc.M)
        tc.wpix[(size_t)tc.M * K2 + m_base + tid] = s_dot[tid] * s_inv[tid] * s_inv[tid];
    } else {
      const int n_units = 2 * ((F + 31) / 32);
      const bool vec_ok = (tc.M & 3) == 0;
      for (int unit = q; unit < n_units; unit += kTmWaves) {
        const int jb = unit >> 1, pb = unit & 1;
        const float* __restrict__ arow = s_w + (size_t)(pb * 32 + l31) * WS + l5;     // A[i = pixel][k = cc]
        const float* __restrict__ brow = s_rt + (size_t)(jb * 32 + l31) * RS + l5;    // B[k = cc][j = feature]
        const int j = jb * 32 + l31;
        const int jr = j < F ? j : F;
        f32x16 acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll
        for (int k0 = 0; k0 < K2; k0 += 2) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[k0], brow[k0], acc2, 0, 0, 0);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int p0 = pb * 32 + 8 * g4 + 4 * l5;
          const f32x4 inv4 = *reinterpret_cast<const f32x4*>(s_inv + p0);
          const f32x4 dot4 = *reinterpret_cast<const f32x4*>(s_dot + p0);
          f32x4 out;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            out[e] = 2.0f * acc2[4 * g4 + e] * inv4[e] - s_v[(p0 + e) * FS + jr] * (dot4[e] * inv4[e] * inv4[e]);
          const int64_t m0 = m_base + p0;
          if (j < F) {
            float* __restrict__ dst = gfeat_t + (size_t)j * tc.M + m0;
            if (vec_ok && m0 + 3 < tc.M) {
              *reinterpret_cast<f32x4*>(dst) = out;
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (m0 + e < tc.M) dst[e] = out[e];
            }
          }
        }
      }
    }
    mark(2);
    __builtin_amdgcn_sched_barrier(0);
    // the next tile's global loads go out now (behind this tile's stores in the memory queue) and land while
    // product (3) runs and the workgroup crosses the barrier
    if (tile + gridDim.x < tiles) issue_gather(tile + gridDim.x);
    mark(4);
    __builtin_amdgcn_sched_barrier(0);
    // ---- (3) h += (W2 / |v|)^T v over the tile's 64 pixels: this wave's 16-column blocks; chunks of two k-steps
    //      (8 pixels), the fragments of chunk c + 1 in flight while the MFMAs of chunk c issue
    {
      float a[2][2][NBC], bv[2][2][JBMAX];
      auto fetch3 = [&](int buf, int k0) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int row = k0 + 4 * s2 + l4;
#pragma unroll
          for (int mb = 0; mb < NBC; ++mb) a[buf][s2][mb] = s_w3[(size_t)row * WS + mb * 16 + l15];
#pragma unroll
          for (int u = 0; u < JBMAX; ++u) {   // blocks beyond n_jb read block 0 (any valid address) and are not used
            const int jb = q + kTmWaves * u;
            bv[buf][s2][u] = s_v[(size_t)row * FS + (jb < n_jb ? jb : 0) * 16 + l15];
          }
        }
      };
      auto fma3 = [&](int buf) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int u = 0; u < JBMAX; ++u) {
            if (q + kTmWaves * u < n_jb) {
#pragma unroll
              for (int mb = 0; mb < NBC; ++mb)
                acch[u][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[buf][s2][mb], bv[buf][s2][u], acch[u][mb], 0, 0, 0);
            }
          }
      };
      fetch3(0, 0);
#pragma unroll
      for (int c = 0; c < 8; c += 2) {
        fetch3(1, 8 * (c + 1));
        fma3(0);
        __builtin_amdgcn_sched_barrier(0);   // keep the two-deep pipeline: no hoisting of later fetches (register budget)
        if (c + 2 < 8) fetch3(0, 8 * (c + 2));
        fma3(1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    mark(3);
  }
  if (stamp) {
    for (int i = 0; i < 5; ++i) tc.stamps[i] = st_sum[i];
    tc.stamps[5] = (unsigned long long)((tiles - blockIdx.x + gridDim.x - 1) / gridDim.x);
  }
  // one slab per workgroup: hpart[block][cc][j], j <= F (column F = the value every pad column shares)
  float* __restrict__ hp = hpart + (size_t)blockIdx.x * K2 * (F + 1);
#pragma unroll
  for (int u = 0; u < JBMAX; ++u) {
    const int jb = q + kTmWaves * u;
    const int j = jb * 16 + l15;
    if (jb < n_jb && j <= F) {
#pragma unroll
      for (int mb = 0; mb < NBC; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) hp[(size_t)(mb * 16 + 4 * l4 + r) * (F + 1) + j] = acch[u][mb][r];
    }
  }
}

}  // namespace qiddm
