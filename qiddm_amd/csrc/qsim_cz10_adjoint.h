// qsim_cz10_adjoint.h -- reverse-mode gradients of a 10-qubit CZ circuit (no / RZ encoding), register-resident.
//
// Replaces ``diff_method="backprop"`` (torch autograd through default.qubit.torch, reference nn/qdense.py:419, 458-472)
// for BASELINE config 3's circuit family: differN_noise(28, L, N) -- 10 wires, L blocks of [RZ(x) ; SEL(2 layers, CZ)],
// probabilities -- one QNode round per launch.  adjoint_kernel<T, 10> walks the round gate by gate (its folded reverse
// sweep spills at 16 amplitudes of psi AND lambda per lane) and needs 256 + 212 registers: one wave per SIMD.  Here the
// folded un-application of qsim_wide_cz_adjoint.h runs on the whole 2^10-amplitude state as ONE tile that never leaves
// the registers (the n - 10 = 0 tile bits of that geometry): per layer
//     [real RY^dagger on all 10 positions, d/dtheta from the pairs]  .  [conj diagonal, d/dalpha as signed sums]
// on psi and lambda together, wave-uniform tables in scalar registers, the first layer's gradients from the
// product-state contraction.  One wavefront per sample, no workgroup barrier inside a sample, fixed-order sums.
// Output: the slab layout adjoint_finalize_folded_kernel reads ([layer][theta | alpha][16]) and grad_inputs (B, 10).
#pragma once
#include "qsim_wide_cz_adjoint.h"

namespace qiddm {

constexpr int kCz10Waves = 4;

template <typename T>
struct Cz10AdjSmem {
  // [ry][ua]: L*10 complex each; per wave: ux 16 complex, then doubles row 48, gin 16, g 16, acc L*32
  __host__ __device__ static size_t bytes(int64_t layers) {
    return (size_t)layers * 10 * 2 * 2 * sizeof(T) + (size_t)kCz10Waves * 16 * 2 * sizeof(T) +
           (size_t)kCz10Waves * (48 + 16 + 16 + (size_t)layers * 32) * sizeof(double);
  }
};

template <typename T>
__global__ __launch_bounds__(kCz10Waves* kWave, (sizeof(T) == 4 ? 2 : 1)) void cz10_adjoint_kernel(const T* __restrict__ inputs,
                                                                          const T* __restrict__ tail,
                                                                          const T* __restrict__ gout,
                                                                          T* __restrict__ partials, int64_t slab_stride,
                                                                          T* __restrict__ grad_inputs, int64_t gin_ld,
                                                                          const KScalars p) {
  constexpr int N = 10, R = 16;
  using W = WideCzAdj<T, N>;
  using B = WideCz<T, N>;
  using C = V2<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int L = p.n_blocks * p.sel_layers;
  C* s_ry = reinterpret_cast<C*>(smem_raw);
  C* s_ua = s_ry + (size_t)L * N;
  C* s_ux_all = s_ua + (size_t)L * N;
  double* s_dbl = reinterpret_cast<double*>(s_ux_all + kCz10Waves * 16);
  const int tid = threadIdx.x;
  W w;
  w.lane = tid & (kWave - 1);
  w.llane = logical_lane(w.lane);
  w.wave = tid >> 6;
  w.waves = blockDim.x >> 6;
  w.eng.lane = w.lane;
  w.eng.llane = w.llane;
  w.eng.sub = w.llane;
  w.s_ry = s_ry;
  w.s_ua = s_ua;
  C* s_ux = s_ux_all + w.wave * 16;
  w.s_ux = s_ux;
  w.n_layers_round = L;
  const size_t per_wave = 48 + 16 + 16 + (size_t)L * 32;
  double* row = s_dbl + w.wave * per_wave;   // [48]: theta of the layer, alpha of the layer, theta of layer 0
  double* s_gin = row + 48;                  // [16]
  double* s_g = s_gin + 16;                  // [16]
  double* s_acc = s_g + 16;                  // [L][2][16]
  for (int i = tid; i < L * 2 * N; i += blockDim.x) {
    const int l = i / (2 * N), e = i - l * 2 * N;
    const C v = C{tail[2 * (size_t)i], tail[2 * (size_t)i + 1]};
    if (e < N) s_ry[l * N + e] = v;
    else s_ua[l * N + (e - N)] = v;
  }
  for (int i = w.lane; i < L * 32; i += kWave) s_acc[i] = 0.0;
  __syncthreads();

  auto wave_sync = [&]() {   // LDS hand-over inside the wavefront
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // row -> accumulators: kind 0 = theta of `layer`, kind 1 = alpha of `layer` (and the input gradient at a block start),
  // kind 2 = theta of layer 0
  auto flush = [&](int layer, bool has01, bool block_start, bool has2) {
    wave_sync();
    if (w.lane < 48) {
      const int kind = w.lane >> 4, wire = w.lane & 15;
      const double v = row[w.lane];
      if (wire < N) {
        if (kind == 0 && has01) s_acc[(layer * 2 + 0) * 16 + wire] += v;
        if (kind == 1 && has01) {
          s_acc[(layer * 2 + 1) * 16 + wire] += v;
          if (block_start) s_gin[wire] += v;
        }
        if (kind == 2 && has2) s_acc[(0 * 2 + 0) * 16 + wire] += v;
      }
      row[w.lane] = 0.0;
    }
    wave_sync();
  };
  for (int i = w.lane; i < 48; i += kWave) row[i] = 0.0;

  const int64_t total_waves = (int64_t)gridDim.x * w.waves;
  for (int64_t sample = (int64_t)blockIdx.x * w.waves + w.wave; sample < p.batch; sample += total_waves) {
    const T* g_row = gout + sample * p.g_ld;
    if (w.lane < 16) {
      s_gin[w.lane] = 0.0;
      s_g[w.lane] = (p.measure == 1 && w.lane < N) ? (double)g_row[w.lane] : 0.0;
      double xs = (p.encoding == 2 && w.lane < N) ? (double)inputs[sample * p.in_ld + w.lane] * p.enc_scale : 0.0;
      double s, c;
      sincos(0.5 * xs, &s, &c);
      s_ux[w.lane] = C{(T)c, (T)s};
    }
    wave_sync();
    // ---- forward: the product state of layer 0, then D^l . RY^l for l = 1 .. L-1 ---------------------------------
    C a[R], l[R];
    {
      typename B::Init in;
      w.template build_init<0>(0, in);
      w.gen_tile(in, 0, a);
    }
    for (int li = 1; li < L; ++li) {
      typename B::Diag d;
      w.template build_diag<0>(li, p.encoding == 2 && (li % p.sel_layers == 0), (((li - 1) % p.sel_layers) % (N - 1)) + 1, d);
      w.template apply_diag<0>(a, d, 0);
      w.template ry_from<0, 0>(a, li);
    }
    // ---- lambda = dL/d(psi*) ------------------------------------------------------------------------------------
    if (p.measure == 1) {
      T g0 = 0, g_lane = 0;
#pragma unroll
      for (int wi = 0; wi < N; ++wi) g0 += (T)s_g[wi];
#pragma unroll
      for (int j = 1; j <= 6; ++j) g_lane += ((w.llane >> (j - 1)) & 1) ? (T)s_g[N - 1 - j] : (T)0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        T gr = ((r & 1) ? (T)s_g[N - 1 - 0] : (T)0);
#pragma unroll
        for (int rb = 1; rb < 4; ++rb) gr += ((r >> rb) & 1) ? (T)s_g[N - 1 - (6 + rb)] : (T)0;
        const T ge = g0 - (T)2 * (g_lane + gr);
        l[r] = C{ge * a[r].x, ge * a[r].y};
      }
    } else {
#pragma unroll
      for (int r1 = 0; r1 < R / 2; ++r1) {
        using T2 = V2<T>;
        const T2 gg = *reinterpret_cast<const T2*>(g_row + w.template pair_index<0>(0, r1));
        l[2 * r1] = C{gg.x * a[2 * r1].x, gg.x * a[2 * r1].y};
        l[2 * r1 + 1] = C{gg.y * a[2 * r1 + 1].x, gg.y * a[2 * r1 + 1].y};
      }
    }
    // ---- reverse: layers L-1 .. 1 ---------------------------------------------------------------------------------
    for (int li = L - 1; li >= 1; --li) {
      T th[10];
#pragma unroll
      for (int i = 0; i < 10; ++i) th[i] = 0;
      w.template undo_down_to<0, 9, 0>(a, l, li, th);
      typename W::Signed m;
      m.tile[0] = 0;
      w.alpha_tile(a, l, 0, m);
      const bool bs = p.encoding == 2 && (li % p.sel_layers == 0);
      {
        typename B::Diag d;
        w.template build_diag<0>(li, bs, (((li - 1) % p.sel_layers) % (N - 1)) + 1, d);
        w.template undo_diag<0>(a, l, d, 0, W::signed_token(m) + th[0]);
      }
      w.template put_theta<0>(row, 0, th, 0);
      w.template put_signed<0>(row, 16, m);
      flush(li, true, bs, false);
    }
    // ---- layer 0: d/dtheta_w = Re (RY^dagger lambda)[e_w] --------------------------------------------------------
    {
      C cs_lane[6], cs_reg[4];
      T fl = 1, fl_ex[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        cs_lane[j] = s_ry[N - 1 - (j + 1)];
        fl *= ((w.llane >> j) & 1) ? cs_lane[j].y : cs_lane[j].x;
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        T v = 1;
#pragma unroll
        for (int j2 = 0; j2 < 6; ++j2) {
          const bool b = (w.llane >> j2) & 1;
          v *= j2 == j ? (b ? cs_lane[j2].x : -cs_lane[j2].y) : (b ? cs_lane[j2].y : cs_lane[j2].x);
        }
        fl_ex[j] = v;
      }
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) cs_reg[rb] = wide_uniform2<T>(s_ry[N - 1 - (rb == 0 ? 0 : 6 + rb)]);
      T pl[8], ex[4][8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        pl[r] = cs_reg[0].x * l[2 * r].x + cs_reg[0].y * l[2 * r + 1].x;
        ex[0][r] = cs_reg[0].x * l[2 * r + 1].x - cs_reg[0].y * l[2 * r].x;
      }
#pragma unroll
      for (int k = 1; k < 4; ++k) {
        const int cnt = 8 >> k;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          if (r < cnt) {
            const T p0 = pl[2 * r], p1 = pl[2 * r + 1];
            ex[k][r] = cs_reg[k].x * p1 - cs_reg[k].y * p0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (j < k) ex[j][r] = cs_reg[k].x * ex[j][2 * r] + cs_reg[k].y * ex[j][2 * r + 1];
            pl[r] = cs_reg[k].x * p0 + cs_reg[k].y * p1;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const T v = group_sum<T, 6>(fl_ex[j] * pl[0], w.lane);
        if (w.lane == 0) row[32 + (N - 1 - (j + 1))] = (double)v;
      }
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        const T v = group_sum<T, 6>(fl * ex[rb][0], w.lane);
        if (w.lane == 0) row[32 + (N - 1 - (rb == 0 ? 0 : 6 + rb))] = (double)v;
      }
      flush(0, false, false, true);
    }
    if (grad_inputs != nullptr && w.lane < N) grad_inputs[sample * gin_ld + w.lane] = (T)(s_gin[w.lane] * p.enc_scale);
    wave_sync();
  }
  // one slab per workgroup: the waves' accumulators in a fixed order
  __syncthreads();
  T* __restrict__ slab = partials + (size_t)blockIdx.x * slab_stride;
  for (int i = tid; i < L * 32; i += blockDim.x) {
    double v = 0.0;
    for (int wv = 0; wv < w.waves; ++wv) v += s_dbl[wv * per_wave + 48 + 16 + 16 + i];
    slab[i] = (T)v;
  }
}

}  // namespace qiddm
