// qsim_quad.h -- low-latency dense-net sampler for small batches: FOUR wavefronts per sample.
//
// At BASELINE config 2 (batch 256, n = 8) the one-wave-per-sample kernels leave three of every four
// SIMDs idle and the step is pure latency (phase stamps).  Here one 256-thread
// workgroup owns a sample: amplitude index k = (r << 8) | (wave << 6) | lane, i.e. index bits 6 and 7
// live in the wave number.  The circuit runs on folded per-layer tables (below): one phase multiply per amplitude,
// then real RY gates -- lane bits fetch their partner with a DPP / permlane move (xlane2), register bits pair in
// registers, and the two wave-bit gates of a layer are applied TOGETHER as one real 4x4 exchange through a
// double-buffered LDS slab (one s_barrier per layer).  linear_down / linear_up are split over all 256
// threads, and the whole sampling loop x <- net(x) (reference src/models.py:124-136) can run for
// `n_steps` iterations inside the launch with x held in registers.
//
// Restricted to what the dense nets need: RZ data re-uploading, CZ rings, <Z> read-out, 2 <= n <= 10.
// n = 6, 7 (the reference's own `QIDDM_LL_noise(784, 6, 14, 2)`, src/mnist_exm.py:46): the state fits one wavefront
// (k = (r << 6) | lane), so every wave runs the whole circuit on its own copy -- no wave-bit exchange, no barrier
// inside the layer loop -- and the four waves still share the two linears.  Below 6 qubits (C1's 4-qubit nets) the
// spare lane bits hold further copies of the state.
#pragma once
#include "qsim_adjoint.h"  // wave_reduce8_into
#include "qsim_fused.h"

namespace qiddm {

struct QuadScalars {
  int64_t x_ld, y_ld, y_step_stride;  // y: (n_steps, batch, out_features)
  int32_t in_features, out_features;
  int32_t post_mode, n_steps;
  double noise_factor;
  unsigned long long* stamps;  // diagnostics only (tools/stamp_dense.py)
};

// Rot = RZ(omega) RY(theta) RZ(phi), and everything between two RY layers is diagonal:
//     RZ(phi^{l+1}) . CZ-ring^l . RZ(omega^l)        (x the data re-upload diagonal at a block start)
// so the circuit runs as  [phase table] -> [RY on every wire] -> [phase table] -> ...  The phase of amplitude k
// in front of layer l depends on the weights only; it is tabulated once per launch:
//     t_lo[l][k & 255]  (bits 0..7, one entry per thread; for n = 8 the CZ sign is folded in)
//     t_hi[l][k >> 8]   (register bits 8.., n > 8; the CZ sign then comes from the parity bits as before)
// The diagonal after the last RY layer does not reach |amplitude|^2 and is dropped.  An RY has real entries:
// a lane-bit gate is  c * own +- s * partner  (one packed multiply, one packed FMA, two floats of gate data).
// index bits owned by the thread number: lane bits 0..5, plus the wave bits 6, 7 from 8 qubits on
template <int N>
__host__ __device__ constexpr int quad_thread_bits() { return N >= 8 ? 8 : (N >= 6 ? 6 : N); }
// index bits of a thread: wave bits (8+ qubits) and the low LOGICAL lane bits; below 6 qubits the remaining lane bits
// (like the waves below 8) hold further copies of the state
template <int N>
__device__ __forceinline__ uint32_t quad_kbase(int wv, int llane) {
  constexpr int TB = quad_thread_bits<N>();
  return (TB == 8 ? ((uint32_t)wv << 6) : 0u) | ((uint32_t)llane & ((1u << (TB < 6 ? TB : 6)) - 1u));
}

template <typename T, int N>
struct QuadSmem {
  static constexpr int R = 1 << (N - quad_thread_bits<N>());
  __host__ __device__ static size_t ry_bytes(int64_t n_rot) { return (size_t)n_rot * 2 * sizeof(T); }
  static constexpr int TL = 1 << quad_thread_bits<N>();  // phase-table entries per layer (one per thread-owned index)
  __host__ __device__ static size_t tlo_bytes(int64_t n_rot) { return (size_t)(n_rot / N) * TL * 2 * sizeof(T); }
  __host__ __device__ static size_t thi_bytes(int64_t n_rot) {
    return ((size_t)(n_rot / N) * R * 2 * sizeof(T) + 15) / 16 * 16;
  }
  static constexpr size_t kCzBytes = (size_t)(N - 1) * 256 * 4;
  static constexpr size_t kSlabBytes = (size_t)2 * 4 * R * kWave * 2 * sizeof(T);
  static constexpr size_t kMiscBytes = (2 * 4 * 16 + 3 * 4 * 16) * sizeof(double);   // two partial buffers; xs / cs / sn per wave
  __host__ __device__ static size_t bytes(int64_t n_rot) {
    return (ry_bytes(n_rot) + 15) / 16 * 16 + tlo_bytes(n_rot) + thi_bytes(n_rot) + kCzBytes + kSlabBytes + kMiscBytes +
           (size_t)n_rot * sizeof(double);  // staging scratch: the summed RZ angle per (layer, wire)
  }
};

// RY(theta) on lane bit Q: own' = c own +- s partner ('+' on the lanes whose bit is set)
template <int Q, typename T, int R>
__device__ __forceinline__ void ry_lane(V2<T> (&a)[R], T c, T s_signed, int lane) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const V2<T> par = xlane2<(1 << Q), T>(a[r], lane);
    a[r] = __builtin_elementwise_fma(bcast<T>(s_signed), par, bcast<T>(c) * a[r]);
  }
}
// RY(theta) on lane bit 4 or 5 without a partner fetch: ONE permlane swap of (re, im) puts the real parts of both pair
// members into the lanes with the bit clear and their imaginary parts into the lanes with it set (the swap exchanges the
// set-bit half of its first operand with the clear-bit half of its second).  An RY is real and the same 2 x 2 on both
// parts, so the gate is in-register and lane-uniform -- lo' = c lo - s hi, hi' = s lo + c hi, the same products as the
// partner form -- and a second swap puts everything back: 4 instructions in the dependent chain of a layer instead of
// 10 (two copies, the swap and a select per component for the partner's value).
template <int Q>
__device__ __forceinline__ void permlane_swap_bit(uint32_t& x, uint32_t& y) {
  static_assert(Q == 4 || Q == 5, "row-crossing lane bits");
  if constexpr (Q == 5) {
    const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    x = r[0], y = r[1];
  } else {
    const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    x = r[0], y = r[1];
  }
}
template <int Q>
__device__ __forceinline__ void swap_parts(float& x, float& y) {
  uint32_t ux = __float_as_uint(x), uy = __float_as_uint(y);
  permlane_swap_bit<Q>(ux, uy);
  x = __uint_as_float(ux), y = __uint_as_float(uy);
}
template <int Q>
__device__ __forceinline__ void swap_parts(double& x, double& y) {
  uint32_t xl = (uint32_t)__double2loint(x), xh = (uint32_t)__double2hiint(x);
  uint32_t yl = (uint32_t)__double2loint(y), yh = (uint32_t)__double2hiint(y);
  permlane_swap_bit<Q>(xl, yl);
  permlane_swap_bit<Q>(xh, yh);
  x = __hiloint2double((int)xh, (int)xl), y = __hiloint2double((int)yh, (int)yl);
}
template <int Q, typename T, int R>
__device__ __forceinline__ void ry_lane_xy(V2<T> (&a)[R], T c, T s) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    T lo = a[r].x, hi = a[r].y;
    swap_parts<Q>(lo, hi);   // (lo member, hi member) of the real parts / of the imaginary parts
    T nlo = fma(-s, hi, c * lo), nhi = fma(s, lo, c * hi);
    swap_parts<Q>(nlo, nhi);
    a[r] = V2<T>{nlo, nhi};
  }
}
// RY(theta) between register pairs (r, r | J)
template <int J, typename T, int R>
__device__ __forceinline__ void ry_regs(V2<T> (&a)[R], T c, T s) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if ((r & J) == 0) {
      const V2<T> a0 = a[r], a1 = a[r | J];
      a[r] = __builtin_elementwise_fma(bcast<T>(-s), a1, bcast<T>(c) * a0);
      a[r | J] = __builtin_elementwise_fma(bcast<T>(s), a0, bcast<T>(c) * a1);
    }
  }
}

// what one thread needs for one layer, fetched a layer ahead
template <typename T, int N>
struct QuadLayerData {
  using C = V2<T>;
  static constexpr int R = 1 << (N - quad_thread_bits<N>());
  C ry[N];     // (cos, sin)(theta / 2) per WIRE
  C tlo;       // this thread's phase
  C thi[R];    // register-bit phases (n > 8)
  uint32_t cz; // parity bits of the PREVIOUS layer's ring (n > 8)

  // this thread's phase alone (the first thing a layer needs)
  __device__ __forceinline__ static C load_tlo(const C* s_tlo, int layer, int tid) {
    constexpr int TL = 1 << quad_thread_bits<N>();
    return s_tlo[layer * TL + (quad_thread_bits<N>() < 6 ? (int)quad_kbase<N>(0, logical_lane(tid & 63)) : (tid & (TL - 1)))];
  }
  // everything but tlo
  __device__ __forceinline__ void load_rest(const C* s_ry, const C* s_thi, const uint32_t* s_cz, int layer,
                                            int prev_range, int tid) {
#pragma unroll
    for (int w = 0; w < N; ++w) ry[w] = s_ry[layer * N + w];
    if constexpr (R > 1) {
#pragma unroll
      for (int r = 0; r < R; ++r) thi[r] = s_thi[layer * R + r];
      cz = prev_range >= 0 ? s_cz[prev_range * 256 + tid] : 0u;
    }
  }
  __device__ __forceinline__ void load(const C* s_ry, const C* s_tlo, const C* s_thi, const uint32_t* s_cz, int layer,
                                       int prev_range, int tid) {
#pragma unroll
    for (int w = 0; w < N; ++w) ry[w] = s_ry[layer * N + w];
    constexpr int TL = 1 << quad_thread_bits<N>();
    tlo = s_tlo[layer * TL + (quad_thread_bits<N>() < 6 ? (int)quad_kbase<N>(0, logical_lane(tid & 63)) : (tid & (TL - 1)))];
    if constexpr (R > 1) {
#pragma unroll
      for (int r = 0; r < R; ++r) thi[r] = s_thi[layer * R + r];
      cz = prev_range >= 0 ? s_cz[prev_range * 256 + tid] : 0u;
    }
  }
};

// RY coefficients and phase tables of the quad layout from the raw angles (256 threads; `alpha` is LDS scratch of
// n_rot doubles).  ry / tlo / thi may point to LDS (built per launch) or to global memory (built once per weights).
template <typename T, int N>
__device__ __forceinline__ void quad_build_tables(const double* __restrict__ angles, int n_rot, int layers_per_round,
                                                  int sel_layers, V2<T>* ry, V2<T>* tlo, V2<T>* thi, double* alpha,
                                                  int tid, uint32_t kbase) {
  using C = V2<T>;
  constexpr int TB = quad_thread_bits<N>();
  constexpr int R = 1 << (N - TB);
  const int n_layers_all = n_rot / N;
  for (int g = tid; g < n_rot; g += 256) {
    double c, sn;
    table_sincos<T>(0.5 * angles[g * 3 + 1], &sn, &c);
    ry[g] = C{(T)c, (T)sn};
    // the RZ(omega) of the layer before (same round) merges with this layer's RZ(phi)
    const int li = (g / N) % layers_per_round;
    alpha[g] = angles[g * 3 + 0] + (li > 0 ? angles[(g - N) * 3 + 2] : 0.0);
  }
  __syncthreads();
  for (int l = 0; l < n_layers_all; ++l) {
    const int li = l % layers_per_round;
    // RZ(alpha) = diag(e^{-i alpha/2}, e^{+i alpha/2}): the phases of all wires add up to one angle
    double ang = 0.0;
#pragma unroll
    for (int q = 0; q < TB; ++q) {
      const double al = alpha[l * N + (N - 1 - q)];
      ang += ((kbase >> q) & 1u) ? 0.5 * al : -0.5 * al;
    }
    double c, sn;
    table_sincos<T>(ang, &sn, &c);
    if constexpr (R == 1) {
      if (li > 0 && cz_ring_parity<N>(kbase, ((li - 1) % sel_layers) % (N - 1) + 1)) {
        c = -c;
        sn = -sn;
      }
    }
    constexpr int TL = 1 << TB;
    // one writer per entry: thread tid owns index kbase (== its slot for TB >= 6; below, the first copy writes)
    if (tid < 64 || TB == 8) {
      if (TB >= 6) {
        if (tid < TL) tlo[l * TL + tid] = C{(T)c, (T)sn};
      } else if ((logical_lane(tid) >> TB) == 0) {
        tlo[l * TL + (int)kbase] = C{(T)c, (T)sn};
      }
    }
  }
  if constexpr (R > 1) {
    for (int i = tid; i < n_layers_all * R; i += 256) {
      const int l = i / R, r = i % R;
      double ang = 0.0;
#pragma unroll
      for (int j = 0; j < N - TB; ++j) {
        const double al = alpha[l * N + (N - 1 - (TB + j))];
        ang += ((r >> j) & 1) ? 0.5 * al : -0.5 * al;
      }
      double c, sn;
      table_sincos<T>(ang, &sn, &c);
      thi[i] = C{(T)c, (T)sn};
    }
  }
}

// the tables as dense_quad_kernel lays them out in LDS ([ry | tlo | thi]), written once per weights
template <typename T, int N>
__global__ __launch_bounds__(256) void quad_tables_kernel(const double* __restrict__ angles, T* __restrict__ tables,
                                                          const KScalars p) {
  using C = V2<T>;
  using QS = QuadSmem<T, N>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int n_rot = p.n_rounds * p.n_blocks * p.sel_layers * N;
  const int tid = threadIdx.x;
  const uint32_t kbase = quad_kbase<N>(tid >> 6, logical_lane(tid & 63));
  unsigned char* base = reinterpret_cast<unsigned char*>(tables);
  C* ry = reinterpret_cast<C*>(base);
  C* tlo = reinterpret_cast<C*>(base + (QS::ry_bytes(n_rot) + 15) / 16 * 16);
  C* thi = reinterpret_cast<C*>(base + (QS::ry_bytes(n_rot) + 15) / 16 * 16 + QS::tlo_bytes(n_rot));
  quad_build_tables<T, N>(angles, n_rot, p.n_blocks * p.sel_layers, p.sel_layers, ry, tlo, thi,
                          reinterpret_cast<double*>(smem_raw), tid, kbase);
}

template <typename T, int N, int PPT>
__global__ __launch_bounds__(256) void dense_quad_kernel(
    const double* __restrict__ x, const double* __restrict__ wd, const double* __restrict__ bd,
    const double* __restrict__ angles, const double* __restrict__ wu, const double* __restrict__ bu,
    double* __restrict__ y, const T* __restrict__ tables, const QuadScalars d, const KScalars p) {
  static_assert(N >= 2 && N <= 10, "quad layout: 2..10 qubits");
  using C = V2<T>;
  constexpr int TB = quad_thread_bits<N>();  // 8: index bits 6, 7 live in the wave number; <= 6: every wave holds a copy
  constexpr int R = 1 << (N - TB);
  // PPT pixels per thread (in/out features <= 256 * PPT) stay in registers across the steps
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int n_rot = p.n_rounds * p.n_blocks * p.sel_layers * N;
  const int n_layers_all = n_rot / N;
  using QS = QuadSmem<T, N>;
  unsigned char* cursor = smem_raw;
  C* s_ry = reinterpret_cast<C*>(cursor);
  cursor += (QS::ry_bytes(n_rot) + 15) / 16 * 16;
  C* s_tlo = reinterpret_cast<C*>(cursor);
  cursor += QS::tlo_bytes(n_rot);
  C* s_thi = reinterpret_cast<C*>(cursor);
  cursor += QS::thi_bytes(n_rot);
  uint32_t* s_cz = reinterpret_cast<uint32_t*>(cursor);
  C* s_slab = reinterpret_cast<C*>(reinterpret_cast<unsigned char*>(s_cz) + QS::kCzBytes);
  double* s_part = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(s_slab) + QS::kSlabBytes);
  // Barriers: the step needs the two cross-wave sums (linear_down, <Z>) and nothing else outside the layers.  Every wave
  // combines the four partials ITSELF (lanes < n, redundantly) and keeps its own copy of the round's angles and their
  // sin / cos -- wave-level ordering only -- and the two sums use separate partial buffers, so neither has to wait for the
  // other's readers: 2 + layers barriers per step instead of 7 + layers.
  double* s_part_z = s_part + 4 * 16;  // [4][16] partials of the read-out (s_part: linear_down)
  const int wv_ = (int)(threadIdx.x >> 6);
  double* s_xs = s_part_z + 4 * 16 + wv_ * 16;      // [16] angles of the round, this wave's copy
  double* s_cs = s_part_z + 2 * 4 * 16 + wv_ * 16;  // [16] cos(x/2)   (after the read-out: plain <Z_w>)
  double* s_sn = s_part_z + 3 * 4 * 16 + wv_ * 16;  // [16] sin(x/2)
  double* s_alpha = s_part_z + 4 * 4 * 16;          // [n_rot] staging only: phi^l_w + omega^{l-1}_w

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int llane = logical_lane(lane);
  const bool stamp = d.stamps != nullptr && blockIdx.x == 0 && tid == 0;
  if (stamp) d.stamps[0] = __builtin_amdgcn_s_memtime();
  const int P = d.in_features, Q = d.out_features;

  // ---- issue every global load of the first sample BEFORE staging: their latency hides behind it ----
  // weights of this thread's pixels stay in registers for all samples and steps of the launch
  double xr[PPT], wdr[PPT][N], wur[PPT][N], bur[PPT];
  {
    const int64_t s0 = blockIdx.x < p.batch ? blockIdx.x : 0;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int pix = tid + i * 256;
      xr[i] = pix < P ? x[s0 * d.x_ld + pix] : 0.0;
#pragma unroll
      for (int j = 0; j < N; ++j) wdr[i][j] = pix < P ? wd[(size_t)j * P + pix] : 0.0;
#pragma unroll
      for (int j = 0; j < N; ++j) wur[i][j] = pix < Q ? wu[(size_t)pix * N + j] : 0.0;
      bur[i] = (bu && pix < Q) ? bu[pix] : 0.0;
    }
  }

  // ---- staging: RY coefficients and phase tables (built here, or copied when the caller prepared them once per
  //      weights with quad_tables_kernel), CZ parity bits ------------------------------------------------------
  const int layers_per_round = p.n_blocks * p.sel_layers;
  const uint32_t kbase = quad_kbase<N>(wv, llane);  // index bits 0..TB-1 of this thread
  if (tables != nullptr) {
    const int n_t = (int)((QS::ry_bytes(n_rot) + 15) / 16 * 16 + QS::tlo_bytes(n_rot) + QS::thi_bytes(n_rot)) / (int)sizeof(T);
    T* dst = reinterpret_cast<T*>(s_ry);
    for (int i = tid; i < n_t; i += 256) dst[i] = tables[i];
  } else {
    quad_build_tables<T, N>(angles, n_rot, layers_per_round, p.sel_layers, s_ry, s_tlo, s_thi, s_alpha, tid, kbase);
  }
  if constexpr (R > 1) {
    for (int rr = 1; rr < N; ++rr) {
      uint32_t bits = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) bits |= cz_ring_parity<N>(((uint32_t)r << TB) | kbase, rr) << r;
      s_cz[(rr - 1) * 256 + tid] = bits;
    }
  }
  // +-1 by this thread's index bit: the sign of sin in its row of RY
  T pm[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) pm[q] = ((kbase >> q) & 1u) ? (T)1 : (T)-1;  // (bits 6, 7 unused when TB == 6)
  const double bd_mine = (bd && lane < N) ? bd[lane] : 0.0;
  auto wave_sync = [&]() {   // LDS hand-over inside the wavefront
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  __syncthreads();
  if (stamp) d.stamps[1] = __builtin_amdgcn_s_memtime();

  int xbuf_parity = 0;
  bool first = true;
  for (int64_t sample = blockIdx.x; sample < p.batch; sample += gridDim.x) {
    if (!first) {
#pragma unroll
      for (int i = 0; i < PPT; ++i) {
        const int pix = tid + i * 256;
        xr[i] = pix < P ? x[sample * d.x_ld + pix] : 0.0;
      }
    }
    first = false;
    for (int step = 0; step < d.n_steps; ++step) {
      // ---- linear_down over the whole workgroup -----------------------------------------------------
      double acc[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = 0.0;
#pragma unroll
      for (int i = 0; i < PPT; ++i) {
#pragma unroll
        for (int j = 0; j < N; ++j) acc[j] = fma(xr[i], wdr[i][j], acc[j]);  // padded slots hold zeros
      }
      // (s_part is free: its last readers ran before the barrier of the previous step's read-out)
      {
        double v8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = acc[j];
        wave_reduce8_into<double, true>(v8, lane, llane, s_part + wv * 16);
        if constexpr (N > 8) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v8[j] = acc[8 + j];
          wave_reduce8_into<double, true>(v8, lane, llane, s_part + wv * 16 + 8);
        }
      }
      __syncthreads();
      if (lane < N) {
        // wave_reduce8_into leaves value idx in slot idx of its 8-slot group
        const double h = s_part[lane] + s_part[16 + lane] + s_part[32 + lane] + s_part[48 + lane] + bd_mine;
        s_xs[lane] = h * p.enc_scale;
      }
      if (stamp && step == 0) d.stamps[2] = __builtin_amdgcn_s_memtime();

      // ---- circuit rounds ------------------------------------------------------------------------------
      for (int round = 0; round < p.n_rounds; ++round) {
        if (lane < N) {   // the lane that wrote s_xs[lane] (this wave's copy)
          if constexpr (sizeof(T) == 4) {
            float s, c;
            data_sincos_f32(0.5 * s_xs[lane], &s, &c);
            s_cs[lane] = (double)c;
            s_sn[lane] = (double)s;
          } else {
            double s, c;
            sincos(0.5 * s_xs[lane], &s, &c);
            s_cs[lane] = c;
            s_sn[lane] = s;
          }
        }
        // first layer's data while the angles' sin/cos settle
        QuadLayerData<T, N> cur;
        cur.load(s_ry, s_tlo, s_thi, s_cz, round * layers_per_round, -1, tid);
        wave_sync();
        // per-sample RZ diagonal of this thread's amplitudes
        C dx[R];
        {
          T fr = 1, fi = 0;
#pragma unroll
          for (int q = 0; q < TB; ++q) {  // lane bits 0..5 (and wave bits 6, 7)
            const T c = (T)s_cs[N - 1 - q];
            const T si = ((kbase >> q) & 1u) ? (T)s_sn[N - 1 - q] : -(T)s_sn[N - 1 - q];
            const T nr = fr * c - fi * si;
            fi = fr * si + fi * c;
            fr = nr;
          }
          dx[0] = C{fr, fi};
#pragma unroll
          for (int j = 0; j < N - TB; ++j) {
            const T c = (T)s_cs[N - 1 - (TB + j)], s = (T)s_sn[N - 1 - (TB + j)];
#pragma unroll
            for (int r = 0; r < (1 << j); ++r) {
              const C dd = dx[r];
              dx[r | (1 << j)] = C{dd.x * c - dd.y * s, dd.x * s + dd.y * c};
              dx[r] = C{dd.x * c + dd.y * s, dd.y * c - dd.x * s};
            }
          }
        }
        // The round's first layer acts on |0..0>: its diagonal (data angles included) is a global phase and its RYs make a
        // real product state, amplitude k = prod_q (bit q of k ? sin : cos)(theta_q / 2).  Generated per thread -- no
        // cross-lane move, no exchange, no barrier -- instead of simulated: one layer less per round.
        C a[R];
        {
          T fe = 1, fo = 1;
#pragma unroll
          for (int q = 0; q < TB; ++q) {
            const C cs = cur.ry[N - 1 - q];
            const T f = ((kbase >> q) & 1u) ? cs.y : cs.x;
            if (q & 1) fo *= f;
            else fe *= f;
          }
          const T f = fe * fo;
#pragma unroll
          for (int r = 0; r < R; ++r) {
            T g = f;
#pragma unroll
            for (int j = 0; j < N - TB; ++j) {
              const C cs = cur.ry[N - 1 - (TB + j)];
              g *= ((r >> j) & 1) ? cs.y : cs.x;
            }
            a[r] = C{g, (T)0};
          }
          const int l_next = round * layers_per_round + 1;
          cur.load(s_ry, s_tlo, s_thi, s_cz, l_next < n_layers_all ? l_next : 0, 0, tid);
        }

        for (int li = 1; li < layers_per_round; ++li) {
          const int s = li % p.sel_layers;
          // ---- everything diagonal in front of this layer's RYs: one complex multiply per amplitude ----
#pragma unroll
          for (int r = 0; r < R; ++r) {
            C ph = cur.tlo;
            if constexpr (R > 1) ph = cmul2<T>(cur.thi[r], ph, times_i<T>(ph));
            if (s == 0) {  // block start: data re-upload.  A real (scalar) branch: if-converted, the multiply ran in
                           // every layer and a select threw it away in all but the block starts
              asm volatile("" ::: "memory");
              ph = cmul2<T>(dx[r], ph, times_i<T>(ph));
            }
            C v = cmul2<T>(ph, a[r], times_i<T>(a[r]));
            if constexpr (R > 1) {
              const uint32_t sb = ((cur.cz >> r) & 1u) << 31;  // CZ ring of the previous layer
              v = C{flip_sign(v.x, sb), flip_sign(v.y, sb)};
            }
            a[r] = v;
          }
          // ---- RY on every wire: wire w <-> index bit N-1-w.  Register bits, lane bits, then the wave pair ----
          if constexpr (N > TB) ry_regs<1, T, R>(a, cur.ry[N - 1 - TB].x, cur.ry[N - 1 - TB].y);
          if constexpr (N > TB + 1)
            ry_regs<2, T, R>(a, cur.ry[N > TB + 1 ? N - 2 - TB : 0].x, cur.ry[N > TB + 1 ? N - 2 - TB : 0].y);
          if constexpr (N > 5) ry_lane_xy<5, T, R>(a, cur.ry[N > 5 ? N - 1 - 5 : 0].x, cur.ry[N > 5 ? N - 1 - 5 : 0].y);
          if constexpr (N > 4) ry_lane_xy<4, T, R>(a, cur.ry[N > 4 ? N - 1 - 4 : 0].x, cur.ry[N > 4 ? N - 1 - 4 : 0].y);
          if constexpr (N > 3) ry_lane<3, T, R>(a, cur.ry[N > 3 ? N - 1 - 3 : 0].x, cur.ry[N > 3 ? N - 1 - 3 : 0].y * pm[3], lane);
          if constexpr (N > 2) ry_lane<2, T, R>(a, cur.ry[N > 2 ? N - 1 - 2 : 0].x, cur.ry[N > 2 ? N - 1 - 2 : 0].y * pm[2], lane);
          if constexpr (N > 1) ry_lane<1, T, R>(a, cur.ry[N > 1 ? N - 1 - 1 : 0].x, cur.ry[N > 1 ? N - 1 - 1 : 0].y * pm[1], lane);
          if constexpr (N > 0) ry_lane<0, T, R>(a, cur.ry[N > 0 ? N - 1 - 0 : 0].x, cur.ry[N > 0 ? N - 1 - 0 : 0].y * pm[0], lane);
          if constexpr (TB == 8) {
            // ---- bits 6 and 7 together: new = sum_j M[wv][wv ^ j] * amp(wave wv ^ j),  M = RY_7 (x) RY_6 (real) ----
            const T c6 = cur.ry[N - 1 - 6].x, t6 = cur.ry[N - 1 - 6].y * pm[6];
            const T c7 = cur.ry[N - 1 - 7].x, t7 = cur.ry[N - 1 - 7].y * pm[7];
            const T k0 = c7 * c6, k1 = c7 * t6, k2 = t7 * c6, k3 = t7 * t6;
            C* buf = s_slab + (size_t)xbuf_parity * (4 * R * kWave);
            xbuf_parity ^= 1;
#pragma unroll
            for (int r = 0; r < R; ++r) buf[(wv * R + r) * kWave + lane] = a[r];
            C own[R];
#pragma unroll
            for (int r = 0; r < R; ++r) own[r] = bcast<T>(k0) * a[r];   // this wave's term: before the barrier
            // the next layer's phase -- the first thing it needs -- is read in front of the barrier (one read, back long
            // before the slab write is), the rest of its tables behind the partners' amplitudes
            const int l_next = round * layers_per_round + li + 1;
            const int l_load = l_next < n_layers_all ? l_next : 0;
            const C tlo_next = QuadLayerData<T, N>::load_tlo(s_tlo, l_load, tid);
            __syncthreads();
            C p1[R], p2[R], p3[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
              p1[r] = buf[((wv ^ 1) * R + r) * kWave + lane];
              p2[r] = buf[((wv ^ 2) * R + r) * kWave + lane];
              p3[r] = buf[((wv ^ 3) * R + r) * kWave + lane];
            }
            cur.load_rest(s_ry, s_thi, s_cz, l_load, s % (N - 1), tid);
            cur.tlo = tlo_next;
#pragma unroll
            for (int r = 0; r < R; ++r) {   // two chains of depth two behind the reads instead of one of depth three
              const C o = __builtin_elementwise_fma(bcast<T>(k1), p1[r], own[r]);
              const C t = __builtin_elementwise_fma(bcast<T>(k3), p3[r], bcast<T>(k2) * p2[r]);
              a[r] = o + t;
            }
          } else {
            // the whole state is in this wave: nothing to exchange, only the next layer's data to fetch
            const int l_next = round * layers_per_round + li + 1;
            cur.load(s_ry, s_tlo, s_thi, s_cz, l_next < n_layers_all ? l_next : 0, s % (N - 1), tid);
          }
        }
        // (the ring and the RZ(omega) after the last RY layer are diagonal: they do not reach |amplitude|^2)
        // ---- <Z_w> -----------------------------------------------------------------------------------------------
        T ez[16];
#pragma unroll
        for (int w = 0; w < 16; ++w) ez[w] = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const T pr = a[r].x * a[r].x + a[r].y * a[r].y;
          const uint32_t k = ((uint32_t)r << TB) | kbase;
#pragma unroll
          for (int w = 0; w < N; ++w) ez[w] += ((k >> (N - 1 - w)) & 1u) ? -pr : pr;
        }
        wave_sync();   // (this wave's readers of s_cs / s_sn are done)
        // (summed over the wavefront in the engine's own precision -- the 64 terms are |amplitude|^2 of that precision --
        //  and over the four waves in double)
        T* s_pz = reinterpret_cast<T*>(s_part_z);
        {
          T v8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v8[j] = ez[j];
          wave_reduce8_into<T, true>(v8, lane, llane, s_pz + wv * 16);
          if constexpr (N > 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v8[j] = ez[8 + j];
            wave_reduce8_into<T, true>(v8, lane, llane, s_pz + wv * 16 + 8);
          }
        }
        __syncthreads();
        if (lane < N) {
          // TB == 6: every wave summed the whole (replicated) state -- take wave 0's
          // (below 6 qubits the wave's 64 lanes summed 2^(6-TB) copies)
          const double e = TB == 8 ? (double)s_pz[lane] + (double)s_pz[16 + lane] + (double)s_pz[32 + lane] +
                                         (double)s_pz[48 + lane]
                                   : (double)s_pz[lane] * (1.0 / (double)(1 << (TB < 6 ? 6 - TB : 0)));
          s_xs[lane] = e * p.enc_scale;  // next round's angles
          s_cs[lane] = e;                // plain <Z_w> for linear_up (s_cs is rebuilt next round)
        }
      }
      wave_sync();
      if (stamp && step == 0) d.stamps[3] = __builtin_amdgcn_s_memtime();
      double ev[N];
#pragma unroll
      for (int j = 0; j < N; ++j) ev[j] = s_cs[j];

      // ---- linear_up (+ sampling update); the result is the next step's image -------------------------------
#pragma unroll
      for (int i = 0; i < PPT; ++i) {
        const int pix = tid + i * 256;
        double o = bur[i];
#pragma unroll
        for (int j = 0; j < N; ++j) o = fma(ev[j], wur[i][j], o);
        if (d.post_mode == 1) o = fmin(fmax(xr[i] - (o - 0.5) * 0.1 * d.noise_factor, 0.0), 1.0);
        if (pix < Q) {
          y[(size_t)step * d.y_step_stride + sample * d.y_ld + pix] = o;
          xr[i] = o;
        }
      }
      if (stamp && step == 0) d.stamps[4] = __builtin_amdgcn_s_memtime();
    }
  }
}

}  // namespace qiddm
