// qsim_quad.h -- low-latency dense-net sampler for small batches: FOUR wavefronts per sample.
//
// At BASELINE config 2 (batch 256, n = 8) the one-wave-per-sample kernels leave three of every four
// SIMDs idle and the step is pure latency (phase stamps).  Here one 256-thread
// workgroup owns a sample: amplitude index k = (r << 8) | (wave << 6) | lane, i.e. index bits 6 and 7
// live in the wave number.  Gates on lane / register bits reuse the Engine primitives (packed FMA, DPP,
// permlane swaps); the two wave-bit gates of a layer are applied TOGETHER as one 4x4 exchange through
// a double-buffered LDS slab (one s_barrier per layer).  linear_down / linear_up are split over all 256
// threads, and the whole sampling loop x <- net(x) (reference src/models.py:124-136) can run for
// `n_steps` iterations inside the launch with x held in registers.
//
// Restricted to what the dense nets need: RZ data re-uploading, CZ rings, <Z> read-out, 8 <= n <= 10.
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

template <typename T, int N>
struct QuadSmem {
  static constexpr int R = 1 << (N - 8);
  __host__ __device__ static size_t gate_bytes(int64_t n_rot) { return (size_t)n_rot * kLdsGateReals * sizeof(T); }
  static constexpr size_t kCzBytes = (size_t)(N - 1) * 256 * 4;
  static constexpr size_t kSlabBytes = (size_t)2 * 4 * R * kWave * 2 * sizeof(T);
  static constexpr size_t kMiscBytes = (4 * 16 + 16 + 16 + 16) * sizeof(double);
  __host__ __device__ static size_t bytes(int64_t n_rot) {
    return gate_bytes(n_rot) + kCzBytes + kSlabBytes + kMiscBytes;
  }
};

// gate on lane bit Q (index bit Q, wire N-1-Q) in the quad layout
template <typename T, int N, int Q>
__device__ __forceinline__ void lane_bit_gate(const Engine<T, N - 2>& eng, V2<T> (&a)[1 << (N - 8)],
                                              const T* s_gates, int gate0, int llane) {
  using C = V2<T>;
  constexpr int R = 1 << (N - 8);
  const C* gp = reinterpret_cast<const C*>(s_gates + (size_t)(gate0 + (N - 1 - Q)) * kLdsGateReals);
  if constexpr (Q >= 4 && R >= 2) {
    C m[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = gp[i];
    eng.template swap_reg0_with_lane_bit<Q>(a);
    eng.template gate_regs<1>(a, m, m + 4);
    eng.template swap_reg0_with_lane_bit<Q>(a);
  } else {
    const C* hp = gp + (((llane >> Q) & 1) << 2);
    C h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = hp[i];
    eng.template gate_lane<Q>(a, h);
  }
}

// all gate data one thread needs for one SEL layer, loaded together so that a single LDS round trip
// (issued a layer ahead) covers the whole layer
template <typename T, int N>
struct QuadLayerGates {
  using C = V2<T>;
  static constexpr int NREG = N - 8;
  C lane[6][4];                    // lane bits 0..5 without register partner: this lane's half
  C full[2][8];                    // lane bits 4, 5 when exchanged with register bit 0 (R >= 2)
  C reg[NREG > 0 ? NREG : 1][8];   // register bits 8..N-1
  C w6[4], w7[4];                  // wave bits 6, 7: this wave's halves

  __device__ __forceinline__ void load(const T* s_gates, int gate0, int llane, int wv) {
    constexpr int R = 1 << (N - 8);
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const C* gp = reinterpret_cast<const C*>(s_gates + (size_t)(gate0 + (N - 1 - q)) * kLdsGateReals);
      if (q >= 4 && R >= 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) full[q - 4][i] = gp[i];
      } else {
        const C* hp = gp + (((llane >> q) & 1) << 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) lane[q][i] = hp[i];
      }
    }
#pragma unroll
    for (int j = 0; j < NREG; ++j) {
      const C* gp = reinterpret_cast<const C*>(s_gates + (size_t)(gate0 + (N - 1 - (8 + j))) * kLdsGateReals);
#pragma unroll
      for (int i = 0; i < 8; ++i) reg[j][i] = gp[i];
    }
    const C* g6 = reinterpret_cast<const C*>(s_gates + (size_t)(gate0 + (N - 1 - 6)) * kLdsGateReals) + ((wv & 1) << 2);
    const C* g7 = reinterpret_cast<const C*>(s_gates + (size_t)(gate0 + (N - 1 - 7)) * kLdsGateReals) +
                  (((wv >> 1) & 1) << 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      w6[i] = g6[i];
      w7[i] = g7[i];
    }
  }
};

template <typename T, int N, int PPT>
__global__ __launch_bounds__(256) void dense_quad_kernel(
    const double* __restrict__ x, const double* __restrict__ wd, const double* __restrict__ bd,
    const double* __restrict__ angles, const double* __restrict__ wu, const double* __restrict__ bu,
    double* __restrict__ y, const QuadScalars d, const KScalars p) {
  static_assert(N >= 8 && N <= 10, "quad layout: 8..10 qubits");
  using E = Engine<T, N - 2>;  // lane bits 0..5 + register bits; its register bit j is index bit 8 + j
  using C = V2<T>;
  constexpr int R = 1 << (N - 8);
  // PPT pixels per thread (in/out features <= 256 * PPT) stay in registers across the steps
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int n_rot = p.n_rounds * p.n_blocks * p.sel_layers * N;
  T* s_gates = reinterpret_cast<T*>(smem_raw);
  uint32_t* s_cz = reinterpret_cast<uint32_t*>(smem_raw + QuadSmem<T, N>::gate_bytes(n_rot));
  C* s_slab = reinterpret_cast<C*>(reinterpret_cast<unsigned char*>(s_cz) + QuadSmem<T, N>::kCzBytes);
  double* s_part = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(s_slab) + QuadSmem<T, N>::kSlabBytes);
  double* s_xs = s_part + 4 * 16;  // [16] angles of the round
  double* s_cs = s_xs + 16;        // [16] cos(x/2)   (after the read-out: plain <Z_w>)
  double* s_sn = s_cs + 16;        // [16] sin(x/2)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int llane = logical_lane(lane);
  const bool stamp = d.stamps != nullptr && blockIdx.x == 0 && tid == 0;
  if (stamp) d.stamps[0] = __builtin_amdgcn_s_memtime();
  E eng;
  eng.s_gates = s_gates;
  eng.lane = lane;
  eng.llane = llane;
  eng.sub = llane;
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

  // ---- staging: gate images from the raw angles, CZ sign bits per (range, thread) -----------------
  for (int g = tid; g < n_rot; g += 256) {
    const double phi = angles[g * 3 + 0], theta = angles[g * 3 + 1], omega = angles[g * 3 + 2];
    double c, s, ca, sa, cb, sb;
    table_sincos<T>(0.5 * theta, &s, &c);
    table_sincos<T>(0.5 * (phi + omega), &sa, &ca);
    table_sincos<T>(0.5 * (phi - omega), &sb, &cb);
    E::put_gate(s_gates + (size_t)g * kLdsGateReals, (T)(ca * c), (T)(-sa * c), (T)(-cb * s), (T)(-sb * s),
                (T)(cb * s), (T)(-sb * s), (T)(ca * c), (T)(sa * c));
  }
  const uint32_t kbase = ((uint32_t)wv << 6) | (uint32_t)llane;  // index bits 0..7 of this thread
  for (int rr = 1; rr < N; ++rr) {
    uint32_t bits = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) bits |= cz_ring_parity<N>(((uint32_t)r << 8) | kbase, rr) << r;
    s_cz[(rr - 1) * 256 + tid] = bits;
  }
  const double bd_mine = (bd && tid < N) ? bd[tid] : 0.0;
  __syncthreads();
  if (stamp) d.stamps[1] = __builtin_amdgcn_s_memtime();

  const int layers_per_round = p.n_blocks * p.sel_layers;
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
      __syncthreads();  // s_part free (previous readers done)
      if (lane < 16) s_part[wv * 16 + lane] = 0.0;
      {
        double v8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = acc[j];
        wave_reduce8_into<double>(v8, lane, llane, s_part + wv * 16);
        if constexpr (N > 8) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v8[j] = acc[8 + j];
          wave_reduce8_into<double>(v8, lane, llane, s_part + wv * 16 + 8);
        }
      }
      __syncthreads();
      if (tid < N) {
        // wave_reduce8_into leaves value idx in slot idx of its 8-slot group
        const double h = s_part[tid] + s_part[16 + tid] + s_part[32 + tid] + s_part[48 + tid] + bd_mine;
        s_xs[tid] = h * p.enc_scale;
      }
      if (stamp && step == 0) d.stamps[2] = __builtin_amdgcn_s_memtime();

      // ---- circuit rounds ------------------------------------------------------------------------------
      for (int round = 0; round < p.n_rounds; ++round) {
        __syncthreads();  // s_xs ready
        if (tid < N) {
          double s, c;
          table_sincos<T>(0.5 * s_xs[tid], &s, &c);
          s_cs[tid] = c;
          s_sn[tid] = s;
        }
        // first layer's gates while the angles' sin/cos settle
        QuadLayerGates<T, N> cur;
        cur.load(s_gates, round * layers_per_round * N, llane, wv);
        __syncthreads();
        // per-sample RZ diagonal of this thread's amplitudes
        C dx[R];
        {
          T fr = 1, fi = 0;
#pragma unroll
          for (int q = 0; q < 8; ++q) {  // lane bits 0..5 and wave bits 6, 7
            const T c = (T)s_cs[N - 1 - q];
            const T si = ((kbase >> q) & 1u) ? (T)s_sn[N - 1 - q] : -(T)s_sn[N - 1 - q];
            const T nr = fr * c - fi * si;
            fi = fr * si + fi * c;
            fr = nr;
          }
          dx[0] = C{fr, fi};
#pragma unroll
          for (int j = 0; j < N - 8; ++j) {
            const T c = (T)s_cs[N - 1 - (8 + j)], s = (T)s_sn[N - 1 - (8 + j)];
#pragma unroll
            for (int r = 0; r < (1 << j); ++r) {
              const C dd = dx[r];
              dx[r | (1 << j)] = C{dd.x * c - dd.y * s, dd.x * s + dd.y * c};
              dx[r] = C{dd.x * c + dd.y * s, dd.y * c - dd.x * s};
            }
          }
        }
        C a[R];
#pragma unroll
        for (int r = 0; r < R; ++r) a[r] = C{(T)0, (T)0};
        a[0] = C{kbase == 0 ? (T)1 : (T)0, (T)0};

        for (int li = 0; li < layers_per_round; ++li) {
          const int s = li % p.sel_layers;
          if (s == 0) {  // block start: data re-upload
#pragma unroll
            for (int r = 0; r < R; ++r) a[r] = cmul2<T>(dx[r], a[r], times_i<T>(a[r]));
          }
          // wire w <-> index bit N-1-w.  Register bits, lane bits, then the wave pair.
          if constexpr (N > 8) eng.template gate_regs<1>(a, cur.reg[0], cur.reg[0] + 4);
          if constexpr (N > 9) eng.template gate_regs<2>(a, cur.reg[N > 9 ? 1 : 0], cur.reg[N > 9 ? 1 : 0] + 4);
          if constexpr (R >= 2) {
            eng.template swap_reg0_with_lane_bit<5>(a);
            eng.template gate_regs<1>(a, cur.full[1], cur.full[1] + 4);
            eng.template swap_reg0_with_lane_bit<5>(a);
            eng.template swap_reg0_with_lane_bit<4>(a);
            eng.template gate_regs<1>(a, cur.full[0], cur.full[0] + 4);
            eng.template swap_reg0_with_lane_bit<4>(a);
          } else {
            eng.template gate_lane<5>(a, cur.lane[5]);
            eng.template gate_lane<4>(a, cur.lane[4]);
          }
          eng.template gate_lane<3>(a, cur.lane[3]);
          eng.template gate_lane<2>(a, cur.lane[2]);
          eng.template gate_lane<1>(a, cur.lane[1]);
          eng.template gate_lane<0>(a, cur.lane[0]);
          // ---- bits 6 and 7 together: new = sum_j M[wv][wv ^ j] * amp(wave wv ^ j) -------------------
          const C a6 = cur.w6[0], b6 = cur.w6[2], a7 = cur.w7[0], b7 = cur.w7[2];  // own / partner coefficients
          const C c0 = cmul2<T>(a7, a6, times_i<T>(a6));
          const C c1 = cmul2<T>(a7, b6, times_i<T>(b6));
          const C c2 = cmul2<T>(b7, a6, times_i<T>(a6));
          const C c3 = cmul2<T>(b7, b6, times_i<T>(b6));
          C* buf = s_slab + (size_t)xbuf_parity * (4 * R * kWave);
          xbuf_parity ^= 1;
#pragma unroll
          for (int r = 0; r < R; ++r) buf[(wv * R + r) * kWave + lane] = a[r];
          // next layer's gates: in flight across the barrier
          {
            const int g_next = (round * layers_per_round + li + 1) * N;
            cur.load(s_gates, g_next < n_rot ? g_next : 0, llane, wv);
          }
          const uint32_t czbits = s_cz[(s % (N - 1)) * 256 + tid];
          __syncthreads();
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const C p1 = buf[((wv ^ 1) * R + r) * kWave + lane];
            const C p2 = buf[((wv ^ 2) * R + r) * kWave + lane];
            const C p3 = buf[((wv ^ 3) * R + r) * kWave + lane];
            C o = cmul2<T>(a[r], c0, times_i<T>(c0));
            o = cfma<T>(p1, c1, times_i<T>(c1), o);
            o = cfma<T>(p2, c2, times_i<T>(c2), o);
            o = cfma<T>(p3, c3, times_i<T>(c3), o);
            const uint32_t sb = ((czbits >> r) & 1u) << 31;  // CZ ring
            a[r] = C{flip_sign(o.x, sb), flip_sign(o.y, sb)};
          }
        }
        // ---- <Z_w> -----------------------------------------------------------------------------------------------
        T ez[16];
#pragma unroll
        for (int w = 0; w < 16; ++w) ez[w] = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const T pr = a[r].x * a[r].x + a[r].y * a[r].y;
          const uint32_t k = ((uint32_t)r << 8) | kbase;
#pragma unroll
          for (int w = 0; w < N; ++w) ez[w] += ((k >> (N - 1 - w)) & 1u) ? -pr : pr;
        }
        __syncthreads();  // s_part free
        if (lane < 16) s_part[wv * 16 + lane] = 0.0;
        {
          double v8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v8[j] = (double)ez[j];
          wave_reduce8_into<double>(v8, lane, llane, s_part + wv * 16);
          if constexpr (N > 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v8[j] = (double)ez[8 + j];
            wave_reduce8_into<double>(v8, lane, llane, s_part + wv * 16 + 8);
          }
        }
        __syncthreads();
        if (tid < N) {
          const double e = s_part[tid] + s_part[16 + tid] + s_part[32 + tid] + s_part[48 + tid];
          s_xs[tid] = e * p.enc_scale;  // next round's angles
          s_cs[tid] = e;                // plain <Z_w> for linear_up (s_cs is rebuilt next round)
        }
      }
      __syncthreads();
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
