// qsim_lean.h -- the sampling loop of the 8- and 6-qubit dense nets with the layer's dependent chain cut to the bone.
//
// Same decomposition as qsim_quad.h (8 qubits: four wavefronts per sample, amplitude k = (wave << 6) | lane, one LDS
// exchange per layer for the two wave-bit gates; 6 qubits: the state fits one wavefront, every wave runs the circuit on
// its own copy and there is no exchange and no barrier in a round), but built around what the microbenchmarks of
// tools/ubench/ say a
// LONE wavefront on a SIMD pays: ~11 cycles per DEPENDENT vector instruction whatever it is, ~155 cycles for the LDS
// round trip + barrier.  A layer is a dependent chain (every gate acts on the same amplitudes), so its time is
// (chain depth) x 11 + 155 and the only lever is depth:
//
//   * RY in tangent form.  RY = c [[1, -t], [t, 1]], t = tan(theta / 2): own' = own +- t * partner is ONE instruction
//     whose partner operand is fetched by DPP inside it (`v_fmac_f32_dpp`, accumulator = own value, in place) --
//     depth 1 per lane-bit gate instead of 3 (dpp mov, wait state, packed fma behind a packed multiply).  The factors c
//     of a layer commute with everything and are folded into that layer's phase table, so they cost nothing.
//   * the row-crossing bits 4 / 5 keep the permlane swap of (re, im) (qsim_quad.h), with the 2 x 2 in tangent form:
//     swap, one fma per member, swap -- depth 3 instead of 4; at 8 qubits the two gates share their swaps and never swap
//     back (the LDS exchange that follows stores into natural slots): depth 4 for both.
//   * the wave-bit 4 x 4 in tangent form: own + k1 p1 + k2 p2 + k3 p3 with per-wave k = +-t6, +-t7, their product;
//     nothing to multiply before the barrier.
//   * per-layer data (this thread's phase, eight tangents) is read one layer ahead while the exchange is in flight,
//     into one of two register sets that alternate (no copies).
//   * linear_down is NOT evaluated per step: with goal = "data" the loop is x <- net(x) with no clamp (reference
//     src/models.py:127-129), so the next step's angles are (W_down W_up) z + (W_down b_up + b_down) with z = <Z> of
//     this step -- an n x n product built once per weights (lean_tables_kernel).  Where the angles cannot reach the
//     state at all (one block per round: RZ on |0..0> is a global phase, finding F2) they are not computed.
//
// Tried on top and dropped (round 3): the tail of a step (|amplitude|^2, signed sums, linear_up, stores: ~1 250 of ~6 900
// cycles, and independent of the next step's circuit when there is no re-upload) overlapped with the next step's layers
// (a) as pieces placed in the two exchange windows of the first five layers -- behind in-order issue a wave that waits for
// its side chain waits with its main chain too: the layers grew by what the tail had cost (6 924 -> 6 492 cycles per step);
// (b) on a fifth, helper wavefront sharing SIMD 0 with wave 0, applying the last layer's 4 x 4 itself and owning all 256
// probabilities -- correct, but every barrier now waits for five waves and wave 0 shares its issue: 3.35 -> 3.51 us per
// step in float32 (4.28 -> 4.11 in complex128).
//
// The tangent form needs cos(theta / 2) away from zero: the table builder records max |t|; the host routes weights with
// max |t| > kLeanMaxTan to dense_quad_kernel (qiddm_dense_sample_lean_check).  Shipped checkpoints have |theta / 2| < 0.9.
#pragma once
#include "qsim_quad.h"

namespace qiddm {

constexpr double kLeanMaxTan = 16.0;
constexpr int kLeanHeaderDoubles = 80;   // [0] max |t|, [1 .. n n] M = W_down W_up (row-major [j][i]), then v [n], pad

// layout of the lean tables behind the header, in elements of T (built once per weights, copied to LDS per launch)
template <typename T, int N>
struct LeanTables {
  static_assert(N == 6 || N == 8, "lean sampling loop: 6 or 8 qubits");
  static constexpr int TL = N == 8 ? 256 : 64;   // amplitudes = phase-table entries per layer (thread / lane order)
  // per layer: ph[TL] complex, un[8] = tangent of index bits 0..N-1
  __host__ __device__ static constexpr size_t ph_elems(int layers) { return (size_t)layers * TL * 2; }
  __host__ __device__ static constexpr size_t un_elems(int layers) { return (size_t)layers * 8; }
  __host__ __device__ static constexpr size_t a0_elems(int rounds) { return (size_t)rounds * TL; }
  __host__ __device__ static size_t elems(int layers, int rounds) {
    return ph_elems(layers) + un_elems(layers) + a0_elems(rounds);
  }
  __host__ __device__ static size_t bytes(int layers, int rounds) {
    return (size_t)kLeanHeaderDoubles * sizeof(double) + elems(layers, rounds) * sizeof(T);
  }
  // LDS of dense_lean_kernel: the tables, the double-buffered exchange slab, partial sums and per-wave angle copies
  __host__ __device__ static size_t lds_bytes(int layers, int rounds) {
    return (elems(layers, rounds) * sizeof(T) + 15) / 16 * 16 + (size_t)2 * 4 * kWave * 2 * sizeof(T) +
           (size_t)(3 * 4 * 16 + 3 * 4 * 16 + 80) * sizeof(double);
  }
};

// ---- table builder: one 256-thread workgroup --------------------------------------------------------------------------
template <typename T, int N>
__global__ __launch_bounds__(256) void lean_tables_kernel(const double* __restrict__ angles,
                                                          const double* __restrict__ wd, const double* __restrict__ bd,
                                                          const double* __restrict__ wu, const double* __restrict__ bu,
                                                          int features, unsigned char* __restrict__ tables,
                                                          const KScalars p) {
  using C = V2<T>;
  using LT = LeanTables<T, N>;
  __shared__ double s_alpha[1024];   // phi^l_w + omega^{l-1}_w per (layer, wire): layers <= 128
  __shared__ double s_c[1024], s_s[1024];
  __shared__ double s_max[256];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lpr = p.n_blocks * p.sel_layers, layers = p.n_rounds * lpr, n_rot = layers * N;
  double* head = reinterpret_cast<double*>(tables);
  T* body = reinterpret_cast<T*>(tables + kLeanHeaderDoubles * sizeof(double));
  C* ph = reinterpret_cast<C*>(body);
  T* un = body + LT::ph_elems(layers);
  T* a0 = un + LT::un_elems(layers);
  double tmax = 0.0;
  for (int g = tid; g < n_rot; g += 256) {
    double c, sn;
    sincos(0.5 * angles[g * 3 + 1], &sn, &c);
    s_c[g] = c;
    s_s[g] = sn;
    const int li = (g / N) % lpr, w = g % N;
    s_alpha[g] = angles[g * 3 + 0] + (li > 0 ? angles[(g - N) * 3 + 2] : 0.0);
    T t = (T)0;
    if (li > 0) {                                   // a round's first layer is generated from (c, s), never divided
      tmax = fmax(tmax, c != 0.0 ? fabs(sn / c) : 1e300);
      t = (T)(c != 0.0 ? sn / c : 0.0);
    }
    un[(g / N) * 8 + (N - 1 - w)] = t;              // tangent of index bit q = N-1-w
  }
  if (N < 8)
    for (int i = tid; i < layers * (8 - N); i += 256) un[(i / (8 - N)) * 8 + N + i % (8 - N)] = (T)0;
  s_max[tid] = tmax;
  __syncthreads();
  if (tid == 0) {
    double m = 0.0;
    for (int i = 0; i < 256; ++i) m = fmax(m, s_max[i]);
    head[0] = m;
  }
  // this thread's amplitude index (8 qubits: one per thread; 6 qubits: the first wavefront writes, one per lane)
  const uint32_t k = (N == 8 ? ((uint32_t)wv << 6) : 0u) | (uint32_t)logical_lane(lane);
  const int slot = N == 8 ? tid : lane;
  if (N == 8 || wv == 0) {
    for (int l = 0; l < layers; ++l) {
      const int li = l % lpr;
      if (li == 0) {
        // the round's first layer acts on |0..0>: a real product state (its diagonal is a global phase)
        double f = 1.0;
#pragma unroll
        for (int q = 0; q < N; ++q) f *= ((k >> q) & 1u) ? s_s[l * N + (N - 1 - q)] : s_c[l * N + (N - 1 - q)];
        a0[(l / lpr) * LT::TL + slot] = (T)f;
        ph[l * LT::TL + slot] = C{(T)1, (T)0};
        continue;
      }
      double ang = 0.0, scale = 1.0;
#pragma unroll
      for (int q = 0; q < N; ++q) {
        const double al = s_alpha[l * N + (N - 1 - q)];
        ang += ((k >> q) & 1u) ? 0.5 * al : -0.5 * al;
        scale *= s_c[l * N + q];                       // every factor cos(theta_w / 2) of THIS layer's RYs
      }
      double c, sn;
      sincos(ang, &sn, &c);
      if (cz_ring_parity<N>(k, ((li - 1) % p.sel_layers) % (N - 1) + 1)) scale = -scale;   // CZ ring of the layer before
      ph[l * LT::TL + slot] = C{(T)(c * scale), (T)(sn * scale)};
    }
  }
  // M = W_down W_up and v = W_down b_up + b_down (float64): the angles of the NEXT step from this step's <Z>
  if (wd != nullptr && wu != nullptr && tid < N * N + N) {
    double acc = 0.0;
    if (tid < N * N) {
      const int j = tid / N, i = tid % N;
      for (int px = 0; px < features; ++px) acc = fma(wd[(size_t)j * features + px], wu[(size_t)px * N + i], acc);
      head[1 + tid] = acc;
    } else {
      const int j = tid - N * N;
      for (int px = 0; px < features; ++px) acc = fma(wd[(size_t)j * features + px], bu ? bu[px] : 0.0, acc);
      head[1 + N * N + j] = acc + (bd ? bd[j] : 0.0);
    }
  }
}

// ---- gate helpers (tangent form) ------------------------------------------------------------------------------------------
// The four lane-bit gates with a DPP partner (index bits 0..3: quad_perm, quad_perm, row_half_mirror, row_ror:8):
// own += t_signed * partner on both components, `v_fmac_f32_dpp` with accumulator == own value, in place.  Written out
// because the fused-operand form is not reachable from the compiler, and because the hazard recogniser does not look
// inside inline asm: a DPP read of a register needs two wait states behind the vector instruction that wrote it.  The
// imaginary-part instruction of a gate is one of them, `s_nop 0` the other (`s_nop 1` in front: the phase multiply wrote
// the pair; `s_nop 1` behind: a permlane swap follows).  They sit in the shadow of the ~11-cycle dependent issue.
__device__ __forceinline__ void ry_t_dpp4(V2<float>& a, float t0, float t1, float t2, float t3) {
  float x = a.x, y = a.y;
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f32_dpp %0, %0, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %1, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 0\n\t"
      "v_fmac_f32_dpp %0, %0, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %1, %1, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 0\n\t"
      "v_fmac_f32_dpp %0, %0, %4 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %1, %1, %4 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 0\n\t"
      "v_fmac_f32_dpp %0, %0, %5 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %1, %1, %5 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1"
      : "+v"(x), "+v"(y)
      : "v"(t0), "v"(t1), "v"(t2), "v"(t3));
  a = V2<float>{x, y};
}
template <int CTRL>
__device__ __forceinline__ void ry_t_dpp(V2<double>& a, double ts) {
  const double px = __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(a.x), CTRL, 0xF, 0xF, true),
                                     __builtin_amdgcn_mov_dpp(__double2loint(a.x), CTRL, 0xF, 0xF, true));
  const double py = __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(a.y), CTRL, 0xF, 0xF, true),
                                     __builtin_amdgcn_mov_dpp(__double2loint(a.y), CTRL, 0xF, 0xF, true));
  a = V2<double>{fma(ts, px, a.x), fma(ts, py, a.y)};
}
__device__ __forceinline__ void ry_t_dpp4(V2<double>& a, double t0, double t1, double t2, double t3) {
  ry_t_dpp<0xB1>(a, t0);
  ry_t_dpp<0x4E>(a, t1);
  ry_t_dpp<0x141>(a, t2);
  ry_t_dpp<0x128>(a, t3);
}
// row-crossing lane bit (4, 5): permlane swap of (re, im), the 2 x 2 on (low member, high member), swap back
template <int Q, typename T>
__device__ __forceinline__ void ry_t_swap(V2<T>& a, T t) {
  T lo = a.x, hi = a.y;
  swap_parts<Q>(lo, hi);
  T nlo = fma(-t, hi, lo), nhi = fma(t, lo, hi);
  swap_parts<Q>(nlo, nhi);
  a = V2<T>{nlo, nhi};
}

// what a thread holds for one layer
template <typename T>
struct LeanLayer {
  V2<T> ph;
  T ts[4];       // index bits 0..3, signed by this lane's bit
  T t4, t5;
  T k1, k2, k3;  // wave-bit exchange: partners wave^1, wave^2, wave^3
};

// REUP: the circuit re-uploads its data angles (n_blocks > 1); without it the angles never reach the state and no
//       angle code is compiled at all (no branch in the layer either: a taken branch costs a lone wavefront ~50 cycles).
// LPR:  layers per round as a compile-time constant (fully unrolled layer sequence; 14 = the flagship QNN_noise(784, 8,
//       14), 12 / 28 = the LL-style (6 / 14 blocks x 2) -- with REUP a compiled-in count also means TWO layers per
//       block), or 0 for a runtime count (loop over pairs of layers, block starts by comparison).
// POST: the "noise"-goal update of the sampling loop, x <- clamp(x - (net(x) - 0.5) * 0.1 * noise_factor, 0, 1)
//       (reference src/models.py:130-134).  The clamp breaks the composite map, so the image stays in registers
//       (PPT pixels per thread) and a re-uploading net runs its whole linear_down every step, weights in registers too.
template <typename T, int N, int PPT, bool REUP, int LPR, bool POST>
__global__ __launch_bounds__(256) void dense_lean_kernel(
    const double* __restrict__ x, const double* __restrict__ wd, const double* __restrict__ bd,
    const double* __restrict__ wu, const double* __restrict__ bu, double* __restrict__ y,
    const unsigned char* __restrict__ tables, const QuadScalars d, const KScalars p) {
  using C = V2<T>;
  using QT = LeanTables<T, N>;
  constexpr int TL = QT::TL;
  using V4 = T __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lpr = LPR > 0 ? LPR : p.n_blocks * p.sel_layers, layers = p.n_rounds * lpr;
  T* s_body = reinterpret_cast<T*>(smem_raw);
  const C* s_ph = reinterpret_cast<const C*>(s_body);
  const T* s_un = s_body + QT::ph_elems(layers);
  const T* s_a0 = s_un + QT::un_elems(layers);
  C* s_slab = reinterpret_cast<C*>(smem_raw + (QT::elems(layers, p.n_rounds) * sizeof(T) + 15) / 16 * 16);
  double* s_part = reinterpret_cast<double*>(s_slab + 2 * 4 * kWave);   // [4][16] partials of linear_down
  double* s_part_z = s_part + 4 * 16;                                    // [2][4][16] partials of the read-out, alternating
  double* s_part2 = s_part_z + 4 * 4 * 16;                               // second buffer of linear_down's partials
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int llane = logical_lane(lane);
  double* s_xs = s_part_z + 2 * 4 * 16 + wv * 16;   // [16] angles of the round, this wave's copy
  double* s_cs = s_part_z + 3 * 4 * 16 + wv * 16;   // [16] cos(x/2) then sin(x/2) of the round, in T
  const double* head = reinterpret_cast<const double*>(tables);
  const int P = d.in_features, Q = d.out_features;
  // diagnostics (tools/stamp_lean.py): s_memtime of workgroup 0 / thread 0 in the launch's second step
  const bool stamp = d.stamps != nullptr && blockIdx.x == 0 && tid == 0;
  if (stamp) d.stamps[0] = __builtin_amdgcn_s_memtime();

  // ---- per-launch setup: linear_up weights of this thread's pixels in registers, tables into LDS --------------------
  double wur[PPT][N], bur[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int pix = tid + i * 256;
#pragma unroll
    for (int j = 0; j < N; ++j) wur[i][j] = pix < Q ? wu[(size_t)pix * N + j] : 0.0;   // (zeros beyond the image)
    bur[i] = (bu && pix < Q) ? bu[pix] : 0.0;
  }
  // "noise" goal: the image, and for a re-uploading net linear_down's weights and bias, live in registers
  double xr[POST ? PPT : 1], wdr[POST && REUP ? PPT : 1][N], bdr = 0.0;
  if constexpr (POST && REUP) {
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int pix = tid + i * 256;
#pragma unroll
      for (int j = 0; j < N; ++j) wdr[i][j] = pix < P ? wd[(size_t)j * P + pix] : 0.0;
    }
    bdr = (bd && lane < N) ? bd[lane] : 0.0;
  }
  {
    // 16 bytes per thread and trip, four trips in flight (the element count is a multiple of four)
    const V4* src = reinterpret_cast<const V4*>(tables + kLeanHeaderDoubles * sizeof(double));
    V4* dst = reinterpret_cast<V4*>(s_body);
    const int n4 = (int)(QT::elems(layers, p.n_rounds) / 4);
    int i = tid;
    for (; i + 768 < n4; i += 1024) {
      const V4 v0 = src[i], v1 = src[i + 256], v2 = src[i + 512], v3 = src[i + 768];
      dst[i] = v0;
      dst[i + 256] = v1;
      dst[i + 512] = v2;
      dst[i + 768] = v3;
    }
    for (; i < n4; i += 256) dst[i] = src[i];
  }
  const uint32_t kbase = (N == 8 ? ((uint32_t)wv << 6) : 0u) | (uint32_t)llane;
  const int slot = N == 8 ? tid : lane;   // this thread's entry of a per-amplitude table
  // where this lane's two values go after the un-swapped bit-5 / bit-4 gates (8 qubits; elements of T inside the wave's
  // 64 (re, im) slots): lanes 0-15 hold re[l], re[l+16]; 16-31 re[l+16], re[l+32]; 32-47 im[l-32], im[l-16];
  // 48-63 im[l-16], im[l] -- first value at 2 * index + component, second 16 amplitudes (32 elements) further
  const int scatter_off = 2 * (lane < 16 ? lane : lane < 32 ? lane + 16 : lane < 48 ? lane - 32 : lane - 16) + (lane >> 5);
  T pm[8];   // +-1 by this thread's index bit
#pragma unroll
  for (int q = 0; q < 8; ++q) pm[q] = ((kbase >> q) & 1u) ? (T)1 : (T)-1;
  // composite map of the next step's angles (M row-major, then v) in LDS
  double* s_map = s_part_z + 5 * 4 * 16;   // [72]
  if (REUP && tid < N * N + N) s_map[tid] = head[1 + tid];
  auto wave_sync = [&]() {   // LDS hand-over inside the wavefront
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // a layer's data in two halves: the raw LDS reads (issued a layer ahead, at the top of the layer before), and what is
  // derived from them (signs by this thread's index bits, k3) -- multiplies that ride in the chain's empty issue slots
  struct Raw {
    C ph;
    V4 lo, hi;   // t of index bits 0..3 / 4..7
  };
  auto fetch_layer = [&](Raw& r, int l) {
    r.ph = s_ph[l * TL + slot];
    const V4* u = reinterpret_cast<const V4*>(s_un + l * 8);
    r.lo = u[0];
    r.hi = u[1];
  };
  // every global load of the setup has landed before the loops: the steps' stores then never wait for a load counter
  // (loads and stores share it, in order -- a wait for a setup load inside the loop is a wait for the previous store)
  __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
  __syncthreads();
  if (stamp) d.stamps[1] = __builtin_amdgcn_s_memtime();

  int xbuf_parity = 0, zbuf_parity = 0, dbuf_parity = 0;
  double ev[8];   // <Z_w> of the last round (N used): linear_up's input and the next step's composite input
#pragma unroll
  for (int j = 0; j < 8; ++j) ev[j] = 0.0;

  for (int64_t sample = blockIdx.x; sample < p.batch; sample += gridDim.x) {
    for (int step = 0; step < d.n_steps; ++step) {
      const bool st = stamp && step == 1;
      if (st) d.stamps[2] = __builtin_amdgcn_s_memtime();
      // ---- this step's data angles -------------------------------------------------------------------------
      if constexpr (POST) {
        if (step == 0) {
#pragma unroll
          for (int i = 0; i < PPT; ++i) {
            const int pix = tid + i * 256;
            xr[i] = pix < P ? x[sample * d.x_ld + pix] : 0.0;
          }
        }
      }
      if constexpr (REUP) {
        if (POST || step == 0) {
          // linear_down on the image: the launch's input, or ("noise" goal) the registers' current image every step.
          // (two partial buffers in turn: with six qubits no barrier separates one step's readers from the next one's
          //  writers)
          double* part = dbuf_parity ? s_part2 : s_part;
          dbuf_parity ^= 1;
          double acc[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = 0.0;
          if constexpr (POST) {
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
#pragma unroll
              for (int j = 0; j < N; ++j) acc[j] = fma(xr[i], wdr[i][j], acc[j]);
            }
          } else {
#pragma unroll 1
            for (int i = 0; i < PPT; ++i) {     // (pixel by pixel: once per launch and sample, keep its registers few)
              const int pix = tid + i * 256;
              const double xv = pix < P ? x[sample * d.x_ld + pix] : 0.0;
#pragma unroll
              for (int j = 0; j < N; ++j) acc[j] = fma(xv, pix < P ? wd[(size_t)j * P + pix] : 0.0, acc[j]);
            }
          }
          wave_reduce8_into<double, true>(acc, lane, llane, part + wv * 16);
          __syncthreads();
          if (lane < N) {
            const double h = part[lane] + part[16 + lane] + part[32 + lane] + part[48 + lane] +
                             (POST ? bdr : (bd ? bd[lane] : 0.0));
            s_xs[lane] = h * p.enc_scale;
          }
        } else {
          // x <- net(x) without a clamp: linear_down(linear_up(z)) = M z + v, z = the last step's <Z> (in registers);
          // lane j < 8 takes row j (every lane computes a row -- j = lane & 7 -- so nothing branches), two chains of four
          const double* mr = s_map + (lane & 7) * N;   // (rows beyond N read the table's padding: never stored)
          double h0 = s_map[N * N + (lane & 7)], h1 = 0.0;
#pragma unroll
          for (int i = 0; i < N; i += 2) {
            h0 = fma(mr[i], ev[i], h0);
            h1 = fma(mr[i + 1], ev[i + 1], h1);
          }
          if (lane < N) s_xs[lane] = (h0 + h1) * p.enc_scale;
        }
      }
      if (st) d.stamps[3] = __builtin_amdgcn_s_memtime();
      // ---- circuit rounds ------------------------------------------------------------------------------------
      for (int round = 0; round < p.n_rounds; ++round) {
        const int l0 = round * lpr;
        C dx{(T)1, (T)0};
        if constexpr (REUP) {
          T* s_cst = reinterpret_cast<T*>(s_cs);   // [8] cos(x_w / 2), then [8] sin(x_w / 2) (N used): this wave's copy
          if (lane < N) {
            if constexpr (sizeof(T) == 4) {
              float s, c;
              data_sincos_f32(0.5 * s_xs[lane], &s, &c);
              s_cst[lane] = c;
              s_cst[8 + lane] = s;
            } else {
              double s, c;
              sincos(0.5 * s_xs[lane], &s, &c);
              s_cst[lane] = c;
              s_cst[8 + lane] = s;
            }
          }
          wave_sync();
          // RZ(x) diagonal of this thread's amplitude, prod_q (cos + i sigma_q sin)(x_{7-q} / 2) with sigma = +-1 by the
          // thread's index bit: a tree of complex products (depth 3), not a chain of eight
          const V4* cs4 = reinterpret_cast<const V4*>(s_cst);
          const V4 c_lo = cs4[0], c_hi = cs4[1], s_lo = cs4[2], s_hi = cs4[3];   // wires 0..3 / 4..7
          const T cw[8] = {c_lo.x, c_lo.y, c_lo.z, c_lo.w, c_hi.x, c_hi.y, c_hi.z, c_hi.w};
          const T sw[8] = {s_lo.x, s_lo.y, s_lo.z, s_lo.w, s_hi.x, s_hi.y, s_hi.z, s_hi.w};
          C z[8];
#pragma unroll
          for (int q = 0; q < N; ++q) z[q] = C{cw[N - 1 - q], sw[N - 1 - q] * pm[q]};
          if constexpr (N == 8) {
#pragma unroll
            for (int q = 0; q < 4; ++q) z[q] = cmul2<T>(z[q], z[q + 4], times_i<T>(z[q + 4]));
            z[0] = cmul2<T>(z[0], z[2], times_i<T>(z[2]));
            z[1] = cmul2<T>(z[1], z[3], times_i<T>(z[3]));
            dx = cmul2<T>(z[0], z[1], times_i<T>(z[1]));
          } else {
#pragma unroll
            for (int q = 0; q < 3; ++q) z[q] = cmul2<T>(z[q], z[q + 3], times_i<T>(z[q + 3]));
            z[0] = cmul2<T>(z[0], z[1], times_i<T>(z[1]));
            dx = cmul2<T>(z[0], z[2], times_i<T>(z[2]));
          }
        }
        if (st && round == 0) d.stamps[7] = __builtin_amdgcn_s_memtime();
        C a{s_a0[round * TL + slot], (T)0};   // the round's first layer, generated
        LeanLayer<T> ca, cb;
        Raw raw;
        int next_upload = REUP ? p.sel_layers : 0x7fffffff;   // first layer of block 1
        // what layer li will multiply the state by: its phase entry, times the data diagonal at a block start (selected,
        // not branched on: the product rides in empty issue slots a layer ahead)
        auto derive_layer = [&](LeanLayer<T>& c, const Raw& r, int li) {
          c.ph = r.ph;
          if constexpr (REUP && LPR > 0) {
            // layers per round compiled in => two SEL layers per block (every LL / PL net of the reference; the host
            // checks): block starts are the even layers, a constant after unrolling -- no select, no multiply elsewhere
            if (li % 2 == 0 && li < LPR) c.ph = cmul2<T>(dx, r.ph, times_i<T>(r.ph));
          } else if constexpr (REUP) {
            const bool up = li == next_upload;
            if (up) next_upload += p.sel_layers;     // (scalar select, no branch)
            const C dxs = C{up ? dx.x : (T)1, up ? dx.y : (T)0};
            c.ph = cmul2<T>(dxs, r.ph, times_i<T>(r.ph));
          }
          c.ts[0] = r.lo.x * pm[0];
          c.ts[1] = r.lo.y * pm[1];
          c.ts[2] = r.lo.z * pm[2];
          c.ts[3] = r.lo.w * pm[3];
          c.t4 = r.hi.x;
          c.t5 = r.hi.y;
          if constexpr (N == 8) {
            c.k1 = r.hi.z * pm[6];
            c.k2 = r.hi.w * pm[7];
            c.k3 = c.k1 * c.k2;
          }
        };
        if (lpr > 1) {
          fetch_layer(raw, l0 + 1);
          derive_layer(ca, raw, 1);
        }
        auto layer = [&](const LeanLayer<T>& cur, LeanLayer<T>& nxt, int li) {
          // the next layer's phase and tangents first: they land while this layer's chain runs, and what is derived from
          // them fills issue slots the chain leaves empty -- nothing table-related is left behind the barrier
          // (unconditional -- the last layer re-reads its own entry -- so that the wait counters are exact on every path)
          fetch_layer(raw, l0 + (li + 1 < lpr ? li + 1 : li));
          __builtin_amdgcn_sched_barrier(0);
          a = cmul2<T>(cur.ph, a, times_i<T>(a));
          ry_t_dpp4(a, cur.ts[0], cur.ts[1], cur.ts[2], cur.ts[3]);
          if constexpr (N == 8) {
            // Bits 5 and 4 WITHOUT swapping back: after the permlane32 swap a lane holds (low, high) members of bit 5 of
            // one component; a permlane16 swap of THAT puts (low, high) members of bit 4 into every lane (rows 0 / 1 of
            // the real parts, rows 2 / 3 of the imaginary parts -- table at s_scatter below), so both 2 x 2 run
            // in-register.  The way back is free: the wave-bit exchange goes through LDS anyway, so the two values are
            // stored straight into their amplitudes' natural (re, im) slots (`ds_write2_b32`, second slot 16 amplitudes
            // up) and every thread reads its own amplitude back next to the three partners'.  Two swaps (and their wait
            // states) less in the chain of every layer.
            T lo = a.x, hi = a.y;
            swap_parts<5>(lo, hi);
            T nlo = fma(-cur.t5, hi, lo), nhi = fma(cur.t5, lo, hi);
            derive_layer(nxt, raw, li + 1);   // (the reads were issued ~100 cycles ago)
            swap_parts<4>(nlo, nhi);
            const T vx = fma(-cur.t4, nhi, nlo), vy = fma(cur.t4, nlo, nhi);
            C* buf = s_slab + (size_t)xbuf_parity * (4 * kWave);
            xbuf_parity ^= 1;
            T* slot = reinterpret_cast<T*>(buf + wv * kWave) + scatter_off;
            slot[0] = vx;
            slot[32] = vy;
            __syncthreads();
            const C p0 = buf[wv * kWave + lane];
            const C p1 = buf[(wv ^ 1) * kWave + lane];
            const C p2 = buf[(wv ^ 2) * kWave + lane];
            const C p3 = buf[(wv ^ 3) * kWave + lane];
            const C o = __builtin_elementwise_fma(bcast<T>(cur.k1), p1, p0);
            const C t = __builtin_elementwise_fma(bcast<T>(cur.k3), p3, bcast<T>(cur.k2) * p2);
            a = o + t;
          } else {
            ry_t_swap<5, T>(a, cur.t5);
            derive_layer(nxt, raw, li + 1);   // (the reads were issued ~100 cycles ago)
            ry_t_swap<4, T>(a, cur.t4);
          }
        };
        if constexpr (LPR > 0) {
#pragma unroll
          for (int li = 1; li + 1 < LPR; li += 2) {
            layer(ca, cb, li);
            layer(cb, ca, li + 1);
          }
          if constexpr (LPR > 1 && (LPR - 1) % 2 == 1) layer(ca, cb, LPR - 1);
        } else {
          int li = 1;
          for (; li + 1 < lpr; li += 2) {
            layer(ca, cb, li);
            layer(cb, ca, li + 1);
          }
          if (li < lpr) layer(ca, cb, li);
        }
        if (st && round == 0) d.stamps[4] = __builtin_amdgcn_s_memtime();
        // ---- <Z_w> (the ring and the RZ(omega) behind the last RY layer are diagonal) ----------------------
        const T pr = a.x * a.x + a.y * a.y;
        T ez[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) ez[w] = (w < N && ((kbase >> (N - 1 - w)) & 1u)) ? -pr : (w < N ? pr : (T)0);
        // (two partial buffers in turn: a round without simulated layers has no barrier between one read-out's readers and
        //  the next one's writers)
        T* s_pz = reinterpret_cast<T*>(s_part_z + zbuf_parity * 4 * 16);
        zbuf_parity ^= 1;
        wave_reduce8_into<T, true>(ez, lane, llane, s_pz + wv * 16);
        if constexpr (N == 8) {
          __syncthreads();
          // every thread adds the four waves' partials itself (eight broadcast reads, no second LDS round trip), pairwise,
          // in the engine's precision; float64 from there on
          const V4* pz = reinterpret_cast<const V4*>(s_pz);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const V4 q = (pz[h] + pz[4 + h]) + (pz[8 + h] + pz[12 + h]);
            ev[4 * h + 0] = (double)q.x;
            ev[4 * h + 1] = (double)q.y;
            ev[4 * h + 2] = (double)q.z;
            ev[4 * h + 3] = (double)q.w;
          }
        } else {
          // every wave summed its own copy of the whole state: its own eight slots, no barrier
          wave_sync();
          const V4* pz = reinterpret_cast<const V4*>(s_pz + wv * 16);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const V4 q = pz[h];
            ev[4 * h + 0] = (double)q.x;
            ev[4 * h + 1] = (double)q.y;
            ev[4 * h + 2] = (double)q.z;
            ev[4 * h + 3] = (double)q.w;
          }
        }
        if constexpr (REUP) {
          if (round + 1 < p.n_rounds) {   // next round's angles: lane j < 8 takes <Z_j> (a select chain, no indexing)
            double e = ev[0];
#pragma unroll
            for (int j = 1; j < N; ++j) e = lane == j ? ev[j] : e;
            if (lane < N) s_xs[lane] = e * p.enc_scale;
          }
        }
      }
      if (st) d.stamps[5] = __builtin_amdgcn_s_memtime();
      // ---- linear_up: this step's image.  The pixels of a thread advance together, each as two partial sums: eight
      //      independent chains of four (a float64 fma waits ~20 cycles on its predecessor) ---------------------------
      double o0[PPT], o1[PPT];
#pragma unroll
      for (int i = 0; i < PPT; ++i) {
        o0[i] = bur[i];
        o1[i] = 0.0;
      }
#pragma unroll
      for (int j = 0; j < N; j += 2) {
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
          o0[i] = fma(ev[j], wur[i][j], o0[i]);
          o1[i] = fma(ev[j + 1], wur[i][j + 1], o1[i]);
        }
      }
      double* yrow = y + (size_t)step * d.y_step_stride + sample * d.y_ld;
#pragma unroll
      for (int i = 0; i < PPT; ++i) {
        o0[i] += o1[i];
        if constexpr (POST) {
          o0[i] = fmin(fmax(xr[i] - (o0[i] - 0.5) * 0.1 * d.noise_factor, 0.0), 1.0);
          xr[i] = o0[i];
        }
        asm volatile("" : "+v"(o0[i]));   // (keeps the sums out of the stores' predicated blocks: all of them advance together)
      }
#pragma unroll
      for (int i = 0; i < PPT; ++i) {
        const int pix = tid + i * 256;
        if (Q >= 256 * (i + 1)) yrow[pix] = o0[i];   // whole group of 256 pixels inside the image: a scalar test
        else if (pix < Q) yrow[pix] = o0[i];
      }
      if (st) d.stamps[6] = __builtin_amdgcn_s_memtime();
    }
  }
}

}  // namespace qiddm
