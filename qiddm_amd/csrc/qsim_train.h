// qsim_train.h -- the training step of the denoise loop on the device.
//
// Reference: Diffusion.run_training_step_data / _noise (src/models.py:44-104) around a net of the
// linear_down -> circuit -> linear_up family (QNN_noise nn/qdense.py:267-289, QIDDM_LL_noise :1620-1642) with
// add_normal_noise_multiple (src/noise.py:105-126) as the forward-noising schedule:
//
//     whole  = noise_f(x, T+1, decay 3)                  (B, T+1, P)
//     noisy  = whole[:, 1:], clean = whole[:, :-1]       (B*T, P) rows i = b*T + t-1
//     out    = W_up <Z>(circuit(W_down noisy + b_down)) + b_up
//     loss   = mean((out - clean)^2)                                   goal "data"
//            = mean(((out - 0.5) * 0.1 - (noisy - clean))^2)           goal "noise"
//     loss.backward()
//
// Nothing of size (B*T, P) is materialised: the noisy / clean rows are re-derived from x (B, P) and the one
// noise field (B, P) wherever they are needed.  Four launches:
//   1a. train_project_kernel one wavefront per (sample, noise level): the blended row's dot products with the
//                            columns of W_down and W_up -- the only place the circuit stage touches pixels.
//   1b. train_rows_kernel    one wavefront per row: circuit forward (all rounds) -> d loss / d <Z> from the
//                            projections -> adjoint sweep -> d loss / d (W_down x).  Writes <Z> (rows, n), the
//                            input gradient (rows, n), per-workgroup K slabs.  No pixel loop, no weights in LDS.
//   2. train_weight_grads_kernel   the four weight/bias gradients are sums over rows of rank-1 terms; a thread
//                            owns one pixel column, walks a chunk of samples, recomputes out / residual / noisy in
//                            float64 and keeps 2n+1 accumulators.  Also the loss (and, on request, the
//                            reconstruction and the element-wise loss the verbose call returns).
//   3. train_finalize_kernel fixed-order sums of the chunk partials, the loss, b_down's gradient, and the
//                            contraction of K with dRot/d(phi, theta, omega).
// All reductions have a fixed order: the step is bit-reproducible run to run.
#pragma once
#include "qsim_adjoint.h"

#ifndef QIDDM_TRAIN_OCC
#define QIDDM_TRAIN_OCC 2  // waves per SIMD the reverse-sweep kernel is compiled for (register budget 512 / OCC).
                           // Measured at C2 (2560 rows) per step: folded sweep 1 -> 185 us, 2 -> 167, 3 -> 222;
                           // general sweep 1 -> 241, 2 -> 225, 3 -> 210.  The forward-only variant is fastest
                           // unconstrained (84 us vs 94-103)
#endif

namespace qiddm {

struct TrainScalars {
  int64_t x_ld, noise_ld, rows;  // rows = B * T
  int32_t pixels, T, goal, train_quantum;
  int32_t samples_per_chunk, n_chunks, want_recon, want_elem;
  int32_t fold, layers_per_round;  // fold: the reverse sweep runs on the folded tables (qsim_adjoint.h), slabs hold
                                   // per-layer gradient sums instead of K
  double grad_scale;  // d(mean loss) / d(out) = grad_scale * residual
};

// x * (1 - w) + noise * w, clamped -- in torch's order of roundings for a float64 x and float32 noise / w
// (src/noise.py:118-124): (1 - w) and noise * w are float32 results, the rest is float64, no contraction.
__device__ __forceinline__ double blend_noise(double xv, float nz, float w) {
  const float omw = __fsub_rn(1.0f, w);
  const float nw = __fmul_rn(nz, w);
  const double v = __dadd_rn(__dmul_rn(xv, (double)omw), (double)nw);
  return fmin(fmax(v, 0.0), 1.0);
}

// N(0.5, 0.2) field generated in the launch (optional): Philox4x32-10 keyed by the caller's seed, counter =
// (element index, step offset), Box-Muller on the first two words.  One value per element, so every kernel (and
// every wavefront that needs the element) derives the same number.
__device__ __forceinline__ float philox_normal(uint64_t seed, uint64_t offset, uint64_t index) {
  uint32_t c0 = (uint32_t)index, c1 = (uint32_t)(index >> 32), c2 = (uint32_t)offset, c3 = (uint32_t)(offset >> 32);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0;
    c1 = lo1;
    c2 = hi0 ^ c3 ^ k1;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  const float u1 = ((float)c0 + 1.0f) * 2.3283064365386963e-10f;  // (0, 1]
  const float u2 = (float)c1 * 2.3283064365386963e-10f;           // [0, 1): revolutions
  const float n = sqrtf(-2.0f * __logf(u1)) * __builtin_amdgcn_cosf(u2);
  return fmaf(n, 0.2f, 0.5f);
}

__device__ __forceinline__ double residual(int goal, double out, double noisy, double clean) {
  return goal == 0 ? out - clean : (out - 0.5) * 0.1 - (noisy - clean);
}

// ---------------------------------------------------------------------------
// 1a. projections.  Every place the (rows, P) tensors meet a weight matrix is a dot product of a blended row
// with a column of W_down or W_up, so they are all taken here, once, one workgroup per (sample b, four levels t):
//     proj[(b*(T+1)+t)*2n + j]     = sum_p whole[b,t,p] W_down[j,p]            j < n
//     proj[(b*(T+1)+t)*2n + n + j] = sum_p whole[b,t,p] W_up[p,j]
// and n+2 more units with W_up[:,j'], b_up and 1 in place of `whole` (G = W_up^T W_up, c = W_up^T b_up,
// s = W_up^T 1) at unit index B*(T+1) + {0..n-1, n, n+1}.  With these the circuit kernel needs no pixel loop:
//     d loss / d <Z_j> = grad_scale * ((G ev + c)_j - CU_j)                                  goal "data"
//                      = grad_scale * (0.1 (G ev + c)_j - 0.05 s_j - NU_j + CU_j)            goal "noise"
// (NU / CU = the W_up projections of the noisy / clean row).
// ---------------------------------------------------------------------------
constexpr int kProjLevels = 4;  // noise levels per workgroup: each weight fetched once serves this many rows
constexpr int kProjWaves = 4;   // wavefronts per unit, each takes every fourth 64-pixel strip

// a + b summed across the lane pair that differs in bit 4 (MASK 16) / 5 (MASK 32), for two values at once: the lane
// with the bit clear gets  a(own) + a(partner), the lane with the bit set  b(own) + b(partner).  One permlane swap per
// dword exchanges a's set-bit half with b's clear-bit half, after which both sums are simply a + b.
template <int MASK>
__device__ __forceinline__ double swap_pair_sum(double a, double b) {
  uint32_t alo = (uint32_t)__double2loint(a), ahi = (uint32_t)__double2hiint(a);
  uint32_t blo = (uint32_t)__double2loint(b), bhi = (uint32_t)__double2hiint(b);
  if constexpr (MASK == 32) {
    const auto lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    alo = lo[0], blo = lo[1], ahi = hi[0], bhi = hi[1];
  } else {
    static_assert(MASK == 16, "swap steps are the two row-crossing lane bits");
    const auto lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    alo = lo[0], blo = lo[1], ahi = hi[0], bhi = hi[1];
  }
  return __hiloint2double((int)ahi, (int)alo) + __hiloint2double((int)bhi, (int)blo);
}

// Sums of 64 per-lane values over the wavefront by halving: after the step on lane bit k only the half of the values
// whose index has bit k equal to the lane's stays in the lane, so 63 exchanges do what 64 butterflies (384) would.
// The lane with LOGICAL number L returns the total of v[L].  Destroys v.
__device__ __forceinline__ double wave_transpose_sum64(double (&v)[64], int lane, int llane) {
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = swap_pair_sum<32>(v[i], v[32 + i]);
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = swap_pair_sum<16>(v[i], v[16 + i]);
  const bool b3 = llane & 8, b2 = llane & 4, b1 = llane & 2, b0 = llane & 1;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (b3 ? v[8 + i] : v[i]) + xlane<8>(b3 ? v[i] : v[8 + i], lane);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = (b2 ? v[4 + i] : v[i]) + xlane<4>(b2 ? v[i] : v[4 + i], lane);
#pragma unroll
  for (int i = 0; i < 2; ++i) v[i] = (b1 ? v[2 + i] : v[i]) + xlane<2>(b1 ? v[i] : v[2 + i], lane);
  return (b0 ? v[1] : v[0]) + xlane<1>(b0 ? v[0] : v[1], lane);
}

template <int N>
__global__ __launch_bounds__(kProjWaves* kWave) void train_project_kernel(
    const double* __restrict__ x, float* __restrict__ noise, const uint64_t* __restrict__ rng,
    const float* __restrict__ sched, const double* __restrict__ wd, const double* __restrict__ wu,
    const double* __restrict__ bu, double* __restrict__ proj, int64_t batch, const TrainScalars d) {
  constexpr int LG = kProjLevels;
  __shared__ double s_part[kProjWaves][LG * 2 * N];
  const int P = d.pixels;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const int groups_per_sample = (d.T + 1 + LG - 1) / LG;
  const int64_t unit = blockIdx.x;  // grid = n_blend (+ N + 2 when the quantum weights train)
  const int64_t n_blend = batch * groups_per_sample;
  // (as written, finding F1, nothing downstream reads the W_up projections: only W_down x is taken)
  const bool quantum = d.train_quantum != 0;
  const int kind = unit < n_blend ? 0 : (int)(unit - n_blend) + 1;  // 0 blend, 1..N W_up column, N+1 b_up, N+2 ones
  const int64_t b = kind == 0 ? unit / groups_per_sample : 0;
  const int t0 = kind == 0 ? (int)(unit - b * groups_per_sample) * LG : 0;
  float w[LG];
#pragma unroll
  for (int l = 0; l < LG; ++l) w[l] = sched[t0 + l <= d.T ? t0 + l : d.T];
  double acc[LG][2 * N];
#pragma unroll
  for (int l = 0; l < LG; ++l)
#pragma unroll
    for (int j = 0; j < 2 * N; ++j) acc[l][j] = 0.0;
  // 3120 wavefronts at the benchmark shapes (three per SIMD), three or four strips each
#pragma unroll 2
  for (int pix = wave * kWave + lane; pix < P; pix += kProjWaves * kWave) {
    double wdv[N], wuv[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
      wdv[j] = wd[(size_t)j * P + pix];
      wuv[j] = quantum ? wu[(size_t)pix * N + j] : 0.0;
    }
    if (kind == 0) {
      const double xv = x[b * d.x_ld + pix];
      float nz;
      if (rng != nullptr) {  // generate the field here; the first level group of the sample records it
        nz = philox_normal(rng[0], rng[1], (uint64_t)b * (uint64_t)P + (uint64_t)pix);
        if (t0 == 0) noise[b * d.noise_ld + pix] = nz;
      } else {
        nz = noise[b * d.noise_ld + pix];
      }
#pragma unroll
      for (int l = 0; l < LG; ++l) {
        const double v = blend_noise(xv, nz, w[l]);
#pragma unroll
        for (int j = 0; j < N; ++j) acc[l][j] = fma(v, wdv[j], acc[l][j]);
        if (quantum) {
#pragma unroll
          for (int j = 0; j < N; ++j) acc[l][N + j] = fma(v, wuv[j], acc[l][N + j]);
        }
      }
    } else {
      const double v = kind <= N ? wu[(size_t)pix * N + (kind - 1)] : (kind == N + 1 ? (bu ? bu[pix] : 0.0) : 1.0);
#pragma unroll
      for (int j = 0; j < N; ++j) acc[0][N + j] = fma(v, wuv[j], acc[0][N + j]);
    }
  }
  // this wavefront's strip sums -> s_part[wave][l * 2N + j]
  if constexpr (LG == 4 && N <= 8) {  // value (l, j) at index 16 l + j
    const int llane = logical_lane(lane);
    double v[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) v[i] = (i & 15) < 2 * N ? acc[i >> 4][(i & 15) < 2 * N ? (i & 15) : 0] : 0.0;
    const double tot = wave_transpose_sum64(v, lane, llane);
    if ((llane & 15) < 2 * N) s_part[wave][(llane >> 4) * 2 * N + (llane & 15)] = tot;
  } else {
#pragma unroll
    for (int l = 0; l < LG; ++l) {
      double mine = 0.0;
#pragma unroll
      for (int j = 0; j < 2 * N; ++j) {
        const double tot = group_sum<double, 6>(acc[l][j], lane);
        mine = fma((double)(lane == j ? 1 : 0), tot, mine);
      }
      if (lane < 2 * N) s_part[wave][l * 2 * N + lane] = mine;
    }
  }
  __syncthreads();
  // the strips in a fixed order
  for (int i = threadIdx.x; i < LG * 2 * N; i += blockDim.x) {
    double tot = s_part[0][i];
#pragma unroll
    for (int wv = 1; wv < kProjWaves; ++wv) tot += s_part[wv][i];
    const int l = i / (2 * N), j = i - l * (2 * N);
    if (kind == 0) {
      if (t0 + l <= d.T) proj[(b * (d.T + 1) + t0 + l) * (2 * N) + j] = tot;
    } else if (l == 0) {
      proj[(batch * (d.T + 1) + (kind - 1)) * (2 * N) + j] = tot;
    }
  }
}

template <typename T, int N>
__host__ __device__ inline size_t train_lds_bytes(int64_t n_rot_all, int n_rounds, bool cnot, int waves,
                                                  bool quantum) {
  size_t b = Smem<T, N>::bytes(n_rot_all, cnot, waves);
  if (quantum) b += (size_t)n_rot_all * kLdsGateReals * sizeof(T) + (size_t)waves * n_rot_all * 8 * sizeof(T);
  b += (size_t)waves * Layout<N>::SPW * n_rounds * N * sizeof(T);
  if (quantum) {  // G, c, s as doubles (8-byte aligned) + one N-vector per sample in flight to hand d loss / d <Z> around
    b = (b + 7) & ~(size_t)7;
    b += (size_t)(N + 2) * N * sizeof(double) + (size_t)waves * Layout<N>::SPW * N * sizeof(T);
  }
  return b;
}

// ---------------------------------------------------------------------------
// 1b. the circuit, forward and (QUANTUM) reverse, one wavefront per row
// ---------------------------------------------------------------------------
// FOLD (compile time, = d.fold): the folded forward / reverse sweep of CZ circuits; keeping the general gate-by-gate
// path out of that instantiation is worth registers (n = 8: 256 VGPRs + 65 spills with both paths compiled in)
template <typename T, int N, bool QUANTUM, int WPB, bool FOLD>
__global__ __launch_bounds__(WPB* kWave, QUANTUM ? QIDDM_TRAIN_OCC : 1) void train_rows_kernel(
    const double* __restrict__ proj, const double* __restrict__ bd, const double* __restrict__ angles,
    double* __restrict__ ev_out, double* __restrict__ gxr_out, T* __restrict__ k_partials, int64_t batch,
    const TrainScalars d, const KScalars p) {
  using E = Engine<T, N>;
  using L = typename E::L;
  using C = V2<T>;
  constexpr int LB = L::LB, R = L::R, SPW = L::SPW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int n_rot = p.n_blocks * p.sel_layers * N;  // per round
  const int n_rot_all = p.n_rounds * n_rot;
  const bool use_cnot = p.imprimitive == 0;

  AdjointEngine<T, N> adj;
  adj.fwd.carve(smem_raw, n_rot_all);
  unsigned char* cur = smem_raw + Smem<T, N>::bytes(n_rot_all, use_cnot, WPB);
  T* dag_gates = reinterpret_cast<T*>(cur);
  T* kall = dag_gates;
  if constexpr (QUANTUM) {
    kall = dag_gates + (size_t)n_rot_all * kLdsGateReals;
    cur = reinterpret_cast<unsigned char*>(kall + (size_t)WPB * n_rot_all * 8);
  }
  T* hist_all = reinterpret_cast<T*>(cur);
  cur += (size_t)WPB * SPW * p.n_rounds * N * sizeof(T);
  cur = smem_raw + (((size_t)(cur - smem_raw) + 7) & ~(size_t)7);
  double* s_gram = reinterpret_cast<double*>(cur);  // [N][N] G (row j' at s_gram[j' * N + j]), then c[N], s[N]
  T* s_gw_all = reinterpret_cast<T*>(s_gram + (N + 2) * N);
  constexpr bool folded = FOLD && N >= 2 && N <= kFoldedAdjointMaxQubits;   // (QUANTUM = false: the forward alone)
  const int layers = p.n_blocks * p.sel_layers;  // per round
  using AE = AdjointEngine<T, N>;
  const int acc_len = folded ? p.n_rounds * layers * 2 * AE::kFoldSlots : n_rot_all * 8;
  if constexpr (folded)
    adj.fwd.fill_folded_from_angles(angles, p.n_rounds * layers, layers);
  else
    adj.fwd.fill_gates_from_angles(angles, n_rot_all);
  adj.fwd.fill_rings(use_cnot);
  if constexpr (QUANTUM) {
    for (int i = threadIdx.x; i < WPB * acc_len; i += blockDim.x) kall[i] = 0;
    const double* __restrict__ g0 = proj + batch * (d.T + 1) * (2 * N) + N;  // unit u's W_up half at g0[u * 2N + j]
    for (int i = threadIdx.x; i < (N + 2) * N; i += blockDim.x) s_gram[i] = g0[(size_t)(i / N) * (2 * N) + (i % N)];
  }
  __syncthreads();
  if constexpr (QUANTUM) {
    if constexpr (!folded) {
      // U^dagger images from the forward ones: (u00*, u10*; u01*, u11*)
      for (int g = threadIdx.x; g < n_rot_all; g += blockDim.x) {
        const T* f = adj.fwd.s_gates_w + (size_t)g * kLdsGateReals;
        E::put_gate(dag_gates + (size_t)g * kLdsGateReals, f[0], -f[1], f[12], -f[13], f[4], -f[5], f[8], -f[9]);
      }
    }
    __syncthreads();
  }
  const size_t round_gate_stride = folded ? (size_t)layers * Smem<T, N>::kFoldStride : (size_t)n_rot * kLdsGateReals;
  const size_t round_acc_stride = folded ? (size_t)layers * 2 * AE::kFoldSlots : (size_t)n_rot * 8;
  adj.dag = adj.fwd;
  const T* fwd_base = adj.fwd.s_gates;
  const int lane = adj.fwd.lane, sub = adj.fwd.sub;
  const int wave = threadIdx.x >> 6;
  const int swave = adj.fwd.llane >> LB;
  T* kacc_wave = kall + (size_t)wave * acc_len;
  T* hist = hist_all + (size_t)(wave * SPW + swave) * p.n_rounds * N;
  T* s_gw = s_gw_all + (size_t)(wave * SPW + swave) * N;

  const int64_t groups = (d.rows + SPW - 1) / SPW;
  for (int64_t grp = (int64_t)blockIdx.x * WPB + wave; grp < groups; grp += (int64_t)gridDim.x * WPB) {
    const int64_t row_raw = grp * SPW + swave;
    const bool valid = row_raw < d.rows;
    const int64_t row = valid ? row_raw : d.rows - 1;
    const int64_t b = row / d.T;
    const int t = (int)(row - b * d.T) + 1;
    const double* __restrict__ pn = proj + (b * (d.T + 1) + t) * (2 * N);  // noisy row's projections
    const double* __restrict__ pcl = pn - 2 * N;                           // clean row's (level t-1)
    T xs[N];
#pragma unroll
    for (int j = 0; j < N; ++j) xs[j] = (T)((pn[j] + (bd ? bd[j] : 0.0)) * p.enc_scale);

    // ---- the circuit, all rounds; the last round's state stays in registers for the reverse sweep ----
    C psi[R], dx[R];
    T cs[N], sn[N], amp_inv;
    T result[N];
    for (int round = 0; round < p.n_rounds; ++round) {
      if (QUANTUM && sub == 0) {
#pragma unroll
        for (int j = 0; j < N; ++j) hist[round * N + j] = xs[j];
      }
      adj.fwd.s_gates = fwd_base + (size_t)round * round_gate_stride;
      if constexpr (folded)
        adj.forward_round_folded(p, NoSrc{}, xs, psi, dx, cs, sn, amp_inv, layers);
      else
        adj.forward_round(p, NoSrc{}, xs, psi, dx, cs, sn, amp_inv);
      adj.measure_expz(psi, result);
      if (round + 1 < p.n_rounds) {
#pragma unroll
        for (int j = 0; j < N; ++j) xs[j] = result[j] * (T)p.enc_scale;
      }
    }
    {
      double v = 0.0;
#pragma unroll
      for (int j = 0; j < N; ++j) v = fma((double)(sub == j ? 1 : 0), (double)result[j], v);
      if (valid && sub < N) ev_out[row * N + sub] = v;
    }
    if constexpr (QUANTUM) {
      // ---- d loss / d <Z_j> from the projections ---------------------------------------------------
      // lane j of the sample takes component j (one column of G from LDS, 2n + 1 flops), the n results go round
      // through LDS: the whole matrix in scalar registers cost ~1000 v_readlane of spilled SGPRs per row
      T gw[N];
      {
        const int j = sub < N ? sub : 0;
        double ge = s_gram[N * N + j];
#pragma unroll
        for (int jp = 0; jp < N; ++jp) ge = fma(s_gram[jp * N + j], (double)result[jp], ge);
        const double g =
            d.goal == 0 ? ge - pcl[N + j] : 0.1 * ge - 0.05 * s_gram[(N + 1) * N + j] - pn[N + j] + pcl[N + j];
        if (sub < N) s_gw[sub] = valid ? (T)(d.grad_scale * g) : (T)0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // LDS hand-over inside the wavefront
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int jj = 0; jj < N; ++jj) gw[jj] = s_gw[jj];
      }
      // ---- reverse sweep, last round first; earlier rounds are re-run forward from their recorded inputs ----
      for (int round = p.n_rounds - 1; round >= 0; --round) {
        adj.fwd.s_gates = fwd_base + (size_t)round * round_gate_stride;
        adj.dag.s_gates = dag_gates + (size_t)round * n_rot * kLdsGateReals;
        adj.kacc = kacc_wave + (size_t)round * round_acc_stride;
        if (round != p.n_rounds - 1) {
#pragma unroll
          for (int j = 0; j < N; ++j) xs[j] = hist[round * N + j];
          if constexpr (folded)
            adj.forward_round_folded(p, NoSrc{}, xs, psi, dx, cs, sn, amp_inv, layers);
          else
            adj.forward_round(p, NoSrc{}, xs, psi, dx, cs, sn, amp_inv);
        }
        C lam[R];
        adj.seed_expz(gw, psi, lam);
        T gx[N];
        if constexpr (folded)
          adj.reverse_round_folded(p, psi, lam, cs, sn, gx, adj.kacc);
        else
          adj.reverse_round(p, psi, lam, dx, cs, sn, gx);
#pragma unroll
        for (int j = 0; j < N; ++j) gw[j] = group_sum<T, LB>(gx[j], lane) * (T)p.enc_scale;
      }
      double v = 0.0;
#pragma unroll
      for (int j = 0; j < N; ++j) v = fma((double)(sub == j ? 1 : 0), (double)gw[j], v);
      if (valid && sub < N) gxr_out[row * N + sub] = v;
    }
  }
  if constexpr (QUANTUM) {
    __syncthreads();
    for (int i = threadIdx.x; i < acc_len; i += blockDim.x) {  // slab stride n_rot_all * 8 either way
      T tot = 0;
      for (int w = 0; w < WPB; ++w) tot += kall[(size_t)w * acc_len + i];
      k_partials[(size_t)blockIdx.x * n_rot_all * 8 + i] = tot;
    }
  }
}

// ---------------------------------------------------------------------------
// weight / bias gradient partials, loss partials, optional verbose outputs.  grid (pixel tiles, chunks), 64 threads.
// partials layout [chunk][2n+1][P]: rows 0..n-1 dW_up[:, j], row n db_up, rows n+1.. dW_down[j, :]
// ---------------------------------------------------------------------------
constexpr int kGradWaves = 4;

template <int N>
__global__ __launch_bounds__(kGradWaves* kWave) void train_weight_grads_kernel(
    const double* __restrict__ x, const float* __restrict__ noise, const float* __restrict__ sched,
    const double* __restrict__ wu, const double* __restrict__ bu, const double* __restrict__ ev,
    const double* __restrict__ gxr, double* __restrict__ partials, double* __restrict__ loss_partials,
    double* __restrict__ recon, double* __restrict__ elem, int64_t batch, const TrainScalars d) {
  __shared__ double s_red[kGradWaves - 1][2 * N + 2][kWave];
  // the <Z> and input-gradient rows of the sample a wave is on (T x n doubles each): fetched with ONE coalesced load per
  // sample and read back as LDS broadcasts -- read row by row from global memory they are wave-uniform scalar loads, a
  // dependent L2 round trip per noise level (the wave sat 9.6 us on 565 instructions)
  constexpr int kRowCap = 32 * N;   // T <= 32 levels staged; beyond that the rows are read in place
  __shared__ double s_rows[kGradWaves][2][kRowCap];
  const int P = d.pixels;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const bool staged = d.T * N <= kRowCap;
  const int pix = blockIdx.x * kWave + lane;
  const bool pvalid = pix < P;
  const int pc = pvalid ? pix : 0;
  const int chunk = blockIdx.y;
  const int64_t b0 = (int64_t)chunk * d.samples_per_chunk;
  const int64_t b1 = b0 + d.samples_per_chunk < batch ? b0 + d.samples_per_chunk : batch;
  const bool quantum = d.train_quantum != 0;
  double wur[N], acc_wu[N], acc_wd[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    wur[j] = wu[(size_t)pc * N + j];
    acc_wu[j] = 0.0;
    acc_wd[j] = 0.0;
  }
  const double buv = bu ? bu[pc] : 0.0;
  double acc_bu = 0.0, loss = 0.0;
  for (int64_t b = b0 + wave; b < b1; b += kGradWaves) {  // the chunk's samples are dealt to the waves
    const double xv = x[b * d.x_ld + pc];
    const float nz = noise[b * d.noise_ld + pc];
    if (staged) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the previous sample's readers are done
      __builtin_amdgcn_wave_barrier();
      for (int i = lane; i < d.T * N; i += kWave) {
        s_rows[wave][0][i] = ev[b * d.T * N + i];
        s_rows[wave][1][i] = quantum ? gxr[b * d.T * N + i] : 0.0;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // one loop per source of the rows (two instantiations: the LDS one reads with ds_read broadcasts, not flat loads)
    auto levels = [&](const double* __restrict__ ev_rows, const double* __restrict__ gx_rows) {
      double clean = blend_noise(xv, nz, sched[0]);
      for (int t = 1; t <= d.T; ++t) {
        const double noisy = blend_noise(xv, nz, sched[t]);
        const int64_t row = b * d.T + (t - 1);
        const double* __restrict__ evp = ev_rows + (t - 1) * N;
        double o = buv;
#pragma unroll
        for (int j = 0; j < N; ++j) o = fma(evp[j], wur[j], o);
        const double r = residual(d.goal, o, noisy, clean);
        const double g = d.grad_scale * r;
        loss = fma(r, r, loss);
        acc_bu += g;
#pragma unroll
        for (int j = 0; j < N; ++j) acc_wu[j] = fma(g, evp[j], acc_wu[j]);
        if (quantum) {
          const double* __restrict__ gp = gx_rows + (t - 1) * N;
#pragma unroll
          for (int j = 0; j < N; ++j) acc_wd[j] = fma(gp[j], noisy, acc_wd[j]);
        }
        if (pvalid) {
          if (d.want_recon)
            recon[row * P + pix] = d.goal == 0 ? o : fmin(fmax(noisy - (o - 0.5) * 0.1, 0.0), 1.0);
          if (d.want_elem) elem[row * P + pix] = r * r;
        }
        clean = noisy;
      }
    };
    if (staged) levels(&s_rows[wave][0][0], &s_rows[wave][1][0]);
    else levels(ev + b * d.T * N, gxr + b * d.T * N);
  }
  // waves 1.. hand their sums to wave 0 (fixed order)
  if (wave > 0) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
      s_red[wave - 1][j][lane] = acc_wu[j];
      s_red[wave - 1][N + 1 + j][lane] = acc_wd[j];
    }
    s_red[wave - 1][N][lane] = acc_bu;
    s_red[wave - 1][2 * N + 1][lane] = loss;
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 0; w < kGradWaves - 1; ++w) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
      acc_wu[j] += s_red[w][j][lane];
      acc_wd[j] += s_red[w][N + 1 + j][lane];
    }
    acc_bu += s_red[w][N][lane];
    loss += s_red[w][2 * N + 1][lane];
  }
  if (pvalid) {
    double* dst = partials + (size_t)chunk * (2 * N + 1) * P + pix;
#pragma unroll
    for (int j = 0; j < N; ++j) dst[(size_t)j * P] = acc_wu[j];
    dst[(size_t)N * P] = acc_bu;
    if (quantum) {
#pragma unroll
      for (int j = 0; j < N; ++j) dst[(size_t)(N + 1 + j) * P] = acc_wd[j];
    }
  }
  loss = group_sum<double, 6>(pvalid ? loss : 0.0, lane);
  if (lane == 0) loss_partials[(size_t)chunk * gridDim.x + blockIdx.x] = loss;
}

// roles by workgroup: [0, wblocks) weight sums; wblocks: loss; n x db_down[j]; then one wavefront per Rot gate
template <typename T>
__global__ __launch_bounds__(kWave) void train_finalize_kernel(
    const double* __restrict__ partials, const double* __restrict__ loss_partials, int64_t n_loss_partials,
    const double* __restrict__ gxr, const T* __restrict__ k_partials, int64_t n_k_partials,
    const double* __restrict__ angles, int n, int64_t n_rot_all, int wblocks, double* __restrict__ loss,
    double* __restrict__ g_wd, double* __restrict__ g_bd, double* __restrict__ g_angles,
    double* __restrict__ g_wu, double* __restrict__ g_bu, uint64_t* __restrict__ rng, const TrainScalars d) {
  const int P = d.pixels;
  const int lane = threadIdx.x;
  const int role = blockIdx.x;
  if (role < wblocks) {
    const int rows = d.train_quantum ? 2 * n + 1 : n + 1;
    const int64_t idx = (int64_t)role * kWave + lane;
    if (idx >= (int64_t)rows * P) return;
    const int k = (int)(idx / P), pix = (int)(idx - (int64_t)k * P);
    double tot = 0.0;
#pragma unroll 8
    for (int c = 0; c < d.n_chunks; ++c) tot += partials[((size_t)c * (2 * n + 1) + k) * P + pix];
    if (k < n) g_wu[(size_t)pix * n + k] = tot;
    else if (k == n) g_bu[pix] = tot;
    else g_wd[(size_t)(k - n - 1) * P + pix] = tot;
    return;
  }
  if (role == wblocks) {
    double tot = 0.0;
#pragma unroll 8
    for (int64_t i = lane; i < n_loss_partials; i += kWave) tot += loss_partials[i];
    tot = group_sum<double, 6>(tot, lane);
    if (lane == 0) {
      loss[0] = tot / ((double)d.rows * (double)P);
      if (rng != nullptr) rng[1] += 1;  // next step draws a fresh field
    }
    return;
  }
  if (role <= wblocks + n) {  // db_down[j]: one wavefront per j
    if (!d.train_quantum) return;
    const int j = role - wblocks - 1;
    double sum = 0.0;
#pragma unroll 8
    for (int64_t r = lane; r < d.rows; r += kWave) sum += gxr[r * n + j];
    sum = group_sum<double, 6>(sum, lane);
    if (lane == 0) g_bd[j] = sum;
    return;
  }
  const int64_t g = role - wblocks - 1 - n;
  if (g >= n_rot_all) return;
  if (d.fold) {
    // folded slabs: [layer][2][slots] sums of d/dtheta, d/dalpha per wire (adjoint_finalize_folded_kernel)
    const int slots = n <= 8 ? 8 : 16;
    const int layer = (int)(g / n), w = (int)(g - (int64_t)layer * n);
    const bool has_next = (layer % d.layers_per_round) + 1 < d.layers_per_round;
    double th = 0, al = 0, an = 0;
#pragma unroll 4
    for (int64_t pidx = lane; pidx < n_k_partials; pidx += kWave) {
      const T* src = k_partials + pidx * n_rot_all * 8 + (size_t)layer * 2 * slots;
      th += (double)src[w];
      al += (double)src[slots + w];
      if (has_next) an += (double)src[3 * slots + w];
    }
    th = group_sum<double, 6>(th, lane);
    al = group_sum<double, 6>(al, lane);
    an = group_sum<double, 6>(an, lane);
    if (lane == 0) {
      g_angles[g * 3 + 0] = al;
      g_angles[g * 3 + 1] = th;
      g_angles[g * 3 + 2] = an;
    }
    return;
  }
  double k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 4
  for (int64_t pidx = lane; pidx < n_k_partials; pidx += kWave) {
    const T* src = k_partials + (pidx * n_rot_all + g) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) k[i] += (double)src[i];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) k[i] = group_sum<double, 6>(k[i], lane);
  if (lane == 0) rot_grad_from_k(k, angles[g * 3 + 0], angles[g * 3 + 1], angles[g * 3 + 2], g_angles + g * 3);
}

// ---------------------------------------------------------------------------
// Adam (torch.optim.Adam defaults of the reference harness, src/mnist_exm.py:170): every parameter tensor of the
// model in ONE launch.  The per-tensor step counters (torch keeps one per parameter) live on the device so a
// recorded graph advances them; the last workgroup to finish bumps them (every workgroup reads its counter
// before signalling).
// ---------------------------------------------------------------------------
constexpr int kAdamMaxTensors = 16;
struct AdamBatch {
  void* param[kAdamMaxTensors];
  const void* grad[kAdamMaxTensors];
  void* exp_avg[kAdamMaxTensors];
  void* exp_avg_sq[kAdamMaxTensors];
  int64_t* step[kAdamMaxTensors];
  int64_t block_end[kAdamMaxTensors];  // exclusive prefix of 256-element blocks
  int64_t numel[kAdamMaxTensors];
  int32_t is_f64[kAdamMaxTensors];
  int32_t n_tensors;
  double lr, beta1, beta2, eps, weight_decay;
};

template <typename T>
__device__ __forceinline__ void adam_update(T* p, const T* g, T* m, T* v, int64_t i, const AdamBatch& a, double bc1,
                                            double bc2_sqrt) {
  double grad = (double)g[i];
  const double prm = (double)p[i];
  if (a.weight_decay != 0.0) grad = fma(a.weight_decay, prm, grad);
  const double m0 = (double)m[i], v0 = (double)v[i];
  const double m1 = m0 + (1.0 - a.beta1) * (grad - m0);               // exp_avg.lerp_(grad, 1 - beta1)
  const double v1 = a.beta2 * v0 + (1.0 - a.beta2) * grad * grad;     // mul_(beta2).addcmul_(grad, grad, 1 - beta2)
  const double denom = sqrt(v1) / bc2_sqrt + a.eps;
  m[i] = (T)m1;
  v[i] = (T)v1;
  p[i] = (T)(prm - (a.lr / bc1) * (m1 / denom));
}

__global__ __launch_bounds__(256) void adam_step_kernel(const AdamBatch a, unsigned int* __restrict__ sync) {
  int t = 0;
  while (t < a.n_tensors - 1 && (int64_t)blockIdx.x >= a.block_end[t]) ++t;
  const int64_t s = *a.step[t] + 1;
  const double bc1 = 1.0 - pow(a.beta1, (double)s);
  const double bc2_sqrt = sqrt(1.0 - pow(a.beta2, (double)s));
  const int64_t first = t == 0 ? 0 : a.block_end[t - 1];
  const int64_t i = ((int64_t)blockIdx.x - first) * 256 + threadIdx.x;
  if (i < a.numel[t]) {
    if (a.is_f64[t])
      adam_update<double>(static_cast<double*>(a.param[t]), static_cast<const double*>(a.grad[t]),
                          static_cast<double*>(a.exp_avg[t]), static_cast<double*>(a.exp_avg_sq[t]), i, a, bc1, bc2_sqrt);
    else
      adam_update<float>(static_cast<float*>(a.param[t]), static_cast<const float*>(a.grad[t]),
                         static_cast<float*>(a.exp_avg[t]), static_cast<float*>(a.exp_avg_sq[t]), i, a, bc1, bc2_sqrt);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned int done = atomicAdd(sync, 1u);
    if (done == gridDim.x - 1) {
      for (int k = 0; k < a.n_tensors; ++k) *a.step[k] += 1;
      *sync = 0u;
    }
  }
}

}  // namespace qiddm
