// qsim_fused.h -- fused whole-circuit statevector engine for gfx950 (CDNA4).
//
// One wavefront (64 lanes) owns the complete 2^n-amplitude slab of a sample in
// registers for the whole circuit (n <= 10): amplitude index k = (r << LB) | sub
// with sub = lane bits (LB = min(n, 6)) and r = register index (R = 2^(n-LB)
// complex amplitudes per lane, each a packed (re, im) register pair so that the
// complex arithmetic issues as v_pk_fma_f32).  For n < 6 a wave carries 64/2^n
// samples.  Wire w is bit q = n-1-w of k.
//
//   * gate on a register bit : pure packed-FMA work, 8 v_pk_fma per amplitude pair
//   * gate on a lane bit     : partner amplitude fetched with VALU cross-lane moves
//                              (DPP quad_perm / row_ror / row_half_mirror,
//                              v_permlane16_swap, v_permlane32_swap -- no LDS
//                              round trip), 4 v_pk_fma per amplitude
//   * gate matrices          : staged once per workgroup in LDS as
//                              [u00, i*u00, u01, i*u01 | u11, i*u11, u10, i*u10] so a lane
//                              reads the half that matches its bit: no selects
//   * CZ ring                : precomputed sign-bit masks (LDS)
//   * CNOT ring              : GF(2)-linear index map, scatter through a per-wave
//                              LDS slab
//   * RZ data re-upload      : one per-sample diagonal, built once per round
//
// HBM is touched twice per sample: inputs in, probabilities / <Z> out.
//
// Replaces the per-gate torch op chains of PennyLane default.qubit.torch and the
// per-sample lightning.qubit loop (reference nn/qdense.py:278-281, 464-465,
// 1437-1441, 1631-1635).  Semantics of every op: oracle/statevector.py.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qiddm {

constexpr int kWave = 64;
constexpr int kMaxWavesPerBlock = 8;  // workgroups are 4 or 8 waves (chosen by the host)
constexpr int kMaxBlock = kWave * kMaxWavesPerBlock;
constexpr int kVariants = 7;       // global gate table: base + six parameter shifts
constexpr int kGateReals = 8;      // global gate table: u00 u01 u10 u11 as (re, im)
constexpr int kLdsGateReals = 16;  // LDS gate table, see above

template <typename T>
using V2 = T __attribute__((ext_vector_type(2)));

template <int N, int LBMAX = 6>
struct Layout {
  static constexpr int LB = N < LBMAX ? N : LBMAX;  // lane bits
  static constexpr int R = 1 << (N - LB);           // complex amplitudes per lane
  static constexpr int LPS = 1 << LB;               // lanes per sample
  static constexpr int SPW = kWave / LPS;           // samples per wave
  static constexpr int D = 1 << N;
  static constexpr int NR = N > 1 ? N - 1 : 1;      // distinct entangler ranges
  static_assert(R <= 32, "CZ sign masks are packed in one dword per lane");
};

// scalar launch parameters (by value in the kernarg segment)
struct KScalars {
  int64_t in_ld, out_ld, g_ld, batch;
  int32_t first_replica;
  int32_t encoding, imprimitive, measure, n_rounds, n_blocks, sel_layers, n_features;
  int32_t fold;        // run the forward of a CZ circuit on the folded tables (host: can_fold())
  int32_t post_cols;   // > 0 (probabilities only): `out` is float64 (B, post_cols) = clamp(p[:, :post_cols] * post_scale, 0, 1)
  double enc_scale, enc_offset, pad_with;
  double post_scale;   // -- the `_post_process` of the probability nets (reference nn/qdense.py:49-54, 443-448)
};

// the probability read-out of a lane's amplitudes: raw (T), or post-processed float64 (KScalars::post_cols)
template <typename T, int R, int LB>
__device__ __forceinline__ void store_probs(const KScalars& p, T* __restrict__ out, int64_t sample, int sub, const T (&pr)[R]) {
  if (p.post_cols > 0) {
    double* __restrict__ o = reinterpret_cast<double*>(out) + sample * p.out_ld;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int k = (r << LB) | sub;
      if (k < p.post_cols) o[k] = fmin(fmax((double)pr[r] * p.post_scale, 0.0), 1.0);
    }
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) out[sample * p.out_ld + ((r << LB) | sub)] = pr[r];
  }
}

// ---------------------------------------------------------------------------
// small math helpers
// ---------------------------------------------------------------------------
// sin/cos of a float angle with the range reduction done in double: < 1 ulp of float
// for |x| < 2^30, no scratch, ~30 instructions.
__device__ __forceinline__ void qsincos(float x, float* s, float* c) {
  const double xd = (double)x;
  const double kd = rint(xd * 0.63661977236758134308);  // 2/pi
  double r = fma(-kd, 1.57079632679489655800e+00, xd);
  r = fma(-kd, 6.12323399573676603587e-17, r);
  const double z = r * r;
  const double w = z * z;
  // fdlibm k_sinf / k_cosf minimax polynomials on [-pi/4, pi/4], evaluated in double
  const double sp = (r + (z * r) * (-0.166666666416265235595 + z * 0.0083333293858894631756)) +
                    (z * r) * w * (-0.000198393348360966317347 + z * 0.0000027183114939898219064);
  const double cp = ((1.0 + z * -0.499999997251031003120) + w * 0.0416666233237390631894) +
                    (w * z) * (-0.00138867637746099294692 + z * 0.0000243904487962774090654);
  const int q = (int)(long long)kd & 3;
  const double ss = (q & 1) ? cp : sp;
  const double cc = (q & 1) ? sp : cp;
  *s = (float)((q & 2) ? -ss : ss);
  *c = (float)(((q + 1) & 2) ? -cc : cc);
}
__device__ __forceinline__ void qsincos(double x, double* s, double* c) { sincos(x, s, c); }

// sin/cos in double, accurate to ~1e-9 (ample for a value that is then rounded to float):
// used to build the float gate tables from the float64 angles inside the launch.
__device__ __forceinline__ void sincos_for_f32_table(double x, double* s, double* c) {
  const double kd = rint(x * 0.63661977236758134308);
  double r = fma(-kd, 1.57079632679489655800e+00, x);
  r = fma(-kd, 6.12323399573676603587e-17, r);
  const double z = r * r, w = z * z;
  const double sp = (r + (z * r) * (-0.166666666416265235595 + z * 0.0083333293858894631756)) +
                    (z * r) * w * (-0.000198393348360966317347 + z * 0.0000027183114939898219064);
  const double cp = ((1.0 + z * -0.499999997251031003120) + w * 0.0416666233237390631894) +
                    (w * z) * (-0.00138867637746099294692 + z * 0.0000243904487962774090654);
  const int q = (int)(long long)kd & 3;
  const double ss = (q & 1) ? cp : sp, cc = (q & 1) ? sp : cp;
  *s = (q & 2) ? -ss : ss;
  *c = ((q + 1) & 2) ? -cc : cc;
}
// sin/cos of a float64 angle for the float32 engine's DATA angles (one per qubit, round and sample, on the critical path of
// a sampling step): range reduction in double (exact for any angle the nets produce), the fdlibm k_sinf / k_cosf
// polynomials in float -- ~1 ulp of float, a third of the cycles of the all-double evaluation above
__device__ __forceinline__ void data_sincos_f32(double x, float* s, float* c) {
  const double kd = rint(x * 0.63661977236758134308);
  double rd = fma(-kd, 1.57079632679489655800e+00, x);
  rd = fma(-kd, 6.12323399573676603587e-17, rd);
  const float r = (float)rd;
  const float z = r * r;
  const float sp = fmaf(r * z, fmaf(z, fmaf(z, fmaf(z, 2.7183114939898219064e-6f, -1.98393348360966317347e-4f),
                                          8.3333293858894631756e-3f), -1.66666666416265235595e-1f), r);
  const float cp = fmaf(z, fmaf(z, fmaf(z, fmaf(z, 2.43904487962774090654e-5f, -1.38867637746099294692e-3f),
                                       4.16666233237390631894e-2f), -4.99999997251031003120e-1f), 1.0f);
  const int q = (int)(long long)kd & 3;
  const float ss = (q & 1) ? cp : sp, cc = (q & 1) ? sp : cp;
  *s = (q & 2) ? -ss : ss;
  *c = ((q + 1) & 2) ? -cc : cc;
}
template <typename T>
__device__ __forceinline__ void table_sincos(double x, double* s, double* c) {
  if constexpr (sizeof(T) == 4) sincos_for_f32_table(x, s, c);
  else sincos(x, s, c);
}
__device__ __forceinline__ float qsqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double qsqrt(double x) { return sqrt(x); }

template <typename T>
__device__ __forceinline__ V2<T> bcast(T v) {
  return V2<T>{v, v};
}
// acc + a.x * u + a.y * ju  ==  acc + a * u  (complex) when ju = i*u
template <typename T>
__device__ __forceinline__ V2<T> cfma(V2<T> a, V2<T> u, V2<T> ju, V2<T> acc) {
  acc = __builtin_elementwise_fma(bcast<T>(a.x), u, acc);
  return __builtin_elementwise_fma(bcast<T>(a.y), ju, acc);
}
template <typename T>
__device__ __forceinline__ V2<T> cmul2(V2<T> a, V2<T> u, V2<T> ju) {
  return __builtin_elementwise_fma(bcast<T>(a.y), ju, bcast<T>(a.x) * u);
}
template <typename T>
__device__ __forceinline__ V2<T> times_i(V2<T> a) {
  return V2<T>{-a.y, a.x};
}

// ---------------------------------------------------------------------------
// cross-lane exchange, VALU only.  Amplitude indices use LOGICAL lane numbers
//     logical = physical ^ (3 * bit2(physical))        (an involution)
// so that flipping logical bit 0 / 1 / 2 is physical xor 1 / 2 / 7: all three are single DPP
// moves (quad_perm, quad_perm, row_half_mirror).  Bits 3..5 are the same in both numberings.
// xlane<MASK>(v) returns the value held by the lane whose LOGICAL number differs in bit MASK.
// ---------------------------------------------------------------------------
__device__ __forceinline__ constexpr int logical_lane(int physical) {
  return physical ^ (((physical >> 2) & 1) * 3);
}

template <int MASK>
__device__ __forceinline__ int xlane_i32(int v, int lane) {
  if constexpr (MASK == 1) {
    return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
  } else if constexpr (MASK == 2) {
    return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
  } else if constexpr (MASK == 4) {
    // LOGICAL bit 2: lanes are relabelled (logical_lane()) so that flipping it is the single
    // DPP row_half_mirror (physical lane ^ 7) instead of a two-move xor 4
    return __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true);
  } else if constexpr (MASK == 8) {
    return __builtin_amdgcn_mov_dpp(v, 0x128, 0xF, 0xF, true);  // row_ror:8
  } else if constexpr (MASK == 16) {
    // odd rows of the first operand <-> even rows of the second
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    return (lane & 16) ? (int)r[0] : (int)r[1];
  } else {
    static_assert(MASK == 32, "lane masks are single bits below 64");
    // upper half of the first operand <-> lower half of the second
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return (lane & 32) ? (int)r[0] : (int)r[1];
  }
}
template <int MASK>
__device__ __forceinline__ float xlane(float v, int lane) {
  return __int_as_float(xlane_i32<MASK>(__float_as_int(v), lane));
}
template <int MASK>
__device__ __forceinline__ double xlane(double v, int lane) {
  const int lo = xlane_i32<MASK>(__double2loint(v), lane);
  const int hi = xlane_i32<MASK>(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
template <int MASK, typename T>
__device__ __forceinline__ V2<T> xlane2(V2<T> a, int lane) {
  return V2<T>{xlane<MASK>(a.x, lane), xlane<MASK>(a.y, lane)};
}

// sum over the 2^LB lanes of a sample; every lane ends with the total
template <typename T, int LB>
__device__ __forceinline__ T group_sum(T v, int lane) {
  if constexpr (LB > 0) v += xlane<1>(v, lane);
  if constexpr (LB > 1) v += xlane<2>(v, lane);
  if constexpr (LB > 2) v += xlane<4>(v, lane);
  if constexpr (LB > 3) v += xlane<8>(v, lane);
  if constexpr (LB > 4) v += xlane<16>(v, lane);
  if constexpr (LB > 5) v += xlane<32>(v, lane);
  return v;
}

__device__ __forceinline__ float flip_sign(float v, uint32_t signbit) {
  return __int_as_float(__float_as_int(v) ^ (int)signbit);
}
__device__ __forceinline__ double flip_sign(double v, uint32_t signbit) {
  return __hiloint2double(__double2hiint(v) ^ (int)signbit, __double2loint(v));
}

// ---------------------------------------------------------------------------
// GF(2) index maps of the entangler rings
// ---------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ uint32_t cnot_ring_map(uint32_t k, int rr) {
  // CNOT(i, (i+rr) % N) for i = 0..N-1 in order; wire w <-> bit N-1-w.
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int t = (i + rr) % N;
    k ^= ((k >> (N - 1 - i)) & 1u) << (N - 1 - t);
  }
  return k;
}
template <int N>
__device__ __forceinline__ uint32_t cz_ring_parity(uint32_t k, int rr) {
  // sum_i b_i * b_{(i+rr) % N} mod 2  ==  parity(k & rotl_N(k, rr))
  const uint32_t rot = ((k << rr) | (k >> (N - rr))) & ((1u << N) - 1u);
  return __popc(k & rot) & 1u;
}

// ---------------------------------------------------------------------------
// LDS carve-up
// ---------------------------------------------------------------------------
template <typename T, int N, int LBMAX = 6>
struct Smem {
  using L = Layout<N, LBMAX>;
  static constexpr int kCz = L::NR * kWave;      // u32 [range][lane]
  static constexpr int kCnLane = L::NR * kWave;  // u32 [range][lane]
  static constexpr int kCnReg = L::NR * L::R;    // u32 [range][r]
  static constexpr size_t kTableBytes = ((size_t)(kCz + kCnLane + kCnReg) * 4 + 15) / 16 * 16;
  __host__ __device__ static size_t scratch_bytes(int waves) {
    return (size_t)waves * kWave * L::R * 2 * sizeof(T);
  }
  // the gate region holds either the general gate images (16 reals per Rot) or, for CZ circuits, the folded
  // per-layer tables (kFoldStride reals per layer): sized for the larger of the two
  static constexpr int kFoldStride = 2 * (N + L::LPS + L::R);
  static constexpr int kLayerReals = N * kLdsGateReals > kFoldStride ? N * kLdsGateReals : kFoldStride;
  __host__ __device__ static size_t gate_bytes(int64_t n_rot) { return (size_t)(n_rot / N) * kLayerReals * sizeof(T); }
  __host__ __device__ static size_t bytes(int64_t n_rot, bool cnot, int waves) {
    return gate_bytes(n_rot) + kTableBytes + (cnot ? scratch_bytes(waves) : 0);
  }
};

// ---------------------------------------------------------------------------
// Folded tables for CZ circuits.  Rot = RZ(omega) RY(theta) RZ(phi) and the CZ ring is diagonal, so everything
// between two RY layers is one diagonal that depends on the weights only:
//     RZ(phi^l) . [CZ ring^{l-1} . RZ(omega^{l-1})]        (the bracket only inside a round, li > 0)
// Per layer (stride 2 (n + LPS + R) reals):  ry[w] = (cos, sin)(theta_w / 2);
//     t_lo[k]  = exp(i sum_{q < LB} +-alpha_q / 2),  k = lane bits of the amplitude index;
//     t_hi[r]  = the same over the register bits;      alpha_w = phi^l_w + omega^{l-1}_w.
// The ring's sign stays in the parity bits.  The diagonal after the last RY layer of a round does not reach
// |amplitude|^2 and is dropped.  entry e of a layer: [0, n) ry, [n, n + LPS) t_lo, then t_hi.
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void folded_entry(const double* __restrict__ angles, int n, int lb, int layer, int li,
                                             int e, T* __restrict__ dst) {
  const int lps = 1 << lb;
  double c, sn;
  if (e < n) {
    table_sincos<T>(0.5 * angles[((size_t)layer * n + e) * 3 + 1], &sn, &c);
  } else {
    const bool lo = e < n + lps;
    const int k = lo ? e - n : e - n - lps;
    const int q0 = lo ? 0 : lb, q1 = lo ? lb : n;
    double ang = 0.0;
    for (int q = q0; q < q1; ++q) {
      const int w = n - 1 - q;
      double al = angles[((size_t)layer * n + w) * 3 + 0];
      if (li > 0) al += angles[((size_t)(layer - 1) * n + w) * 3 + 2];
      ang += ((k >> (q - q0)) & 1) ? 0.5 * al : -0.5 * al;
    }
    table_sincos<T>(ang, &sn, &c);
  }
  dst[0] = (T)c;
  dst[1] = (T)sn;
}

// entries (complex numbers, type T) of the per-layer tables appended to the gate table for n > 10:
// per layer  [0, n): (cos, sin)(theta_w / 2)     [n, 2n): (cos, sin)(alpha_w / 2),
// alpha_w = phi^l_w + omega^{l-1}_w inside a round (li > 0), phi^l_w for a round's first layer (never used: it acts
// on |0..0> and is a global phase).
template <typename T>
__device__ __forceinline__ void wide_fold_entry(const double* __restrict__ angles, int n, int layer, int li, int e,
                                                T* __restrict__ dst) {
  double c, s;
  if (e < n) {
    table_sincos<T>(0.5 * angles[((size_t)layer * n + e) * 3 + 1], &s, &c);
  } else {
    const int w = e - n;
    double al = angles[((size_t)layer * n + w) * 3 + 0];
    if (li > 0) al += angles[((size_t)(layer - 1) * n + w) * 3 + 2];
    table_sincos<T>(0.5 * al, &s, &c);
  }
  dst[0] = (T)c;
  dst[1] = (T)s;
}

// ---------------------------------------------------------------------------
// the engine: everything a wave needs to push its samples through the circuit
// ---------------------------------------------------------------------------
template <typename T, int N, int LBMAX = 6>
struct Engine {
  using L = Layout<N, LBMAX>;
  using S = Smem<T, N, LBMAX>;
  using C = V2<T>;
  static constexpr int LB = L::LB, R = L::R, LPS = L::LPS, SPW = L::SPW;

  const T* s_gates;
  const uint32_t* s_cz;
  const uint32_t* s_cn_lane;
  const uint32_t* s_cn_reg;
  C* s_slab;  // this wave's sample slot in the CNOT scratch
  int lane;   // physical lane (cross-lane moves, bits 4/5 selects)
  int llane;  // logical lane (amplitude indices, table rows, gate halves)
  int sub;    // logical lane within the sample

  struct Shift {
    int blk = -1, wire = 0;  // shifted input-angle occurrence (block, wire)
    T sign = 0;
  };

  // -- block-level staging (all threads of the block): carve, fill_*, then __syncthreads() ----
  T* s_gates_w;
  uint32_t* s_tables_w;

  __device__ __forceinline__ void carve(unsigned char* smem, int n_rot) {
    s_gates_w = reinterpret_cast<T*>(smem);
    s_tables_w = reinterpret_cast<uint32_t*>(smem + S::gate_bytes(n_rot));
    s_gates = s_gates_w;
    s_cz = s_tables_w;
    s_cn_lane = s_tables_w + S::kCz;
    s_cn_reg = s_tables_w + S::kCz + S::kCnLane;
    const int tid = threadIdx.x;
    lane = tid & (kWave - 1);
    llane = logical_lane(lane);
    sub = llane & (LPS - 1);
    s_slab = reinterpret_cast<C*>(smem + S::gate_bytes(n_rot) + S::kTableBytes) +
             (size_t)((tid >> 6) * SPW + (llane >> LB)) * L::D;
  }

  // LDS image of one gate: [u00, i*u00, u01, i*u01 | u11, i*u11, u10, i*u10]
  __device__ __forceinline__ static void put_gate(T* d, T u00r, T u00i, T u01r, T u01i, T u10r, T u10i,
                                                  T u11r, T u11i) {
    d[0] = u00r;  d[1] = u00i;  d[2] = -u00i;  d[3] = u00r;
    d[4] = u01r;  d[5] = u01i;  d[6] = -u01i;  d[7] = u01r;
    d[8] = u11r;  d[9] = u11i;  d[10] = -u11i; d[11] = u11r;
    d[12] = u10r; d[13] = u10i; d[14] = -u10i; d[15] = u10r;
  }

  __device__ __forceinline__ void fill_gates_from_table(const T* __restrict__ table, int n_rot,
                                                        int shift_gate, int shift_var) {
    for (int g = threadIdx.x; g < n_rot; g += blockDim.x) {
      const int var = (g == shift_gate) ? shift_var : 0;
      const T* u = table + ((size_t)g * kVariants + var) * kGateReals;
      put_gate(s_gates_w + (size_t)g * kLdsGateReals, u[0], u[1], u[2], u[3], u[4], u[5], u[6], u[7]);
    }
  }

  // Rot(phi, theta, omega) = RZ(omega) RY(theta) RZ(phi) straight from the (G, 3) float64 angles
  __device__ __forceinline__ void fill_gates_from_angles(const double* __restrict__ angles, int n_rot) {
    for (int g = threadIdx.x; g < n_rot; g += blockDim.x) {
      const double phi = angles[g * 3 + 0], theta = angles[g * 3 + 1], omega = angles[g * 3 + 2];
      double c, s, ca, sa, cb, sb;
      table_sincos<T>(0.5 * theta, &s, &c);
      table_sincos<T>(0.5 * (phi + omega), &sa, &ca);
      table_sincos<T>(0.5 * (phi - omega), &sb, &cb);
      put_gate(s_gates_w + (size_t)g * kLdsGateReals, (T)(ca * c), (T)(-sa * c), (T)(-cb * s),
               (T)(-sb * s), (T)(cb * s), (T)(-sb * s), (T)(ca * c), (T)(sa * c));
    }
  }

  __device__ __forceinline__ void fill_rings(bool use_cnot) {
    if constexpr (N > 1) {
      uint32_t* cz = s_tables_w;
      uint32_t* cn_lane = cz + S::kCz;
      uint32_t* cn_reg = cn_lane + S::kCnLane;
      const int tid = threadIdx.x;
      if (use_cnot) {
        for (int i = tid; i < L::NR * kWave; i += blockDim.x)
          cn_lane[i] = cnot_ring_map<N>((uint32_t)((i % kWave) & (LPS - 1)), i / kWave + 1);
        for (int i = tid; i < L::NR * R; i += blockDim.x)
          cn_reg[i] = cnot_ring_map<N>((uint32_t)(i % R) << LB, i / R + 1);
      } else {
        for (int i = tid; i < L::NR * kWave; i += blockDim.x) {
          const uint32_t ls = (uint32_t)((i % kWave) & (LPS - 1));
          uint32_t bits = 0;
#pragma unroll
          for (int r = 0; r < R; ++r)
            bits |= cz_ring_parity<N>(((uint32_t)r << LB) | ls, i / kWave + 1) << r;
          cz[i] = bits;
        }
      }
    }
  }

  // -- folded tables (CZ circuits): built in the gate region, stride S::kFoldStride per layer ------------
  __device__ __forceinline__ void fill_folded_from_angles(const double* __restrict__ angles, int n_layers_all,
                                                          int layers_per_round) {
    constexpr int E = N + LPS + R;
    for (int i = threadIdx.x; i < n_layers_all * E; i += blockDim.x) {
      const int l = i / E, e = i - l * E;
      folded_entry<T>(angles, N, LB, l, l % layers_per_round, e, s_gates_w + (size_t)l * S::kFoldStride + 2 * e);
    }
  }
  __device__ __forceinline__ void fill_folded_from_table(const T* __restrict__ tail, int n_layers_all) {
    for (int i = threadIdx.x; i < n_layers_all * S::kFoldStride; i += blockDim.x) s_gates_w[i] = tail[i];
  }
  // Tangent form for forward-only kernels (round 3; see qsim_lean.h): RY = c [[1, -t], [t, 1]] makes a gate ONE
  // multiply-add per amplitude component (fused with the DPP fetch where the partner is a lane) instead of multiply +
  // multiply-add, and the factors c of a layer commute with everything.  Done in place on the staged tables, after
  // fill_folded_*(): for every layer that is SIMULATED (li > 0: a round's first layer stays (cos, sin) for the product
  // state) ry[w] = (cos, sin) becomes (tan, cos) and the lane-part phases t_lo are multiplied by prod_w cos_w.  Returns
  // false -- tables untouched -- when some |tan| exceeds 16 (cos(theta/2) within 3.6 degrees of zero): the caller then
  // runs the (cos, sin) path.  Every thread of the block calls it; it ends with a barrier.
  __device__ __forceinline__ bool tangent_fold(int n_layers_all, int layers_per_round) {
    // (every wavefront looks at every entry and votes by ballot: the same answer in all of them without a word of static
    //  LDS -- __syncthreads_or() brings 256 B of it, and the deep float64 circuits ask for all 160 KiB as dynamic LDS)
    bool bad = false;
    for (int i = threadIdx.x & (kWave - 1); i < n_layers_all * N; i += kWave) {
      const int l = i / N, w = i - l * N;
      if (l % layers_per_round == 0) continue;
      const T* e = s_gates_w + (size_t)l * S::kFoldStride + 2 * w;
      const T c = e[0], sn = e[1];
      bad |= !(fabs(sn) <= (T)16 * fabs(c));
    }
    if (__ballot(bad) != 0) return false;
    __syncthreads();   // (all votes are cast before anyone rewrites the entries)
    for (int i = threadIdx.x; i < n_layers_all * LPS; i += blockDim.x) {
      const int l = i / LPS, k = i - l * LPS;
      if (l % layers_per_round == 0) continue;
      T* base = s_gates_w + (size_t)l * S::kFoldStride;
      T scale = 1;
#pragma unroll
      for (int w = 0; w < N; ++w) scale *= base[2 * w];
      base[2 * (N + k)] *= scale;
      base[2 * (N + k) + 1] *= scale;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_layers_all * N; i += blockDim.x) {
      const int l = i / N, w = i - l * N;
      if (l % layers_per_round == 0) continue;
      T* e = s_gates_w + (size_t)l * S::kFoldStride + 2 * w;
      const T c = e[0], sn = e[1];
      e[0] = sn / c;
      e[1] = c;
    }
    __syncthreads();
    return true;
  }
  struct FoldedLayer {
    C ry[N];
    C tlo;
    C thi[R];
    uint32_t cz;
  };
  __device__ __forceinline__ void load_folded(int layer, int prev_range_index, FoldedLayer& f) const {
    const C* base = reinterpret_cast<const C*>(s_gates + (size_t)layer * S::kFoldStride);
#pragma unroll
    for (int w = 0; w < N; ++w) f.ry[w] = base[w];
    f.tlo = base[N + sub];
    if constexpr (R > 1) {
#pragma unroll
      for (int r = 0; r < R; ++r) f.thi[r] = base[N + LPS + r];
    }
    f.cz = (N > 1 && prev_range_index >= 0) ? s_cz[prev_range_index * kWave + llane] : 0u;
  }
  __device__ __forceinline__ void load_folded_light(int layer, int prev_range_index, FoldedLayer& f) const {
    const C* base = reinterpret_cast<const C*>(s_gates + (size_t)layer * S::kFoldStride);
#pragma unroll
    for (int w = 0; w < N; ++w) f.ry[w] = base[w];
    f.tlo = base[N + sub];
    f.cz = (N > 1 && prev_range_index >= 0) ? s_cz[prev_range_index * kWave + llane] : 0u;
  }
  // RY(theta) on wire W with the coefficients (c, s): register pairs, swap trick or lane partner
  template <int W>
  __device__ __forceinline__ void ry_wires(C (&a)[R], const FoldedLayer& f) const {
    if constexpr (W < N) {
      constexpr int Q = N - 1 - W;
      const T c = f.ry[W].x, s = f.ry[W].y;
      if constexpr (kind_of<W>() == kReg) {
        ry_pairs<(1 << (Q >= LB ? Q - LB : 0))>(a, c, s);
      } else if constexpr (kind_of<W>() == kSwap) {
        swap_reg0_with_lane_bit<Q>(a);
        ry_pairs<1>(a, c, s);
        swap_reg0_with_lane_bit<Q>(a);
      } else {
        const T sg = ((llane >> Q) & 1) ? s : -s;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const C par = xlane2<(1 << Q), T>(a[r], lane);
          a[r] = __builtin_elementwise_fma(bcast<T>(sg), par, bcast<T>(c) * a[r]);
        }
      }
      ry_wires<W + 1>(a, f);
    }
  }
  template <int J>
  __device__ __forceinline__ void ry_pairs(C (&a)[R], T c, T s) const {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if ((r & J) == 0) {
        const C a0 = a[r], a1 = a[r | J];
        a[r] = __builtin_elementwise_fma(bcast<T>(-s), a1, bcast<T>(c) * a0);
        a[r | J] = __builtin_elementwise_fma(bcast<T>(s), a0, bcast<T>(c) * a1);
      }
    }
  }
  // all layers of one round on the folded tables; `first_layer` = index of the round's first layer
  // A round that starts from |0..0> (every encoding but the amplitude embedding): the first layer's diagonal -- data
  // angles included -- is a global phase and its RYs make a real product state, amplitude k = prod_q (bit q of k ? sin :
  // cos)(theta_q / 2).  Generated per lane (n multiplies, no cross-lane move) instead of simulated; `ry` = the layer's
  // (cos, sin) pairs by wire.
  __device__ __forceinline__ void product_state(const C* ry, C (&a)[R]) const {
    T fe = 1, fo = 1;
#pragma unroll
    for (int q = 0; q < LB; ++q) {
      const C cs = ry[N - 1 - q];
      const T f = ((sub >> q) & 1) ? cs.y : cs.x;
      if (q & 1) fo *= f;
      else fe *= f;
    }
    const T f = fe * fo;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      T g = f;
#pragma unroll
      for (int j = 0; j < N - LB; ++j) {
        const C cs = ry[N - 1 - (LB + j)];
        g *= ((r >> j) & 1) ? cs.y : cs.x;
      }
      a[r] = C{g, (T)0};
    }
  }
  __device__ __forceinline__ void folded_round(const KScalars& p, C (&a)[R], const C (&dx)[R], int first_layer,
                                               int n_layers_all) const {
    const int layers = p.n_blocks * p.sel_layers;
    FoldedLayer cur;
    load_folded(first_layer, -1, cur);
    int li0 = 0;
    if (p.encoding != 1) {   // the caller's |0..0> is replaced by the state behind the first layer
      product_state(cur.ry, a);
      load_folded(first_layer + 1 < n_layers_all ? first_layer + 1 : 0, N > 1 ? 0 : -1, cur);
      li0 = 1;
    }
    for (int li = li0; li < layers; ++li) {
      const int s = li % p.sel_layers;
      FoldedLayer nxt;
      {
        const int ln = first_layer + li + 1;
        load_folded(ln < n_layers_all ? ln : 0, N > 1 ? s % (N > 1 ? N - 1 : 1) : -1, nxt);
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        C ph = cur.tlo;
        if constexpr (R > 1) ph = cmul2<T>(cur.thi[r], ph, times_i<T>(ph));
        if (s == 0 && p.encoding == 2) ph = cmul2<T>(dx[r], ph, times_i<T>(ph));  // block start: data re-upload
        C v = cmul2<T>(ph, a[r], times_i<T>(a[r]));
        const uint32_t sb = ((cur.cz >> r) & 1u) << 31;  // CZ ring of the previous layer
        a[r] = C{flip_sign(v.x, sb), flip_sign(v.y, sb)};
      }
      ry_wires<0>(a, cur);
      __builtin_amdgcn_sched_barrier(0);
      cur = nxt;
    }
  }

  // the same round with the wave-uniform table entries (RY coefficients, register-bit phases) held in SCALAR registers
  // and no one-layer-ahead copy of the tables: ~100 fewer VGPRs at n = 10, which is what lets circuit_folded_kernel run
  // two waves per SIMD (the other wave hides the LDS latency the prefetch used to hide)
  __device__ __forceinline__ static T uni(T v) {
    if constexpr (sizeof(T) == 4) {
      return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
    } else {
      return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                              __builtin_amdgcn_readfirstlane(__double2loint(v)));
    }
  }
  // own += ts * partner for every amplitude of the lane, partner across lane bit Q < 4 (one DPP move inside the
  // multiply-add; see qsim_lean.h for the wait states -- here the R >= 1 pairs of a gate sit between the write and the
  // next gate's read of a register, one `s_nop 1` in front of each block covers the code before it)
  template <int CTRL, int CNT>
  __device__ __forceinline__ static void dpp_fmac_block(C* a, float ts) {
    static_assert(CNT == 1 || CNT == 2 || CNT == 4 || CNT == 8, "amplitudes per asm block");
#define QIDDM_DPP_FMAC(CTRLSTR)                                                                                        \
    if constexpr (CNT == 1) {                                                                                          \
      asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %2 " CTRLSTR "\n\tv_fmac_f32_dpp %1, %1, %2 " CTRLSTR "\n\ts_nop 0"   \
                   : "+v"(a[0].x), "+v"(a[0].y) : "v"(ts));                                                            \
    } else if constexpr (CNT == 2) {                                                                                   \
      asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %4 " CTRLSTR "\n\tv_fmac_f32_dpp %1, %1, %4 " CTRLSTR                \
                   "\n\tv_fmac_f32_dpp %2, %2, %4 " CTRLSTR "\n\tv_fmac_f32_dpp %3, %3, %4 " CTRLSTR                        \
                   : "+v"(a[0].x), "+v"(a[0].y), "+v"(a[1].x), "+v"(a[1].y) : "v"(ts));                                \
    } else if constexpr (CNT == 4) {                                                                                   \
      asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %8 " CTRLSTR "\n\tv_fmac_f32_dpp %1, %1, %8 " CTRLSTR                \
                   "\n\tv_fmac_f32_dpp %2, %2, %8 " CTRLSTR "\n\tv_fmac_f32_dpp %3, %3, %8 " CTRLSTR                        \
                   "\n\tv_fmac_f32_dpp %4, %4, %8 " CTRLSTR "\n\tv_fmac_f32_dpp %5, %5, %8 " CTRLSTR                        \
                   "\n\tv_fmac_f32_dpp %6, %6, %8 " CTRLSTR "\n\tv_fmac_f32_dpp %7, %7, %8 " CTRLSTR                        \
                   : "+v"(a[0].x), "+v"(a[0].y), "+v"(a[1].x), "+v"(a[1].y), "+v"(a[2].x), "+v"(a[2].y), "+v"(a[3].x), \
                     "+v"(a[3].y) : "v"(ts));                                                                          \
    } else {                                                                                                           \
      asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %16 " CTRLSTR "\n\tv_fmac_f32_dpp %1, %1, %16 " CTRLSTR              \
                   "\n\tv_fmac_f32_dpp %2, %2, %16 " CTRLSTR "\n\tv_fmac_f32_dpp %3, %3, %16 " CTRLSTR                      \
                   "\n\tv_fmac_f32_dpp %4, %4, %16 " CTRLSTR "\n\tv_fmac_f32_dpp %5, %5, %16 " CTRLSTR                      \
                   "\n\tv_fmac_f32_dpp %6, %6, %16 " CTRLSTR "\n\tv_fmac_f32_dpp %7, %7, %16 " CTRLSTR                      \
                   "\n\tv_fmac_f32_dpp %8, %8, %16 " CTRLSTR "\n\tv_fmac_f32_dpp %9, %9, %16 " CTRLSTR                      \
                   "\n\tv_fmac_f32_dpp %10, %10, %16 " CTRLSTR "\n\tv_fmac_f32_dpp %11, %11, %16 " CTRLSTR                  \
                   "\n\tv_fmac_f32_dpp %12, %12, %16 " CTRLSTR "\n\tv_fmac_f32_dpp %13, %13, %16 " CTRLSTR                  \
                   "\n\tv_fmac_f32_dpp %14, %14, %16 " CTRLSTR "\n\tv_fmac_f32_dpp %15, %15, %16 " CTRLSTR                  \
                   : "+v"(a[0].x), "+v"(a[0].y), "+v"(a[1].x), "+v"(a[1].y), "+v"(a[2].x), "+v"(a[2].y), "+v"(a[3].x), \
                     "+v"(a[3].y), "+v"(a[4].x), "+v"(a[4].y), "+v"(a[5].x), "+v"(a[5].y), "+v"(a[6].x), "+v"(a[6].y), \
                     "+v"(a[7].x), "+v"(a[7].y) : "v"(ts));                                                            \
    }
    if constexpr (CTRL == 0xB1) {
      QIDDM_DPP_FMAC("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1")
    } else if constexpr (CTRL == 0x4E) {
      QIDDM_DPP_FMAC("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1")
    } else if constexpr (CTRL == 0x141) {
      QIDDM_DPP_FMAC("row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1")
    } else {
      static_assert(CTRL == 0x128, "lane bits 0..3");
      QIDDM_DPP_FMAC("row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1")
    }
#undef QIDDM_DPP_FMAC
  }
  template <int Q>
  __device__ __forceinline__ void ry_lane_tan(C (&a)[R], T t) const {
    static_assert(Q >= 0 && Q < 4, "DPP lane bits");
    const T ts = ((llane >> Q) & 1) ? t : -t;
    if constexpr (sizeof(T) == 4) {
      constexpr int CTRL = Q == 0 ? 0xB1 : Q == 1 ? 0x4E : Q == 2 ? 0x141 : 0x128;
      constexpr int CNT = R >= 8 ? 8 : R;
#pragma unroll
      for (int r0 = 0; r0 < R; r0 += CNT) dpp_fmac_block<CTRL, CNT>(&a[r0], ts);
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const C par = xlane2<(1 << Q), T>(a[r], lane);
        a[r] = __builtin_elementwise_fma(bcast<T>(ts), par, a[r]);
      }
    }
  }
  template <int J>
  __device__ __forceinline__ static void ry_pairs_tan(C (&a)[R], T t) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if ((r & J) == 0) {
        const C a0 = a[r], a1 = a[r | J];
        a[r] = __builtin_elementwise_fma(bcast<T>(-t), a1, a0);
        a[r | J] = __builtin_elementwise_fma(bcast<T>(t), a0, a1);
      }
    }
  }
  template <int W>
  __device__ __forceinline__ void ry_wires_u_tan(C (&a)[R], const C* __restrict__ base) const {
    if constexpr (W < N) {
      constexpr int Q = N - 1 - W;
      const T t = uni(base[W].x);
      if constexpr (kind_of<W>() == kReg) {
        ry_pairs_tan<(1 << (Q >= LB ? Q - LB : 0))>(a, t);
      } else if constexpr (kind_of<W>() == kSwap) {
        swap_reg0_with_lane_bit<Q>(a);
        ry_pairs_tan<1>(a, t);
        swap_reg0_with_lane_bit<Q>(a);
      } else {
        ry_lane_tan<Q>(a, t);
      }
      ry_wires_u_tan<W + 1>(a, base);
    }
  }
  template <int W>
  __device__ __forceinline__ void ry_wires_u(C (&a)[R], const C* __restrict__ base) const {
    if constexpr (W < N) {
      constexpr int Q = N - 1 - W;
      const C csw = base[W];
      const T c = uni(csw.x), s = uni(csw.y);
      if constexpr (kind_of<W>() == kReg) {
        ry_pairs<(1 << (Q >= LB ? Q - LB : 0))>(a, c, s);
      } else if constexpr (kind_of<W>() == kSwap) {
        swap_reg0_with_lane_bit<Q>(a);
        ry_pairs<1>(a, c, s);
        swap_reg0_with_lane_bit<Q>(a);
      } else {
        const T sg = ((llane >> Q) & 1) ? s : -s;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const C par = xlane2<(1 << Q), T>(a[r], lane);
          a[r] = __builtin_elementwise_fma(bcast<T>(sg), par, bcast<T>(c) * a[r]);
        }
      }
      ry_wires_u<W + 1>(a, base);
    }
  }
  template <bool TAN = false>
  __device__ __forceinline__ void folded_round_u(const KScalars& p, C (&a)[R], const C (&dx)[R], int first_layer) const {
    const int layers = p.n_blocks * p.sel_layers;
    int li0 = 0;
    if (p.encoding != 1) {   // (see folded_round)
      const C* __restrict__ b0 = reinterpret_cast<const C*>(s_gates + (size_t)first_layer * S::kFoldStride);
      C ry0[N];
#pragma unroll
      for (int w = 0; w < N; ++w) ry0[w] = C{uni(b0[w].x), uni(b0[w].y)};
      product_state(ry0, a);
      li0 = 1;
    }
    for (int li = li0; li < layers; ++li) {
      const int s = li % p.sel_layers;
      const C* __restrict__ base = reinterpret_cast<const C*>(s_gates + (size_t)(first_layer + li) * S::kFoldStride);
      const C tlo = base[N + sub];
      // the CZ ring of the PREVIOUS layer of the round (none in front of a round's first layer)
      const int prev_ri = li == 0 ? -1 : ((li - 1) % p.sel_layers) % (N > 1 ? N - 1 : 1);
      const uint32_t cz = (N > 1 && prev_ri >= 0) ? s_cz[prev_ri * kWave + llane] : 0u;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        C ph = tlo;
        if constexpr (R > 1) {
          const C th = base[N + LPS + r];
          const C thu = C{uni(th.x), uni(th.y)};
          ph = cmul2<T>(thu, ph, times_i<T>(ph));
        }
        if (s == 0 && p.encoding == 2) ph = cmul2<T>(dx[r], ph, times_i<T>(ph));  // block start: data re-upload
        C v = cmul2<T>(ph, a[r], times_i<T>(a[r]));
        const uint32_t sb = ((cz >> r) & 1u) << 31;
        a[r] = C{flip_sign(v.x, sb), flip_sign(v.y, sb)};
      }
      if constexpr (TAN) ry_wires_u_tan<0>(a, base);
      else ry_wires_u<0>(a, base);
    }
  }

  // -- single-qubit gate between register pairs (r, r|J); matrix halves
  //    lo = [u00,iu00,u01,iu01] (row of the bit-clear amplitude), hi = [u11,iu11,u10,iu10]
  template <int J>
  __device__ __forceinline__ void gate_regs(C (&a)[R], const C* lo, const C* hi) const {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if ((r & J) == 0) {
        const C a0 = a[r], a1 = a[r | J];
        a[r] = cfma<T>(a1, lo[2], lo[3], cmul2<T>(a0, lo[0], lo[1]));
        a[r | J] = cfma<T>(a0, hi[2], hi[3], cmul2<T>(a1, hi[0], hi[1]));
      }
    }
  }
  // -- gate on lane bit Q; h = the half that matches this lane's bit: [own, i*own, partner, i*partner]
  template <int Q>
  __device__ __forceinline__ void gate_lane(C (&a)[R], const C* h) const {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const C own = a[r];
      const C par = xlane2<(1 << Q), T>(own, lane);
      a[r] = cfma<T>(par, h[2], h[3], cmul2<T>(own, h[0], h[1]));
    }
  }
  // -- exchange register bit 0 with lane bit Q (4: v_permlane16_swap, 5: v_permlane32_swap).
  //    Afterwards a[r] / a[r|1] hold the amplitudes whose OLD lane bit Q is 0 / 1, so a gate on
  //    that qubit is register-local with wave-uniform coefficients.  The exchange is an involution.
  template <int Q>
  __device__ __forceinline__ static void swap_dword(int& x, int& y) {
    if constexpr (Q == 5) {
      const auto r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)y, false, false);
      x = (int)r[0];
      y = (int)r[1];
    } else {
      const auto r = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)y, false, false);
      x = (int)r[0];
      y = (int)r[1];
    }
  }
  template <int Q>
  __device__ __forceinline__ static void swap_real(float& x, float& y) {
    int xi = __float_as_int(x), yi = __float_as_int(y);
    swap_dword<Q>(xi, yi);
    x = __int_as_float(xi);
    y = __int_as_float(yi);
  }
  template <int Q>
  __device__ __forceinline__ static void swap_real(double& x, double& y) {
    int xl = __double2loint(x), xh = __double2hiint(x), yl = __double2loint(y), yh = __double2hiint(y);
    swap_dword<Q>(xl, yl);
    swap_dword<Q>(xh, yh);
    x = __hiloint2double(xh, xl);
    y = __hiloint2double(yh, yl);
  }
  template <int Q>
  __device__ __forceinline__ void swap_reg0_with_lane_bit(C (&a)[R]) const {
#pragma unroll
    for (int r = 0; r < R; r += 2) {
      T x0 = a[r].x, y0 = a[r].y, x1 = a[r + 1].x, y1 = a[r + 1].y;
      swap_real<Q>(x0, x1);
      swap_real<Q>(y0, y1);
      a[r] = C{x0, y0};
      a[r + 1] = C{x1, y1};
    }
  }

  // how the gate on wire W (bit Q = N-1-W) is executed
  enum Kind { kReg, kSwap, kLane };
  template <int W>
  static constexpr Kind kind_of() {
    constexpr int Q = N - 1 - W;
    if (Q >= LB) return kReg;
    if ((Q == 4 || Q == 5) && R >= 2) return kSwap;
    return kLane;
  }

  // -- LDS -> registers: whole matrix (8 complex) or this lane's half (4 complex) ------------
  template <int W>
  __device__ __forceinline__ void load_gate(int gate0, C (&m)[8]) const {
    constexpr int Q = N - 1 - W;
    const C* gp = reinterpret_cast<const C*>(s_gates + (size_t)(gate0 + W) * kLdsGateReals);
    if constexpr (kind_of<W>() == kLane) {
      const C* hp = gp + (((llane >> Q) & 1) << 2);
#pragma unroll
      for (int i = 0; i < 4; ++i) m[i] = hp[i];
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) m[i] = gp[i];
    }
  }
  template <int W>
  __device__ __forceinline__ void apply_loaded(C (&a)[R], const C (&m)[8]) const {
    constexpr int Q = N - 1 - W;
    if constexpr (kind_of<W>() == kReg) {
      gate_regs<(1 << (Q >= LB ? Q - LB : 0))>(a, m, m + 4);
    } else if constexpr (kind_of<W>() == kSwap) {
      swap_reg0_with_lane_bit<Q>(a);
      gate_regs<1>(a, m, m + 4);
      swap_reg0_with_lane_bit<Q>(a);
    } else {
      gate_lane<Q>(a, m);
    }
  }

  // -- one Rot layer.  `cur` holds the (prefetched) matrix of wire 0; while gate W runs the
  //    matrix of gate W+1 is already in flight; on return `cur` holds wire 0 of `next_gate0`.
  template <int W>
  __device__ __forceinline__ void rot_steps(C (&a)[R], int gate0, int next_gate0, C (&cur)[8],
                                            C (&carry)[8]) const {
    C nxt[8];
    if constexpr (W + 1 < N) {
      load_gate<W + 1>(gate0, nxt);
    } else {
      load_gate<0>(next_gate0, nxt);
    }
    apply_loaded<W>(a, cur);
    // keep the one-ahead prefetch but stop the scheduler from hoisting further gates' LDS reads
    // above this point (it otherwise runs the kernel into the 256-VGPR wall and spills)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (W + 1 < N) {
      rot_steps<W + 1>(a, gate0, next_gate0, nxt, carry);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) carry[i] = nxt[i];
    }
  }
  __device__ __forceinline__ void rot_layer(C (&a)[R], int gate0, int next_gate0, C (&cur)[8]) const {
    C carry[8];
    rot_steps<0>(a, gate0, next_gate0, cur, carry);
#pragma unroll
    for (int i = 0; i < 8; ++i) cur[i] = carry[i];
  }

  // -- per-sample RY(x_w) layer (qml.AngleEmbedding rotation="Y") -----------------------
  template <int W>
  __device__ __forceinline__ void ry_layer(C (&a)[R], const T (&cs)[N], const T (&sn)[N]) const {
    if constexpr (W < N) {
      constexpr int Q = N - 1 - W;
      const T c = cs[W], s = sn[W], z = 0;
      if constexpr (Q >= LB) {
        const C m[8] = {C{c, z}, C{z, c}, C{-s, z}, C{z, -s}, C{c, z}, C{z, c}, C{s, z}, C{z, s}};
        gate_regs<(1 << (Q >= LB ? Q - LB : 0))>(a, m, m + 4);
      } else {
        const T sp = ((llane >> Q) & 1) ? s : -s;
        const C h[4] = {C{c, z}, C{z, c}, C{sp, z}, C{z, sp}};
        gate_lane<Q>(a, h);
      }
      ry_layer<W + 1>(a, cs, sn);
    }
  }

  // -- entangler ring of range index ri (range = ri + 1) --------------------------------
  __device__ __forceinline__ void ring(C (&a)[R], int ri, bool use_cnot) const {
    if constexpr (N > 1) {
      if (!use_cnot) {
        const uint32_t bits = s_cz[ri * kWave + llane];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const uint32_t sb = ((bits >> r) & 1u) << 31;
          a[r] = C{flip_sign(a[r].x, sb), flip_sign(a[r].y, sb)};
        }
      } else {
        const uint32_t lane_term = s_cn_lane[ri * kWave + llane];
#pragma unroll
        for (int r = 0; r < R; ++r) s_slab[lane_term ^ s_cn_reg[ri * R + r]] = a[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int r = 0; r < R; ++r) a[r] = s_slab[(r << LB) | sub];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
  }

  // -- cos/sin of every half input angle; all lanes of the sample end up holding all N ----
  __device__ __forceinline__ void half_angle_sincos(const T (&xs)[N], T (&cs)[N], T (&sn)[N]) const {
    if constexpr (LPS >= N) {
      // lane j of the sample evaluates wire j, then the N results are gathered
      // arithmetic select: a ?: chain over xs[] is turned into a scratch-indexed load by LLVM
      T mine = 0;
#pragma unroll
      for (int j = 0; j < N; ++j) mine = fma((T)(sub == j ? 1 : 0), xs[j], mine);
      T s, c;
      qsincos(mine * (T)0.5, &s, &c);
      const int base = llane & ~(LPS - 1);
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const int src = logical_lane(base | j);  // physical lane of logical lane (base | j)
        cs[j] = __shfl(c, src, kWave);
        sn[j] = __shfl(s, src, kWave);
      }
    } else {
#pragma unroll
      for (int j = 0; j < N; ++j) qsincos(xs[j] * (T)0.5, &sn[j], &cs[j]);
    }
  }

  // -- RZ-encoding diagonal prod_j exp(-+ i x_j / 2) for this lane's R amplitudes ----------
  __device__ __forceinline__ void rz_diagonal(const T (&cs)[N], const T (&sn)[N], C (&dx)[R]) const {
    T accr = 1, acci = 0;
#pragma unroll
    for (int q = 0; q < LB; ++q) {  // lane bits: wire N-1-q
      const T c = cs[N - 1 - q];
      const T si = ((llane >> q) & 1) ? sn[N - 1 - q] : -sn[N - 1 - q];
      const T nr = accr * c - acci * si;
      acci = accr * si + acci * c;
      accr = nr;
    }
    dx[0] = C{accr, acci};
#pragma unroll
    for (int j = 0; j < N - LB; ++j) {  // register bits: wire N-1-(LB+j)
      const T c = cs[N - 1 - (LB + j)], s = sn[N - 1 - (LB + j)];
#pragma unroll
      for (int r = 0; r < (1 << j); ++r) {
        const C d = dx[r];
        dx[r | (1 << j)] = C{d.x * c - d.y * s, d.x * s + d.y * c};  // bit set:   * (c + i s)
        dx[r] = C{d.x * c + d.y * s, d.y * c - d.x * s};             // bit clear: * (c - i s)
      }
    }
  }

  // -- all rounds of the circuit for this wave's sample(s) --------------------------------
  // xs: input angles (already scaled) for RZ/RY encodings; amp_row: feature row for
  // amplitude embedding.  On return pr[] holds the probabilities of this lane's
  // amplitudes and, for the <Z> read-out, result[] the n expectation values.
  // folded: the gate region holds the folded tables (CZ circuits, no RY data encoding, no parameter shift)
  template <typename Src>
  __device__ __forceinline__ void run(const KScalars& p, const Src& amp_src, T (&xs)[N],
                                      const Shift& sh, T (&result)[N], T (&pr)[R], bool folded = false) const {
    const bool use_cnot = p.imprimitive == 0;
    C a[R];
    C dx[R];
    T cs[N], sn[N];
    const int n_rot = p.n_rounds * p.n_blocks * p.sel_layers * N;
    C gate_m[8];  // matrix of the next wire-0 gate, prefetched one layer ahead
    if (!folded) load_gate<0>(0, gate_m);
    for (int round = 0; round < p.n_rounds; ++round) {
      // ---- state preparation ---------------------------------------------------------
      if (p.encoding == 1) {
        T n2 = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int k = (r << LB) | sub;
          T v = (T)p.pad_with;
          if (k < p.n_features) v = amp_src(k) + (T)p.enc_offset;
          a[r] = C{v, (T)0};
          n2 += v * v;
        }
        n2 = group_sum<T, LB>(n2, lane);
        const T inv = (T)1 / qsqrt(n2);
#pragma unroll
        for (int r = 0; r < R; ++r) a[r] = C{a[r].x * inv, (T)0};
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r) a[r] = C{(T)0, (T)0};
        a[0] = C{sub == 0 ? (T)1 : (T)0, (T)0};
      }
      if (p.encoding >= 2) half_angle_sincos(xs, cs, sn);
      if (p.encoding == 2) rz_diagonal(cs, sn, dx);

      // ---- blocks -----------------------------------------------------------------------
      if (folded) {
        folded_round(p, a, dx, round * p.n_blocks * p.sel_layers, n_rot / N);
      } else
      for (int blk = 0; blk < p.n_blocks; ++blk) {
        if (p.encoding == 2) {
#pragma unroll
          for (int r = 0; r < R; ++r) a[r] = cmul2<T>(dx[r], a[r], times_i<T>(a[r]));
          if (blk == sh.blk) {  // parameter shift: extra RZ(+-pi/2) on sh.wire
            const int q = N - 1 - sh.wire;
            const T h = (T)0.70710678118654752440;
#pragma unroll
            for (int r = 0; r < R; ++r) {
              const int k = (r << LB) | sub;
              const T si = ((k >> q) & 1) ? h * sh.sign : -h * sh.sign;
              a[r] = C{a[r].x * h - a[r].y * si, a[r].x * si + a[r].y * h};
            }
          }
        } else if ((p.encoding == 3 && blk == 0) || p.encoding == 4) {
          T cb[N], sb[N];
#pragma unroll
          for (int j = 0; j < N; ++j) {
            cb[j] = cs[j];
            sb[j] = sn[j];
          }
          if (sh.blk == blk) {  // parameter shift of one RY input angle: rotate (c, s) by +-pi/4
            const T h = (T)0.70710678118654752440;
#pragma unroll
            for (int j = 0; j < N; ++j) {
              if (j == sh.wire) {
                cb[j] = h * (cs[j] - sh.sign * sn[j]);
                sb[j] = h * (sn[j] + sh.sign * cs[j]);
              }
            }
          }
          ry_layer<0>(a, cb, sb);
        }
        for (int s = 0; s < p.sel_layers; ++s) {
          const int gate0 = ((round * p.n_blocks + blk) * p.sel_layers + s) * N;
          const int next0 = gate0 + N < n_rot ? gate0 + N : 0;  // last layer: any valid gate
          rot_layer(a, gate0, next0, gate_m);
          if constexpr (N > 1) ring(a, s % (N - 1), use_cnot);
        }
      }

      // ---- measurement -------------------------------------------------------------------
#pragma unroll
      for (int r = 0; r < R; ++r) pr[r] = a[r].x * a[r].x + a[r].y * a[r].y;
      if (p.measure == 1) {
#pragma unroll
        for (int w = 0; w < N; ++w) {
          const int q = N - 1 - w;
          T acc = 0;
#pragma unroll
          for (int r = 0; r < R; ++r) acc += ((((r << LB) | sub) >> q) & 1) ? -pr[r] : pr[r];
          result[w] = group_sum<T, LB>(acc, lane);
        }
      }
      // ---- chain into the next round: x <- out[:, 0:N] ------------------------------------
      if (round + 1 < p.n_rounds) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
          const T v = (p.measure == 1)
                          ? result[j]
                          : __shfl(pr[0], logical_lane((llane & ~(LPS - 1)) | (j & (LPS - 1))), kWave);
          xs[j] = v * (T)p.enc_scale;
        }
      }
    }
  }

  // -- the same for CZ circuits with no / RZ encoding on the folded tables ONLY: nothing of the general gate path, the
  //    CNOT scatter or the amplitude embedding is compiled in, which is what lets the n = 9 / 10 forward fit two
  //    waves per SIMD (circuit_folded_kernel) ---------------------------------------------------------------------
  template <bool TAN = false>
  __device__ __forceinline__ void run_folded(const KScalars& p, T (&xs)[N], T (&result)[N], T (&pr)[R]) const {
    C a[R];
    C dx[R];
    T cs[N], sn[N];
    for (int round = 0; round < p.n_rounds; ++round) {
#pragma unroll
      for (int r = 0; r < R; ++r) a[r] = C{(T)0, (T)0};
      a[0] = C{sub == 0 ? (T)1 : (T)0, (T)0};
      if (p.encoding == 2) {
        half_angle_sincos(xs, cs, sn);
        rz_diagonal(cs, sn, dx);
      }
      folded_round_u<TAN>(p, a, dx, round * p.n_blocks * p.sel_layers);
#pragma unroll
      for (int r = 0; r < R; ++r) pr[r] = a[r].x * a[r].x + a[r].y * a[r].y;
      if (p.measure == 1) {
#pragma unroll
        for (int w = 0; w < N; ++w) {
          const int q = N - 1 - w;
          T acc = 0;
#pragma unroll
          for (int r = 0; r < R; ++r) acc += ((((r << LB) | sub) >> q) & 1) ? -pr[r] : pr[r];
          result[w] = group_sum<T, LB>(acc, lane);
        }
      }
      if (round + 1 < p.n_rounds) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
          const T v = (p.measure == 1)
                          ? result[j]
                          : __shfl(pr[0], logical_lane((llane & ~(LPS - 1)) | (j & (LPS - 1))), kWave);
          xs[j] = v * (T)p.enc_scale;
        }
      }
    }
  }
};

// CZ entanglers and no RY data encoding: the circuit can run on the folded tables
__host__ __device__ inline bool can_fold(int imprimitive, int encoding) { return imprimitive == 1 && encoding <= 2; }

// feature sources for the amplitude embedding
template <typename T>
struct RowSrc {  // a row of a (batch, features) matrix
  const T* __restrict__ row;
  __device__ __forceinline__ T operator()(int k) const { return row[k]; }
};
struct NoSrc {
  __device__ __forceinline__ float operator()(int) const { return 0.f; }
};
// torch.nn.Unfold on the fly: feature f = (c * kh + di) * kw + dj of the patch around output pixel
// (oi, oj), zero padding outside the image (reference nn/qconv.py:23, 76)
template <typename T>
struct PatchSrc {
  const double* __restrict__ img;  // (C, H, W) of this sample
  int H, W, kh, kw, oi, oj;        // oi/oj already shifted by -padding
  __device__ __forceinline__ T operator()(int f) const {
    const int dj = f % kw, t = f / kw;
    const int di = t % kh, c = t / kh;
    const int i = oi + di, j = oj + dj;
    if (i < 0 || i >= H || j < 0 || j >= W) return (T)0;
    return (T)img[((size_t)c * H + i) * W + j];
  }
};

// ---------------------------------------------------------------------------
// circuit kernel: inputs -> probabilities / <Z>   (SHIFT: -> dot with upstream grad)
// ---------------------------------------------------------------------------
template <typename T, int N, bool SHIFT>
__global__ __launch_bounds__(4 * kWave) void circuit_kernel(const T* __restrict__ inputs,
                                                         const T* __restrict__ table,
                                                         T* __restrict__ out,
                                                         const T* __restrict__ gout,
                                                         T* __restrict__ dots, const KScalars p) {
  using E = Engine<T, N>;
  using L = typename E::L;
  constexpr int LB = L::LB, R = L::R, SPW = L::SPW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

  const int n_rot = p.n_rounds * p.n_blocks * p.sel_layers * N;
  int shift_gate = -1, shift_var = 0, replica_local = 0;
  typename E::Shift sh;
  if constexpr (SHIFT) {
    replica_local = blockIdx.y;
    const int rho = p.first_replica + replica_local;
    if (rho < 6 * n_rot) {
      shift_gate = rho / 6;
      shift_var = 1 + rho % 6;
    } else {
      const int q = rho - 6 * n_rot;
      sh.blk = q / (2 * N);
      sh.wire = (q >> 1) % N;
      sh.sign = (q & 1) ? (T)-1 : (T)1;
    }
  }
  E eng;
  eng.carve(smem_raw, n_rot);
  // forward of a CZ circuit: the folded tables that qiddm_prepare_gates appended to the gate table
  const bool folded = !SHIFT && p.fold != 0;
  if (folded)
    eng.fill_folded_from_table(table + (size_t)n_rot * kVariants * kGateReals, n_rot / N);
  else
    eng.fill_gates_from_table(table, n_rot, shift_gate, shift_var);
  eng.fill_rings(p.imprimitive == 0);
  __syncthreads();
  const int lane = eng.lane, sub = eng.sub;
  const int wave = threadIdx.x >> 6;
  const int swave = eng.llane >> LB;

  const int64_t groups = (p.batch + SPW - 1) / SPW;
  const int waves_per_block = blockDim.x >> 6;
  for (int64_t grp = (int64_t)blockIdx.x * waves_per_block + wave; grp < groups;
       grp += (int64_t)gridDim.x * waves_per_block) {
    const int64_t sample_raw = grp * SPW + swave;
    const bool valid = sample_raw < p.batch;
    const int64_t sample = valid ? sample_raw : p.batch - 1;

    T xs[N];
    if (p.encoding >= 2) {
#pragma unroll
      for (int j = 0; j < N; ++j) xs[j] = inputs[sample * p.in_ld + j] * (T)p.enc_scale;
    } else {
#pragma unroll
      for (int j = 0; j < N; ++j) xs[j] = (T)0;
    }
    T result[N], pr[R];
    eng.run(p, RowSrc<T>{inputs + sample * p.in_ld}, xs, sh, result, pr, folded);

    if constexpr (!SHIFT) {
      if (p.measure == 0) {
        if (valid) store_probs<T, R, LB>(p, out, sample, sub, pr);
      } else {
        T v = 0;  // arithmetic select (see half_angle_sincos)
#pragma unroll
        for (int w = 0; w < N; ++w) v = fma((T)(sub == w ? 1 : 0), result[w], v);
        if (valid && sub < N) out[sample * p.out_ld + sub] = v;
      }
    } else {
      T acc = 0;
      if (p.measure == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) acc += gout[sample * p.g_ld + ((r << LB) | sub)] * pr[r];
        acc = group_sum<T, LB>(acc, lane);
      } else {
#pragma unroll
        for (int w = 0; w < N; ++w) acc += gout[sample * p.g_ld + w] * result[w];
      }
      if (valid && sub == 0) dots[(int64_t)replica_local * p.batch + sample] = acc;
    }
  }
}

// ---------------------------------------------------------------------------
// circuit kernel for CZ circuits with no / RZ encoding, forward only, on the folded tables: the lean sibling of
// circuit_kernel<T, N, false> for the wide register-resident sizes (n = 9, 10).  Round-2 counters
// (profiles/r02a/wide_pmc_sq.json): the all-paths kernel needs 256 VGPRs + 205 AGPRs at n = 10 -> ONE wave per SIMD,
// where it sits at 86 % of the single-wave issue cap (one VALU instruction per 4 cycles) = 35 % of the vector peak.
// ---------------------------------------------------------------------------
template <typename T, int N>
__global__ __launch_bounds__(4 * kWave, (sizeof(T) == 4 ? 2 : 1)) void circuit_folded_kernel(const T* __restrict__ inputs,
                                                                      const T* __restrict__ table,
                                                                      T* __restrict__ out, const KScalars p) {
  using E = Engine<T, N>;
  using L = typename E::L;
  constexpr int LB = L::LB, R = L::R, SPW = L::SPW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int n_rot = p.n_rounds * p.n_blocks * p.sel_layers * N;
  E eng;
  eng.carve(smem_raw, n_rot);
  eng.fill_folded_from_table(table + (size_t)n_rot * kVariants * kGateReals, n_rot / N);
  eng.fill_rings(false);
  __syncthreads();
  // tangent-form layers where the weights allow it (a launch-uniform choice; round 3)
  const bool tan_ok = eng.tangent_fold(n_rot / N, p.n_blocks * p.sel_layers);
  const int sub = eng.sub;
  const int wave = threadIdx.x >> 6;
  const int swave = eng.llane >> LB;
  const int64_t groups = (p.batch + SPW - 1) / SPW;
  const int waves_per_block = blockDim.x >> 6;
  for (int64_t grp = (int64_t)blockIdx.x * waves_per_block + wave; grp < groups;
       grp += (int64_t)gridDim.x * waves_per_block) {
    const int64_t sample_raw = grp * SPW + swave;
    const bool valid = sample_raw < p.batch;
    const int64_t sample = valid ? sample_raw : p.batch - 1;
    T xs[N];
    if (p.encoding >= 2) {
#pragma unroll
      for (int j = 0; j < N; ++j) xs[j] = inputs[sample * p.in_ld + j] * (T)p.enc_scale;
    } else {
#pragma unroll
      for (int j = 0; j < N; ++j) xs[j] = (T)0;
    }
    T result[N], pr[R];
    if (tan_ok) eng.template run_folded<true>(p, xs, result, pr);
    else eng.template run_folded<false>(p, xs, result, pr);
    if (p.measure == 0) {
      if (valid) store_probs<T, R, LB>(p, out, sample, sub, pr);
    } else {
      T v = 0;  // arithmetic select (see half_angle_sincos)
#pragma unroll
      for (int w = 0; w < N; ++w) v = fma((T)(sub == w ? 1 : 0), result[w], v);
      if (valid && sub < N) out[sample * p.out_ld + sub] = v;
    }
  }
}

// ---------------------------------------------------------------------------
// dense-net kernel: linear_down -> N chained circuit rounds -> linear_up in ONE launch.
// The whole forward of the reference's QNN_noise / QNN / QIDDM_LL_noise
// (nn/qdense.py:267-289, 346-368, 1620-1642) and, with post_mode 1, the "noise"-goal
// update of the sampling loop (src/models.py:130-134) on top of it.  Classical glue in
// float64 (the reference's dtype), statevector in T.  Gate matrices are built from the
// raw float64 angles while the workgroup stages its LDS tables: no separate launch.
// ---------------------------------------------------------------------------
struct DenseScalars {
  int64_t x_ld, y_ld;
  int32_t in_features, out_features;
  int32_t post_mode;  // 0: y = net(x)   1: y = clamp(x - (net(x) - 0.5) * 0.1 * noise_factor, 0, 1)
  int32_t pad_;
  double noise_factor;
  unsigned long long* stamps;  // diagnostics only (tools/stamp_dense.py): s_memtime at phase ends
};

template <typename T, int N, bool LDSW, int WPB, int U = (WPB >= 4 ? 4 : 7)>
__global__ __launch_bounds__(WPB * kWave) void dense_forward_kernel(
    const double* __restrict__ x, const double* __restrict__ wd, const double* __restrict__ bd,
    const double* __restrict__ angles, const double* __restrict__ wu, const double* __restrict__ bu,
    double* __restrict__ y, const DenseScalars d, const KScalars p) {
  using E = Engine<T, N>;
  using L = typename E::L;
  constexpr int LB = L::LB, R = L::R, SPW = L::SPW, LPS = L::LPS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int n_rot = p.n_rounds * p.n_blocks * p.sel_layers * N;
  const int P = d.in_features, Q = d.out_features;
  const bool stamp = d.stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
  if (stamp) d.stamps[0] = __builtin_amdgcn_s_memtime();
  E eng;
  eng.carve(smem_raw, n_rot);
  // LDSW: both weight matrices live in LDS as [j][pixel] float64 (w_up transposed while staging),
  // so the per-sample GEMV loops read conflict-free consecutive ds_read_b64 instead of L2.
  double* s_wd = reinterpret_cast<double*>(smem_raw + Smem<T, N>::bytes(n_rot, p.imprimitive == 0, blockDim.x >> 6));
  double* s_wu = s_wd + (size_t)N * P;
  if constexpr (LDSW) {
    for (int i = threadIdx.x; i < N * P; i += blockDim.x) s_wd[i] = wd[i];
    for (int i = threadIdx.x; i < N * Q; i += blockDim.x) s_wu[(size_t)(i % N) * Q + i / N] = wu[i];
  }
  const bool folded = p.fold != 0;
  if (folded)
    eng.fill_folded_from_angles(angles, n_rot / N, p.n_blocks * p.sel_layers);
  else
    eng.fill_gates_from_angles(angles, n_rot);
  eng.fill_rings(p.imprimitive == 0);
  __syncthreads();
  if (stamp) d.stamps[1] = __builtin_amdgcn_s_memtime();
  const int lane = eng.lane, sub = eng.sub;
  const int wave = threadIdx.x >> 6;
  const int swave = eng.llane >> LB;
  const typename E::Shift no_shift;

  const int64_t groups = (p.batch + SPW - 1) / SPW;
  const int waves_per_block = blockDim.x >> 6;
  for (int64_t grp = (int64_t)blockIdx.x * waves_per_block + wave; grp < groups;
       grp += (int64_t)gridDim.x * waves_per_block) {
    const int64_t sample_raw = grp * SPW + swave;
    const bool valid = sample_raw < p.batch;
    const int64_t sample = valid ? sample_raw : p.batch - 1;
    const double* __restrict__ xrow = x + sample * d.x_ld;

    // ---- linear_down: h_j = b_j + sum_p x_p W[j, p], lanes stride over the pixels ----------
    double acc[N];
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] = 0.0;
    for (int p0 = sub; p0 < P; p0 += LPS * U) {
      double xv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int pix = p0 + u * LPS;
        xv[u] = pix < P ? xrow[pix] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int pix = p0 + u * LPS;
        const int pc = pix < P ? pix : 0;  // xv is 0 there
#pragma unroll
        for (int j = 0; j < N; ++j) {
          const double wv = LDSW ? s_wd[(size_t)j * P + pc] : wd[(size_t)j * P + pc];
          acc[j] = fma(xv[u], wv, acc[j]);
        }
      }
    }
    T xs[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const double h = group_sum<double, LB>(acc[j], lane) + (bd ? bd[j] : 0.0);
      xs[j] = (T)(h * p.enc_scale);
    }

    if (stamp) d.stamps[2] = __builtin_amdgcn_s_memtime();
    // ---- the circuit --------------------------------------------------------------------------
    T result[N], pr[R];
    eng.run(p, NoSrc{}, xs, no_shift, result, pr, folded);
    double ev[N];
#pragma unroll
    for (int j = 0; j < N; ++j) ev[j] = (double)result[j];
    if (stamp) d.stamps[3] = __builtin_amdgcn_s_memtime();

    // ---- linear_up (+ optional sampling update) --------------------------------------------
    double* __restrict__ yrow = y + sample * d.y_ld;
    for (int p0 = sub; p0 < Q; p0 += LPS * U) {
      double o[U], xin[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int pix = p0 + u * LPS;
        const int pc = pix < Q ? pix : 0;
        o[u] = bu ? bu[pc] : 0.0;
        xin[u] = d.post_mode == 1 ? xrow[pc] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int pix = p0 + u * LPS;
        const int pc = pix < Q ? pix : 0;
#pragma unroll
        for (int j = 0; j < N; ++j) {
          const double wv = LDSW ? s_wu[(size_t)j * Q + pc] : wu[(size_t)pc * N + j];
          o[u] = fma(ev[j], wv, o[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int pix = p0 + u * LPS;
        double v = o[u];
        if (d.post_mode == 1) v = fmin(fmax(xin[u] - (v - 0.5) * 0.1 * d.noise_factor, 0.0), 1.0);
        if (valid && pix < Q) yrow[pix] = v;
      }
    }
    if (stamp) d.stamps[4] = __builtin_amdgcn_s_memtime();
  }
}

// ---------------------------------------------------------------------------
// quantum convolution in one launch: unfold (+0.1) -> AmplitudeEmbedding(pad 0.5) -> SEL(CNOT) ->
// probs -> * D/2, clamp, [::2], [:C_out] -> (B, C_out, H_out, W_out)   (the intended
// _QConv2d_FAST.forward, reference nn/qconv.py:51-87, finding F3).  One circuit per output pixel,
// one wavefront (or a slice of one) per circuit; neither the unfolded patches nor the
// probability rows ever exist in HBM.
// ---------------------------------------------------------------------------
struct ConvScalars {
  int32_t C, H, W, kh, kw, ph, pw, Ho, Wo, C_out;
  double post_scale;  // D / 2
};

template <typename T, int N>
__global__ __launch_bounds__(4 * kWave) void qconv_forward_kernel(const double* __restrict__ x,
                                                                  const double* __restrict__ angles,
                                                                  double* __restrict__ y,
                                                                  const ConvScalars cv, const KScalars p) {
  using E = Engine<T, N>;
  using L = typename E::L;
  constexpr int LB = L::LB, R = L::R, SPW = L::SPW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int n_rot = p.n_rounds * p.n_blocks * p.sel_layers * N;
  E eng;
  eng.carve(smem_raw, n_rot);
  eng.fill_gates_from_angles(angles, n_rot);
  eng.fill_rings(p.imprimitive == 0);
  __syncthreads();
  const int sub = eng.sub;
  const int wave = threadIdx.x >> 6;
  const int swave = eng.llane >> LB;
  const typename E::Shift no_shift;
  const int64_t pixels = (int64_t)cv.Ho * cv.Wo;
  const int waves_per_block = blockDim.x >> 6;
  const int64_t groups = (p.batch + SPW - 1) / SPW;
  for (int64_t grp = (int64_t)blockIdx.x * waves_per_block + wave; grp < groups;
       grp += (int64_t)gridDim.x * waves_per_block) {
    const int64_t m_raw = grp * SPW + swave;
    const bool valid = m_raw < p.batch;
    const int64_t m = valid ? m_raw : p.batch - 1;
    const int64_t b = m / pixels;
    const int pix = (int)(m - b * pixels);
    const int oi = pix / cv.Wo, oj = pix - oi * cv.Wo;
    const PatchSrc<T> src{x + (size_t)b * cv.C * cv.H * cv.W, cv.H, cv.W, cv.kh, cv.kw, oi - cv.ph, oj - cv.pw};
    T xs[N], result[N], pr[R];
#pragma unroll
    for (int j = 0; j < N; ++j) xs[j] = (T)0;
    eng.run(p, src, xs, no_shift, result, pr);
    if (valid) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int k = (r << LB) | sub;
        const int co = k >> 1;
        if ((k & 1) == 0 && co < cv.C_out) {
          double v = (double)pr[r] * cv.post_scale;
          v = fmin(fmax(v, 0.0), 1.0);
          y[(((size_t)b * cv.C_out + co) * cv.Ho + oi) * cv.Wo + oj] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// gate-table preparation: angles (G,3) f64 -> (G, 7, 8) T
// ---------------------------------------------------------------------------
template <typename T>
__global__ void prepare_gates_kernel(const double* __restrict__ angles, T* __restrict__ table,
                                     int64_t n_rot, int n, int layers_per_round, int64_t fold_entries) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rot * kVariants) {
    // the folded tables of a register-resident circuit (n <= 10), appended after the gate variants
    const int64_t j = i - n_rot * kVariants;
    if (j >= fold_entries) return;
    if (n > 10) {  // per-layer tables of the wide CZ kernel (qsim_wide_cz.h): 2n entries per layer
      const int layer = (int)(j / (2 * n)), e = (int)(j - (int64_t)layer * 2 * n);
      wide_fold_entry<T>(angles, n, layer, layer % layers_per_round, e, table + n_rot * kVariants * kGateReals + 2 * j);
      return;
    }
    const int lb = n < 6 ? n : 6;
    const int e_per_layer = n + (1 << lb) + (1 << (n - lb));
    const int64_t n_layers = n_rot / n;
    if (j >= n_layers * e_per_layer) {  // n = 10 CZ circuits: the per-wire tables of cz10_adjoint_kernel behind them
      const int64_t j2 = j - n_layers * e_per_layer;
      const int layer2 = (int)(j2 / (2 * n)), e2 = (int)(j2 - (int64_t)layer2 * 2 * n);
      wide_fold_entry<T>(angles, n, layer2, layer2 % layers_per_round, e2, table + n_rot * kVariants * kGateReals + 2 * j);
      return;
    }
    const int layer = (int)(j / e_per_layer), e = (int)(j - (int64_t)layer * e_per_layer);
    folded_entry<T>(angles, n, lb, layer, layer % layers_per_round, e,
                    table + n_rot * kVariants * kGateReals + (size_t)layer * 2 * e_per_layer + 2 * e);
    return;
  }
  const int64_t g = i / kVariants;
  const int v = (int)(i % kVariants);
  double ang[3] = {angles[g * 3 + 0], angles[g * 3 + 1], angles[g * 3 + 2]};
  if (v > 0) ang[(v - 1) >> 1] += ((v - 1) & 1) ? -1.57079632679489661923 : 1.57079632679489661923;
  const double phi = ang[0], theta = ang[1], omega = ang[2];
  double c, s, ca, sa, cb, sb;
  sincos(0.5 * theta, &s, &c);
  sincos(0.5 * (phi + omega), &sa, &ca);
  sincos(0.5 * (phi - omega), &sb, &cb);
  T* u = table + i * kGateReals;
  u[0] = (T)(ca * c);   u[1] = (T)(-sa * c);  // U00 =  e^{-i(phi+omega)/2} c
  u[2] = (T)(-cb * s);  u[3] = (T)(-sb * s);  // U01 = -e^{+i(phi-omega)/2} s
  u[4] = (T)(cb * s);   u[5] = (T)(-sb * s);  // U10 =  e^{-i(phi-omega)/2} s
  u[6] = (T)(ca * c);   u[7] = (T)(sa * c);   // U11 =  e^{+i(phi+omega)/2} c
}

}  // namespace qiddm
