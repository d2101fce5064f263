// qsim_fused.h -- fused whole-circuit statevector kernels for gfx950 (CDNA4).
//
// One wavefront (64 lanes) owns the complete 2^n-amplitude slab of a sample in
// registers for the whole circuit (n <= 10): amplitude index k = (r << LB) | sub
// with sub = lane bits (LB = min(n, 6)) and r = register index (R = 2^(n-LB)
// complex amplitudes per lane).  For n < 6 a wave carries 64 / 2^n samples.
// Wire w is bit q = n-1-w of k.  A gate on a register bit is pure VALU work; a
// gate on a lane bit pairs lane l with l ^ 2^q through cross-lane moves (DPP /
// ds_swizzle / permlane -- never LDS memory).  CZ rings are precomputed sign
// masks, CNOT rings a GF(2)-linear scatter through a per-wave LDS scratch slab.
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
constexpr int kWavesPerBlock = 4;
constexpr int kBlock = kWave * kWavesPerBlock;
constexpr int kVariants = 7;   // base + six parameter shifts
constexpr int kGateReals = 8;  // u00 u01 u10 u11 as (re, im)

template <int N>
struct Layout {
  static constexpr int LB = N < 6 ? N : 6;   // lane bits
  static constexpr int R = 1 << (N - LB);    // complex amplitudes per lane
  static constexpr int LPS = 1 << LB;        // lanes per sample
  static constexpr int SPW = kWave / LPS;    // samples per wave
  static constexpr int D = 1 << N;
  static constexpr int NR = N > 1 ? N - 1 : 1;  // distinct entangler ranges
};

template <typename T>
struct alignas(2 * sizeof(T)) C2 {
  T x, y;
};

struct KParams {
  const void* inputs;
  const void* table;
  void* out;
  const void* gout;  // shifted mode: upstream gradient
  void* dots;        // shifted mode: (n_replicas, batch)
  int64_t in_ld, out_ld, g_ld, batch;
  int32_t first_replica;
  int32_t encoding, imprimitive, measure, n_rounds, n_blocks, sel_layers, n_features;
  double enc_scale, enc_offset, pad_with;
};

// ---------------------------------------------------------------------------
// cross-lane exchange: value held by lane (l ^ MASK)
// ---------------------------------------------------------------------------
__device__ __forceinline__ int xlane_i32_dyn(int v, int mask) { return __shfl_xor(v, mask, 64); }

template <int MASK>
__device__ __forceinline__ int xlane_i32(int v) {
  if constexpr (MASK == 1) {
    return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
  } else if constexpr (MASK == 2) {
    return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
  } else if constexpr (MASK == 8) {
    return __builtin_amdgcn_mov_dpp(v, 0x128, 0xF, 0xF, true); // row_ror:8
  } else if constexpr (MASK == 4) {
    return __builtin_amdgcn_ds_swizzle(v, 0x101F);             // bitmode xor 4
  } else if constexpr (MASK == 16) {
    return __builtin_amdgcn_ds_swizzle(v, 0x401F);             // bitmode xor 16
  } else {
    return __shfl_xor(v, MASK, 64);                            // xor 32: ds_bpermute
  }
}

template <int MASK>
__device__ __forceinline__ float xlane(float v) {
  return __int_as_float(xlane_i32<MASK>(__float_as_int(v)));
}
template <int MASK>
__device__ __forceinline__ double xlane(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = xlane_i32<MASK>(lo);
  hi = xlane_i32<MASK>(hi);
  return __hiloint2double(hi, lo);
}

// sum over the 2^LB lanes of a sample; every lane ends with the total
template <typename T, int LB>
__device__ __forceinline__ T group_sum(T v) {
  if constexpr (LB > 0) v += xlane<1>(v);
  if constexpr (LB > 1) v += xlane<2>(v);
  if constexpr (LB > 2) v += xlane<4>(v);
  if constexpr (LB > 3) v += xlane<8>(v);
  if constexpr (LB > 4) v += xlane<16>(v);
  if constexpr (LB > 5) v += xlane<32>(v);
  return v;
}

__device__ __forceinline__ void qsincos(float x, float* s, float* c) { sincosf(x, s, c); }
__device__ __forceinline__ void qsincos(double x, double* s, double* c) { sincos(x, s, c); }
__device__ __forceinline__ float qsqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double qsqrt(double x) { return sqrt(x); }

__device__ __forceinline__ float flip_sign(float v, uint32_t bit) {
  return __int_as_float(__float_as_int(v) ^ (int)(bit << 31));
}
__device__ __forceinline__ double flip_sign(double v, uint32_t bit) {
  return __hiloint2double(__double2hiint(v) ^ (int)(bit << 31), __double2loint(v));
}

// ---------------------------------------------------------------------------
// single-qubit gate on bit position Q of the amplitude index.
// u = {u00r,u00i,u01r,u01i,u10r,u10i,u11r,u11i}; wave-uniform (SGPR) for the
// shared Rot gates, per-lane for per-sample encodings.
// ---------------------------------------------------------------------------
template <typename T, int N, int Q>
__device__ __forceinline__ void apply_gate(T (&re)[Layout<N>::R], T (&im)[Layout<N>::R],
                                           const T (&u)[8], int lane) {
  using L = Layout<N>;
  if constexpr (Q >= L::LB) {
    constexpr int J = 1 << (Q - L::LB);
#pragma unroll
    for (int r = 0; r < L::R; ++r) {
      if ((r & J) == 0) {
        const int r1 = r | J;
        const T a0r = re[r], a0i = im[r], a1r = re[r1], a1i = im[r1];
        re[r] = u[0] * a0r - u[1] * a0i + u[2] * a1r - u[3] * a1i;
        im[r] = u[0] * a0i + u[1] * a0r + u[2] * a1i + u[3] * a1r;
        re[r1] = u[4] * a0r - u[5] * a0i + u[6] * a1r - u[7] * a1i;
        im[r1] = u[4] * a0i + u[5] * a0r + u[6] * a1i + u[7] * a1r;
      }
    }
  } else {
    const bool hi = (lane >> Q) & 1;
    const T car = hi ? u[6] : u[0], cai = hi ? u[7] : u[1];  // own amplitude: U11 | U00
    const T cbr = hi ? u[4] : u[2], cbi = hi ? u[5] : u[3];  // partner:       U10 | U01
#pragma unroll
    for (int r = 0; r < L::R; ++r) {
      const T ar = re[r], ai = im[r];
      const T pr = xlane<(1 << Q)>(ar), pi = xlane<(1 << Q)>(ai);
      re[r] = car * ar - cai * ai + cbr * pr - cbi * pi;
      im[r] = car * ai + cai * ar + cbr * pi + cbi * pr;
    }
  }
}

// one Rot layer: gates (gate0 + w) on wire w, w = 0..N-1, matrices from the table
template <typename T, int N, int W>
__device__ __forceinline__ void rot_layer(T (&re)[Layout<N>::R], T (&im)[Layout<N>::R],
                                          const T* __restrict__ table, int gate0, int shift_gate,
                                          int shift_var, int lane) {
  if constexpr (W < N) {
    const int g = gate0 + W;
    const int var = (g == shift_gate) ? shift_var : 0;
    const T* __restrict__ up = table + ((size_t)g * kVariants + var) * kGateReals;
    T u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i] = up[i];
    apply_gate<T, N, N - 1 - W>(re, im, u, lane);
    rot_layer<T, N, W + 1>(re, im, table, gate0, shift_gate, shift_var, lane);
  }
}

// per-sample RY(x_w) layer (qml.AngleEmbedding rotation="Y")
template <typename T, int N, int W>
__device__ __forceinline__ void ry_layer(T (&re)[Layout<N>::R], T (&im)[Layout<N>::R],
                                         const T (&xs)[N], int lane) {
  if constexpr (W < N) {
    T s, c;
    qsincos(xs[W] * (T)0.5, &s, &c);
    const T u[8] = {c, (T)0, -s, (T)0, s, (T)0, c, (T)0};
    apply_gate<T, N, N - 1 - W>(re, im, u, lane);
    ry_layer<T, N, W + 1>(re, im, xs, lane);
  }
}

// ---------------------------------------------------------------------------
// LDS tables (built once per block): CZ-ring sign bits and CNOT-ring GF(2) maps
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

template <typename T, int N>
struct Smem {
  using L = Layout<N>;
  static constexpr int kCz = L::NR * kWave;              // u32 [range][lane]
  static constexpr int kCnLane = L::NR * kWave;          // u32 [range][lane]
  static constexpr int kCnReg = L::NR * L::R;            // u32 [range][r]
  static constexpr int kTableWords = kCz + kCnLane + kCnReg;
  static constexpr size_t kTableBytes = ((size_t)kTableWords * 4 + 15) / 16 * 16;
  static constexpr size_t kScratchBytes = (size_t)kWavesPerBlock * kWave * L::R * 2 * sizeof(T);
  static size_t bytes(bool cnot) { return kTableBytes + (cnot ? kScratchBytes : 0); }
};

// ---------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------
template <typename T, int N, bool SHIFT>
__global__ __launch_bounds__(kBlock) void circuit_kernel(const KParams p) {
  using L = Layout<N>;
  using S = Smem<T, N>;
  constexpr int LB = L::LB, R = L::R, LPS = L::LPS, SPW = L::SPW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint32_t* s_cz = reinterpret_cast<uint32_t*>(smem_raw);
  uint32_t* s_cn_lane = s_cz + S::kCz;
  uint32_t* s_cn_reg = s_cn_lane + S::kCnLane;
  C2<T>* s_scratch = reinterpret_cast<C2<T>*>(smem_raw + S::kTableBytes);

  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = tid >> 6;
  const int sub = lane & (LPS - 1);
  const int swave = lane >> LB;  // sample slot inside the wave
  const bool use_cnot = p.imprimitive == 0;

  if constexpr (N > 1) {
    if (use_cnot) {
      for (int i = tid; i < L::NR * kWave; i += kBlock) {
        const int rr = i / kWave + 1;
        s_cn_lane[i] = cnot_ring_map<N>((uint32_t)((i % kWave) & (LPS - 1)), rr);
      }
      for (int i = tid; i < L::NR * R; i += kBlock) {
        const int rr = i / R + 1;
        s_cn_reg[i] = cnot_ring_map<N>((uint32_t)(i % R) << LB, rr);
      }
    } else {
      for (int i = tid; i < L::NR * kWave; i += kBlock) {
        const int rr = i / kWave + 1;
        const uint32_t ls = (uint32_t)((i % kWave) & (LPS - 1));
        uint32_t bits = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) bits |= cz_ring_parity<N>(((uint32_t)r << LB) | ls, rr) << r;
        s_cz[i] = bits;
      }
    }
  }
  __syncthreads();

  const T* __restrict__ table = static_cast<const T*>(p.table);
  const T* __restrict__ inputs = static_cast<const T*>(p.inputs);
  const int n_rot_total = p.n_rounds * p.n_blocks * p.sel_layers * N;

  // parameter-shift replica (wave-uniform)
  int shift_gate = -1, shift_var = 0, shift_blk = -1, shift_wire = 0;
  T shift_sign = 0;
  int replica_local = 0;
  if constexpr (SHIFT) {
    replica_local = blockIdx.y;
    const int rho = p.first_replica + replica_local;
    if (rho < 6 * n_rot_total) {
      shift_gate = rho / 6;
      shift_var = 1 + rho % 6;
    } else {
      const int q = rho - 6 * n_rot_total;
      shift_blk = q / (2 * N);
      shift_wire = (q >> 1) % N;
      shift_sign = (q & 1) ? (T)-1 : (T)1;
    }
  }

  const int64_t groups = (p.batch + SPW - 1) / SPW;
  for (int64_t grp = (int64_t)blockIdx.x * kWavesPerBlock + wave; grp < groups;
       grp += (int64_t)gridDim.x * kWavesPerBlock) {
    const int64_t sample_raw = grp * SPW + swave;
    const bool valid = sample_raw < p.batch;
    const int64_t sample = valid ? sample_raw : p.batch - 1;

    T re[R], im[R];
    T xs[N];
    T dxr[R], dxi[R];  // per-sample diagonal of the RZ encoding layer

    // ---- round-0 inputs ----------------------------------------------------
    if (p.encoding == 2 || p.encoding == 3) {
#pragma unroll
      for (int j = 0; j < N; ++j) xs[j] = inputs[sample * p.in_ld + j] * (T)p.enc_scale;
    } else {
#pragma unroll
      for (int j = 0; j < N; ++j) xs[j] = (T)0;
    }

    T result[N];  // <Z_i> of the last round (expz)
    T pr[R];      // probabilities of the last round (probs)

    for (int round = 0; round < p.n_rounds; ++round) {
      // ---- state preparation -------------------------------------------------
      if (p.encoding == 1) {
        T n2 = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int k = (r << LB) | sub;
          T v = (T)p.pad_with;
          if (k < p.n_features) v = inputs[sample * p.in_ld + k] + (T)p.enc_offset;
          re[r] = v;
          im[r] = 0;
          n2 += v * v;
        }
        n2 = group_sum<T, LB>(n2);
        const T inv = (T)1 / qsqrt(n2);
#pragma unroll
        for (int r = 0; r < R; ++r) re[r] *= inv;
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          re[r] = 0;
          im[r] = 0;
        }
        re[0] = sub == 0 ? (T)1 : (T)0;
      }

      // ---- per-sample RZ-encoding diagonal: prod_j exp(-+ i x_j / 2) ---------
      if (p.encoding == 2) {
        T accr = 1, acci = 0;
#pragma unroll
        for (int q = 0; q < LB; ++q) {  // lane bits: wires N-1-q
          T s, c;
          qsincos(xs[N - 1 - q] * (T)0.5, &s, &c);
          const T si = ((lane >> q) & 1) ? s : -s;
          const T nr = accr * c - acci * si;
          acci = accr * si + acci * c;
          accr = nr;
        }
        dxr[0] = accr;
        dxi[0] = acci;
#pragma unroll
        for (int j = 0; j < N - LB; ++j) {  // register bits: wires N-1-(LB+j)
          T s, c;
          qsincos(xs[N - 1 - (LB + j)] * (T)0.5, &s, &c);
#pragma unroll
          for (int r = 0; r < (1 << j); ++r) {
            const T ar = dxr[r], ai = dxi[r];
            dxr[r | (1 << j)] = ar * c - ai * s;   // bit set:   * (c + i s)
            dxi[r | (1 << j)] = ar * s + ai * c;
            dxr[r] = ar * c + ai * s;              // bit clear: * (c - i s)
            dxi[r] = ai * c - ar * s;
          }
        }
      }

      // ---- blocks ----------------------------------------------------------------
      for (int blk = 0; blk < p.n_blocks; ++blk) {
        if (p.encoding == 2) {
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const T ar = re[r], ai = im[r];
            re[r] = ar * dxr[r] - ai * dxi[r];
            im[r] = ar * dxi[r] + ai * dxr[r];
          }
          if constexpr (SHIFT) {
            if (blk == shift_blk) {  // extra RZ(+-pi/2) on shift_wire
              const int q = N - 1 - shift_wire;
              const T h = (T)0.70710678118654752440;
#pragma unroll
              for (int r = 0; r < R; ++r) {
                const int k = (r << LB) | sub;
                const T si = ((k >> q) & 1) ? h * shift_sign : -h * shift_sign;
                const T ar = re[r], ai = im[r];
                re[r] = ar * h - ai * si;
                im[r] = ar * si + ai * h;
              }
            }
          }
        } else if (p.encoding == 3 && blk == 0) {
          if constexpr (SHIFT) {
            if (shift_blk == 0) {
#pragma unroll
              for (int j = 0; j < N; ++j)
                if (j == shift_wire) xs[j] += shift_sign * (T)1.57079632679489661923;
            }
          }
          ry_layer<T, N, 0>(re, im, xs, lane);
        }

        for (int s = 0; s < p.sel_layers; ++s) {
          const int gate0 = ((round * p.n_blocks + blk) * p.sel_layers + s) * N;
          rot_layer<T, N, 0>(re, im, table, gate0, shift_gate, shift_var, lane);
          if constexpr (N > 1) {
            const int ri = s % (N - 1);  // range index (range = ri + 1)
            if (!use_cnot) {
              const uint32_t bits = s_cz[ri * kWave + lane];
#pragma unroll
              for (int r = 0; r < R; ++r) {
                const uint32_t b = (bits >> r) & 1u;
                re[r] = flip_sign(re[r], b);
                im[r] = flip_sign(im[r], b);
              }
            } else {
              C2<T>* slab = s_scratch + (size_t)(wave * SPW + swave) * L::D;
              const uint32_t lane_term = s_cn_lane[ri * kWave + lane];
#pragma unroll
              for (int r = 0; r < R; ++r) {
                const uint32_t dst = lane_term ^ s_cn_reg[ri * R + r];
                slab[dst] = C2<T>{re[r], im[r]};
              }
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_wave_barrier();
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
              for (int r = 0; r < R; ++r) {
                const int k = (r << LB) | sub;
                const C2<T> a = slab[k];
                re[r] = a.x;
                im[r] = a.y;
              }
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_wave_barrier();
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
          }
        }
      }

      // ---- measurement -------------------------------------------------------------
#pragma unroll
      for (int r = 0; r < R; ++r) pr[r] = re[r] * re[r] + im[r] * im[r];
      if (p.measure == 1) {
#pragma unroll
        for (int w = 0; w < N; ++w) {
          const int q = N - 1 - w;
          T acc = 0;
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const int k = (r << LB) | sub;
            acc += ((k >> q) & 1) ? -pr[r] : pr[r];
          }
          result[w] = group_sum<T, LB>(acc);
        }
      }
      // ---- chain into the next round: x <- out[:, 0:N] ----------------------------
      if (round + 1 < p.n_rounds) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
          T v;
          if (p.measure == 1) {
            v = result[j];
          } else {
            v = __shfl(pr[0], (lane & ~(LPS - 1)) | (j & (LPS - 1)), 64);
          }
          xs[j] = v * (T)p.enc_scale;
        }
      }
    }

    // ---- epilogue --------------------------------------------------------------------
    if constexpr (!SHIFT) {
      T* __restrict__ out = static_cast<T*>(p.out);
      if (p.measure == 0) {
        if (valid) {
#pragma unroll
          for (int r = 0; r < R; ++r) out[sample * p.out_ld + ((r << LB) | sub)] = pr[r];
        }
      } else {
        T v = result[0];
#pragma unroll
        for (int w = 1; w < N; ++w) v = (sub == w) ? result[w] : v;
        if (valid && sub < N) out[sample * p.out_ld + sub] = v;
      }
    } else {
      const T* __restrict__ gout = static_cast<const T*>(p.gout);
      T* __restrict__ dots = static_cast<T*>(p.dots);
      T acc = 0;
      if (p.measure == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) acc += gout[sample * p.g_ld + ((r << LB) | sub)] * pr[r];
        acc = group_sum<T, LB>(acc);
      } else {
#pragma unroll
        for (int w = 0; w < N; ++w) acc += gout[sample * p.g_ld + w] * result[w];
      }
      if (valid && sub == 0) dots[(int64_t)replica_local * p.batch + sample] = acc;
    }
  }
}

// ---------------------------------------------------------------------------
// gate-table preparation: angles (G,3) f64 -> (G, 7, 8) T
// ---------------------------------------------------------------------------
template <typename T>
__global__ void prepare_gates_kernel(const double* __restrict__ angles, T* __restrict__ table,
                                     int64_t n_rot) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rot * kVariants) return;
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
