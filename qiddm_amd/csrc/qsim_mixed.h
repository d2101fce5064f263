// qsim_mixed.h -- density-matrix execution for the hardware-noise study.
//
// Reference: the `*_noise.py` drivers re-create the layers' QNodes on `default.mixed` for sampling
// (src/mnist_noise.py:214-229) and the `_circuit` bodies insert PhaseDamping / AmplitudeDamping /
// DepolarizingChannel after the encoders or in front of the read-out (nn/qdense.py:98-104, 255-261, 1410-1417).
// PennyLane's default.mixed keeps rho (2^n x 2^n, complex128) and applies U rho U^dagger / sum_k K rho K^dagger.
//
// Here one workgroup owns one sample's rho, index (i << n) | j (i: ket/row, j: bra/column; wire w is bit n-1-w of
// each half).  rho lives in LDS while it fits, else in a per-workgroup slab of the caller's workspace (L2-resident:
// 512 KiB at n = 8 in float32).  The circuit arrives as a short program (the host expands templates and rings):
// every op is one sweep over rho, separated by workgroup barriers.  Single-qubit unitaries and channels work on the
// 2 x 2 blocks M = rho[(b_i, b_j)] of their wire:
//     unitary U:            M <- U M U^dagger
//     PhaseDamping(g):      M01, M10 *= sqrt(1-g)
//     AmplitudeDamping(g):  M00 += g M11;  M11 *= 1-g;  M01, M10 *= sqrt(1-g)
//     Depolarizing(p):      M00, M11 <- (1-2p/3) own + (2p/3) other;  M01, M10 *= 1-4p/3
// (the Kraus sums of PennyLane's channel definitions, written out).  Diagonal gates multiply by u_i conj(u_j); CZ by
// sign(i) sign(j); CNOT permutes rows and columns (an involution: swapped in place).  Read-out: the diagonal.
// Forward only -- the reference differentiates nothing on default.mixed.  n <= 8.
#pragma once
#include "qsim_fused.h"

namespace qiddm {

enum MixedKind : int32_t {
  kMixZero = 0,     // rho = |0..0><0..0|
  kMixAmpEmbed,     // rho = |v><v| / |v|^2, v = features + offset padded with pad_with
  kMixPhase,        // diag(1, e^{i phi}) up to a global phase; phi = p + scale * angle_rows[a][sample] (a < 0: constant)
  kMixRY,           // RY(theta), theta as above
  kMixGate,         // fixed unitary gates[a]
  kMixCZ,           // control wire, target a
  kMixCNOT,         // control wire, target a
  kMixPhaseDamp,    // p = gamma
  kMixAmpDamp,      // p = gamma
  kMixDepol,        // p
};

struct MixedOp {
  int32_t kind, wire, a, pad_;
  double p, scale;
};

struct MixedScalars {
  int32_t n, n_ops, measure, n_features;  // measure 0 probs, 1 <Z_w>
  int64_t batch, rows_ld, feat_ld, out_ld;
  double enc_offset, pad_with;
  int32_t slab_in_lds, pad_;
};

template <typename T>
__device__ __forceinline__ V2<T> cmul(V2<T> a, V2<T> b) {
  return V2<T>{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
template <typename T>
__device__ __forceinline__ V2<T> cmulc(V2<T> a, V2<T> b) {  // a * conj(b)
  return V2<T>{a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y};
}

// index of the block element (bi, bj) for block number t: insert bits at positions qj (column half) and qi = qj + n
__device__ __forceinline__ uint32_t insert_two_bits(uint32_t t, int qlo, int qhi) {
  uint32_t lo = t & ((1u << qlo) - 1u);
  uint32_t rest = t >> qlo;
  uint32_t mid_bits = qhi - qlo - 1;
  uint32_t mid = rest & ((1u << mid_bits) - 1u);
  uint32_t hi = rest >> mid_bits;
  return lo | (mid << (qlo + 1)) | (hi << (qhi + 1));
}

template <typename T>
__global__ __launch_bounds__(256) void mixed_kernel(const MixedOp* __restrict__ prog,
                                                    const double* __restrict__ angle_rows,
                                                    const double* __restrict__ feats,
                                                    const double* __restrict__ gates, double* __restrict__ out,
                                                    V2<T>* __restrict__ workspace, const MixedScalars m) {
  using C = V2<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ double s_red[256];
  const int n = m.n, tid = threadIdx.x;
  const uint32_t D = 1u << n, DD = 1u << (2 * n);
  C* rho = m.slab_in_lds ? reinterpret_cast<C*>(smem_raw) : workspace + (size_t)blockIdx.x * DD;

  for (int64_t sample = blockIdx.x; sample < m.batch; sample += gridDim.x) {
    for (int oi = 0; oi < m.n_ops; ++oi) {
      const MixedOp op = prog[oi];
      const int q = n - 1 - op.wire;  // bit of the column half; the row half's is q + n
      switch (op.kind) {
        case kMixZero: {
          for (uint32_t k = tid; k < DD; k += 256) rho[k] = C{k == 0 ? (T)1 : (T)0, (T)0};
          break;
        }
        case kMixAmpEmbed: {
          const double* __restrict__ row = feats + sample * m.feat_ld;
          double part = 0.0;
          for (uint32_t k = tid; k < D; k += 256) {
            const double v = k < (uint32_t)m.n_features ? row[k] + m.enc_offset : m.pad_with;
            part += v * v;
          }
          s_red[tid] = part;
          __syncthreads();
          for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) s_red[tid] += s_red[tid + s];
            __syncthreads();
          }
          const double inv = 1.0 / s_red[0];
          for (uint32_t k = tid; k < DD; k += 256) {
            const uint32_t i = k >> n, j = k & (D - 1);
            const double vi = i < (uint32_t)m.n_features ? row[i] + m.enc_offset : m.pad_with;
            const double vj = j < (uint32_t)m.n_features ? row[j] + m.enc_offset : m.pad_with;
            rho[k] = C{(T)(vi * vj * inv), (T)0};
          }
          break;
        }
        case kMixPhase: {
          const double phi = op.p + (op.a >= 0 ? op.scale * angle_rows[(size_t)op.a * m.rows_ld + sample] : 0.0);
          double sn, cs;
          sincos(phi, &sn, &cs);
          const C up{(T)cs, (T)sn}, dn{(T)cs, (T)-sn};
          for (uint32_t k = tid; k < DD; k += 256) {
            const int bi = (k >> (q + n)) & 1, bj = (k >> q) & 1;
            if (bi != bj) rho[k] = cmul<T>(rho[k], bi ? up : dn);
          }
          break;
        }
        case kMixRY:
        case kMixGate: {
          C u00, u01, u10, u11;
          if (op.kind == kMixRY) {
            const double th = op.p + (op.a >= 0 ? op.scale * angle_rows[(size_t)op.a * m.rows_ld + sample] : 0.0);
            double sn, cs;
            sincos(0.5 * th, &sn, &cs);
            u00 = C{(T)cs, 0};
            u01 = C{(T)-sn, 0};
            u10 = C{(T)sn, 0};
            u11 = C{(T)cs, 0};
          } else {
            const double* __restrict__ g = gates + (size_t)op.a * 8;
            u00 = C{(T)g[0], (T)g[1]};
            u01 = C{(T)g[2], (T)g[3]};
            u10 = C{(T)g[4], (T)g[5]};
            u11 = C{(T)g[6], (T)g[7]};
          }
          for (uint32_t t = tid; t < DD / 4; t += 256) {
            const uint32_t base = insert_two_bits(t, q, q + n);
            const uint32_t cj = 1u << q, ci = 1u << (q + n);
            const C m00 = rho[base], m01 = rho[base | cj], m10 = rho[base | ci], m11 = rho[base | ci | cj];
            // A = U M
            const C a00 = V2<T>{0, 0} + cmul<T>(u00, m00) + cmul<T>(u01, m10);
            const C a01 = cmul<T>(u00, m01) + cmul<T>(u01, m11);
            const C a10 = cmul<T>(u10, m00) + cmul<T>(u11, m10);
            const C a11 = cmul<T>(u10, m01) + cmul<T>(u11, m11);
            // M' = A U^dagger:  M'_{xy} = sum_k A_{xk} conj(U_{yk})
            rho[base] = cmulc<T>(a00, u00) + cmulc<T>(a01, u01);
            rho[base | cj] = cmulc<T>(a00, u10) + cmulc<T>(a01, u11);
            rho[base | ci] = cmulc<T>(a10, u00) + cmulc<T>(a11, u01);
            rho[base | ci | cj] = cmulc<T>(a10, u10) + cmulc<T>(a11, u11);
          }
          break;
        }
        case kMixCZ: {
          const int qt = n - 1 - op.a;
          for (uint32_t k = tid; k < DD; k += 256) {
            const uint32_t i = k >> n, j = k & (D - 1);
            const int si = ((i >> q) & (i >> qt) & 1), sj = ((j >> q) & (j >> qt) & 1);
            if (si != sj) rho[k] = C{-rho[k].x, -rho[k].y};
          }
          break;
        }
        case kMixCNOT: {
          const int qt = n - 1 - op.a;
          for (uint32_t k = tid; k < DD; k += 256) {
            const uint32_t i = k >> n, j = k & (D - 1);
            const uint32_t pi = i ^ (((i >> q) & 1u) << qt), pj = j ^ (((j >> q) & 1u) << qt);
            const uint32_t pk = (pi << n) | pj;
            if (k < pk) {
              const C tmp = rho[k];
              rho[k] = rho[pk];
              rho[pk] = tmp;
            }
          }
          break;
        }
        case kMixPhaseDamp:
        case kMixAmpDamp:
        case kMixDepol: {
          T off, d_own, d_other_to_0, d_other_to_1, d11;
          if (op.kind == kMixPhaseDamp) {
            off = (T)sqrt(1.0 - op.p); d_own = 1; d11 = 1; d_other_to_0 = 0; d_other_to_1 = 0;
          } else if (op.kind == kMixAmpDamp) {
            off = (T)sqrt(1.0 - op.p); d_own = 1; d11 = (T)(1.0 - op.p); d_other_to_0 = (T)op.p; d_other_to_1 = 0;
          } else {
            off = (T)(1.0 - 4.0 * op.p / 3.0); d_own = (T)(1.0 - 2.0 * op.p / 3.0); d11 = d_own;
            d_other_to_0 = (T)(2.0 * op.p / 3.0); d_other_to_1 = d_other_to_0;
          }
          for (uint32_t t = tid; t < DD / 4; t += 256) {
            const uint32_t base = insert_two_bits(t, q, q + n);
            const uint32_t cj = 1u << q, ci = 1u << (q + n);
            const C m00 = rho[base], m11 = rho[base | ci | cj];
            rho[base] = C{d_own * m00.x + d_other_to_0 * m11.x, d_own * m00.y + d_other_to_0 * m11.y};
            rho[base | ci | cj] = C{d11 * m11.x + d_other_to_1 * m00.x, d11 * m11.y + d_other_to_1 * m00.y};
            const C m01 = rho[base | cj], m10 = rho[base | ci];
            rho[base | cj] = C{off * m01.x, off * m01.y};
            rho[base | ci] = C{off * m10.x, off * m10.y};
          }
          break;
        }
        default: break;
      }
      __syncthreads();
    }
    // ---- read-out: the diagonal ------------------------------------------------------------------------
    if (m.measure == 0) {
      for (uint32_t k = tid; k < D; k += 256) out[sample * m.out_ld + k] = (double)rho[((size_t)k << n) | k].x;
    } else {
      for (int w = 0; w < n; ++w) {
        double part = 0.0;
        for (uint32_t k = tid; k < D; k += 256) {
          const double pk = (double)rho[((size_t)k << n) | k].x;
          part += ((k >> (n - 1 - w)) & 1u) ? -pk : pk;
        }
        s_red[tid] = part;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
          if (tid < s) s_red[tid] += s_red[tid + s];
          __syncthreads();
        }
        if (tid == 0) out[sample * m.out_ld + w] = s_red[0];
        __syncthreads();
      }
    }
    __syncthreads();
  }
}

}  // namespace qiddm
