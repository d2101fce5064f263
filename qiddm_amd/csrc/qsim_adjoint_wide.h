// qsim_adjoint_wide.h -- reverse-mode (adjoint) differentiation for n = 11..16 qubits.
//
// Same mathematics as qsim_adjoint.h (forward once, then un-apply every gate to psi and to lambda = diag(g_eff) psi
// while K_ab = sum conj(lambda_a) psi_b of every Rot gate is accumulated), but the 2^n-amplitude vectors no longer fit
// a wavefront's registers: one WORKGROUP owns a sample, psi and lambda live in a per-workgroup slab pair of the
// caller's workspace (L2 / Infinity-Cache resident), and every gate is one sweep over the pairs of its wire between
// workgroup barriers.  That is the plain per-gate formulation -- 2 x 16 B x 2^n of traffic per gate and vector -- not
// the pass-fused one of the tiled forward; what it buys is the gradient itself: without it the only derivative for
// n > 10 is the parameter-shift sweep (2 forward passes per angle, and none at all for amplitude-embedded inputs),
// e.g. C4's 12-qubit QConv2d could not be trained.  Output: the same K slabs as adjoint_kernel (one per workgroup),
// consumed by adjoint_finalize_kernel.
#pragma once
#include "qsim_adjoint.h"

namespace qiddm {

struct WideAdjointScalars {
  int64_t gin_ld;
  int32_t n, want_inputs;
  // raw != 0: "matrix element" mode -- `inputs` rows are complex start vectors psi_0 (2 D reals, not normalised) and
  // `gout` rows complex lambda at the circuit's end; the K slabs then give 2 Re <lambda| dU/dangle |psi_0> (the
  // weight gradient of the quantum convolution from its per-channel h vectors, qsim_qconv_train.h)
  int32_t raw, pad_;
};

constexpr int kWideThreads = 256;

template <typename T>
__device__ __forceinline__ V2<T> wmul(V2<T> a, V2<T> b) {
  return V2<T>{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
template <typename T>
__device__ __forceinline__ V2<T> wmulc(V2<T> a, V2<T> b) {  // conj(a) * b
  return V2<T>{a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x};
}

// sum of `cnt` per-thread values over the workgroup, added to dst[0..cnt) by thread 0 (fixed order)
template <typename T, int CNT>
__device__ __forceinline__ void block_accumulate(const T (&v)[CNT], T* s_red, T* dst) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int i = 0; i < CNT; ++i) {
    const T t = group_sum<T, 6>(v[i], lane);
    if (lane == 0) s_red[wave * CNT + i] = t;
  }
  __syncthreads();
  if (tid < CNT) {
    T tot = 0;
    for (int w = 0; w < kWideThreads / 64; ++w) tot += s_red[w * CNT + tid];
    dst[tid] += tot;
  }
  __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(kWideThreads) void wide_adjoint_kernel(const T* __restrict__ inputs,
                                                                    const T* __restrict__ table,
                                                                    const T* __restrict__ gout,
                                                                    T* __restrict__ k_partials,
                                                                    T* __restrict__ grad_inputs,
                                                                    V2<T>* __restrict__ ws, const KScalars p,
                                                                    const WideAdjointScalars ad) {
  using C = V2<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int n = ad.n, tid = threadIdx.x;
  const uint32_t D = 1u << n;
  const int n_rot = p.n_blocks * p.sel_layers * n;
  const bool use_cnot = p.imprimitive == 0;
  // LDS: K accumulators of this workgroup [n_rot][8], reduction scratch, per-sample encoding data
  T* s_k = reinterpret_cast<T*>(smem_raw);
  T* s_red = s_k + (size_t)n_rot * 8;                 // [4 waves][16]
  T* s_cs = s_red + 4 * 16;                           // [16] cos(x_w / 2)
  T* s_sn = s_cs + 16;                                // [16] sin(x_w / 2)
  T* s_gx = s_sn + 16;                                // [16] input-angle gradients of the sample
  for (int i = tid; i < n_rot * 8; i += kWideThreads) s_k[i] = 0;
  C* psi = ws + (size_t)blockIdx.x * 2 * D;
  C* lam = psi + D;
  __syncthreads();

  auto gate_at = [&](int g, C (&u)[4]) {  // u00 u01 u10 u11 of Rot gate g (variant 0 of the prepared table)
    const T* t = table + (size_t)g * kVariants * kGateReals;
    u[0] = C{t[0], t[1]};
    u[1] = C{t[2], t[3]};
    u[2] = C{t[4], t[5]};
    u[3] = C{t[6], t[7]};
  };
  // one single-qubit matrix m on bit q of vector v
  auto apply_1q = [&](C* v, int q, const C (&m)[4]) {
    const uint32_t bit = 1u << q;
    for (uint32_t t = tid; t < D / 2; t += kWideThreads) {
      const uint32_t i0 = ((t >> q) << (q + 1)) | (t & (bit - 1)), i1 = i0 | bit;
      const C a0 = v[i0], a1 = v[i1];
      v[i0] = wmul<T>(m[0], a0) + wmul<T>(m[1], a1);
      v[i1] = wmul<T>(m[2], a0) + wmul<T>(m[3], a1);
    }
    __syncthreads();
  };
  // ring operations act on v and (when given) on v2 in the same sweep: one barrier for both vectors
  auto cz_ring = [&](C* v, C* v2, int rr) {
    const uint32_t dmask = D - 1u;
    for (uint32_t k = tid; k < D; k += kWideThreads) {
      const uint32_t rot = ((k << rr) | (k >> (n - rr))) & dmask;
      if (__popc(k & rot) & 1) {
        v[k] = C{-v[k].x, -v[k].y};
        if (v2) v2[k] = C{-v2[k].x, -v2[k].y};
      }
    }
    __syncthreads();
  };
  // CNOT(c, t) is its own inverse; a ring applies i = 0..n-1 in order, its inverse in reverse order
  auto cnot = [&](C* v, C* v2, int c, int t_) {
    const int qc = n - 1 - c, qt = n - 1 - t_;
    for (uint32_t k = tid; k < D; k += kWideThreads) {
      if (((k >> qc) & 1u) && !((k >> qt) & 1u)) {
        const uint32_t k1 = k | (1u << qt);
        const C tmp = v[k];
        v[k] = v[k1];
        v[k1] = tmp;
        if (v2) {
          const C tmp2 = v2[k];
          v2[k] = v2[k1];
          v2[k1] = tmp2;
        }
      }
    }
    __syncthreads();
  };
  auto ring_fwd = [&](C* v, int rr) {
    if (!use_cnot) {
      cz_ring(v, nullptr, rr);
    } else {
      for (int i = 0; i < n; ++i) cnot(v, nullptr, i, (i + rr) % n);
    }
  };
  auto ring_back = [&](C* v, C* v2, int rr) {
    if (!use_cnot) {
      cz_ring(v, v2, rr);
    } else {
      for (int i = n - 1; i >= 0; --i) cnot(v, v2, i, (i + rr) % n);
    }
  };
  auto rz_layer = [&](C* v, C* v2, bool conj) {  // diag prod_w exp(-+ i x_w / 2)
    for (uint32_t k = tid; k < D; k += kWideThreads) {
      T fr = 1, fi = 0;
      for (int q = 0; q < n; ++q) {
        const T c = s_cs[n - 1 - q];
        T si = ((k >> q) & 1u) ? s_sn[n - 1 - q] : -s_sn[n - 1 - q];
        si = conj ? -si : si;
        const T nr = fr * c - fi * si;
        fi = fr * si + fi * c;
        fr = nr;
      }
      v[k] = wmul<T>(C{fr, fi}, v[k]);
      if (v2) v2[k] = wmul<T>(C{fr, fi}, v2[k]);
    }
    __syncthreads();
  };

  for (int64_t sample = blockIdx.x; sample < p.batch; sample += gridDim.x) {
    const T* __restrict__ in_row = inputs ? inputs + sample * p.in_ld : nullptr;
    if (tid < 16) s_gx[tid] = 0;
    if (tid < n && p.encoding >= 2) {
      T s, c;
      qsincos((T)(in_row[tid] * (T)p.enc_scale) * (T)0.5, &s, &c);
      s_cs[tid] = c;
      s_sn[tid] = s;
    }
    // ---- forward --------------------------------------------------------------------------------------------
    T amp_inv = 1;
    if (ad.raw) {
      for (uint32_t k = tid; k < D; k += kWideThreads) psi[k] = C{in_row[2 * k], in_row[2 * k + 1]};
    } else if (p.encoding == 1) {
      T part[1] = {0};
      for (uint32_t k = tid; k < D; k += kWideThreads) {
        const T v = k < (uint32_t)p.n_features ? in_row[k] + (T)p.enc_offset : (T)p.pad_with;
        part[0] += v * v;
      }
      if (tid == 0) s_red[63] = 0;
      __syncthreads();
      block_accumulate<T, 1>(part, s_red, s_red + 63);
      amp_inv = (T)1 / qsqrt(s_red[63]);
      for (uint32_t k = tid; k < D; k += kWideThreads) {
        const T v = k < (uint32_t)p.n_features ? in_row[k] + (T)p.enc_offset : (T)p.pad_with;
        psi[k] = C{v * amp_inv, 0};
      }
    } else {
      for (uint32_t k = tid; k < D; k += kWideThreads) psi[k] = C{k == 0 ? (T)1 : (T)0, 0};
    }
    __syncthreads();
    for (int blk = 0; blk < p.n_blocks; ++blk) {
      if (p.encoding == 2) {
        rz_layer(psi, nullptr, false);
      } else if ((p.encoding == 3 && blk == 0) || p.encoding == 4) {
        for (int w = 0; w < n; ++w) {
          const T c = s_cs[w], s = s_sn[w];
          const C m[4] = {C{c, 0}, C{-s, 0}, C{s, 0}, C{c, 0}};
          apply_1q(psi, n - 1 - w, m);
        }
      }
      for (int s = 0; s < p.sel_layers; ++s) {
        const int gate0 = (blk * p.sel_layers + s) * n;
        for (int w = 0; w < n; ++w) {
          C u[4];
          gate_at(gate0 + w, u);
          apply_1q(psi, n - 1 - w, u);
        }
        ring_fwd(psi, s % (n - 1) + 1);
      }
    }
    // ---- lambda = diag(g_eff) psi ----------------------------------------------------------------------------------
    {
      const T* __restrict__ g_row = gout + sample * p.g_ld;
      for (uint32_t k = tid; k < D; k += kWideThreads) {
        if (ad.raw) {
          lam[k] = C{g_row[2 * k], g_row[2 * k + 1]};
          continue;
        }
        T g;
        if (p.measure == 0) {
          g = g_row[k];
        } else {
          g = 0;
          for (int w = 0; w < n; ++w) g += ((k >> (n - 1 - w)) & 1u) ? -g_row[w] : g_row[w];
        }
        lam[k] = C{g * psi[k].x, g * psi[k].y};
      }
      __syncthreads();
    }
    // ---- reverse sweep ----------------------------------------------------------------------------------------------
    for (int blk = p.n_blocks - 1; blk >= 0; --blk) {
      for (int s = p.sel_layers - 1; s >= 0; --s) {
        ring_back(psi, lam, s % (n - 1) + 1);
        const int gate0 = (blk * p.sel_layers + s) * n;
        for (int w = n - 1; w >= 0; --w) {
          C u[4];
          gate_at(gate0 + w, u);
          // U^dagger = (u00* u10*; u01* u11*)
          const C d00{u[0].x, -u[0].y}, d01{u[2].x, -u[2].y}, d10{u[1].x, -u[1].y}, d11{u[3].x, -u[3].y};
          const int q = n - 1 - w;
          const uint32_t bit = 1u << q;
          T k8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          for (uint32_t t = tid; t < D / 2; t += kWideThreads) {
            const uint32_t i0 = ((t >> q) << (q + 1)) | (t & (bit - 1)), i1 = i0 | bit;
            const C a0 = psi[i0], a1 = psi[i1], l0 = lam[i0], l1 = lam[i1];
            const C b0 = wmul<T>(d00, a0) + wmul<T>(d01, a1), b1 = wmul<T>(d10, a0) + wmul<T>(d11, a1);
            const C k00 = wmulc<T>(l0, b0), k01 = wmulc<T>(l0, b1), k10 = wmulc<T>(l1, b0), k11 = wmulc<T>(l1, b1);
            k8[0] += k00.x; k8[1] += k00.y; k8[2] += k01.x; k8[3] += k01.y;
            k8[4] += k10.x; k8[5] += k10.y; k8[6] += k11.x; k8[7] += k11.y;
            psi[i0] = b0;
            psi[i1] = b1;
            lam[i0] = wmul<T>(d00, l0) + wmul<T>(d01, l1);
            lam[i1] = wmul<T>(d10, l0) + wmul<T>(d11, l1);
          }
          block_accumulate<T, 8>(k8, s_red, s_k + (size_t)(gate0 + w) * 8);
        }
      }
      if (p.encoding == 2) {
        // d/dx_w: sum_k z_w(k) Im(conj(lambda_k) psi_k), right after the encoding layer
        T gw[16];
#pragma unroll
        for (int w = 0; w < 16; ++w) gw[w] = 0;
        for (uint32_t k = tid; k < D; k += kWideThreads) {
          const T t = lam[k].x * psi[k].y - lam[k].y * psi[k].x;
#pragma unroll
          for (int w = 0; w < 16; ++w)
            if (w < n) gw[w] += ((k >> (n - 1 - w)) & 1u) ? -t : t;
        }
        block_accumulate<T, 16>(gw, s_red, s_gx);
        rz_layer(psi, lam, true);
      } else if ((p.encoding == 3 && blk == 0) || p.encoding == 4) {
        for (int w = n - 1; w >= 0; --w) {
          const T c = s_cs[w], sn = s_sn[w];
          const int q = n - 1 - w;
          const uint32_t bit = 1u << q;
          T acc[1] = {0};
          for (uint32_t t = tid; t < D / 2; t += kWideThreads) {
            const uint32_t i0 = ((t >> q) << (q + 1)) | (t & (bit - 1)), i1 = i0 | bit;
            const C a0 = psi[i0], a1 = psi[i1], l0 = lam[i0], l1 = lam[i1];
            // d/dtheta of RY: Re <lambda| (-iY) |psi>, both taken after the gate
            acc[0] += (l1.x * a0.x + l1.y * a0.y) - (l0.x * a1.x + l0.y * a1.y);
            psi[i0] = C{c * a0.x + sn * a1.x, c * a0.y + sn * a1.y};
            psi[i1] = C{c * a1.x - sn * a0.x, c * a1.y - sn * a0.y};
            lam[i0] = C{c * l0.x + sn * l1.x, c * l0.y + sn * l1.y};
            lam[i1] = C{c * l1.x - sn * l0.x, c * l1.y - sn * l0.y};
          }
          block_accumulate<T, 1>(acc, s_red, s_gx + w);
        }
      }
    }
    // ---- input gradients ----------------------------------------------------------------------------------------------
    if (ad.want_inputs && grad_inputs != nullptr) {
      T* __restrict__ gin = grad_inputs + sample * ad.gin_ld;
      if (p.encoding >= 2) {
        // RZ: the angle is x * scale -> chain rule; RY: d/dtheta above is per unit of theta = x * scale, times 1/2
        // is already inside (-iY/2 * 2 Re): both need only the encoding scale
        if (tid < n) gin[tid] = s_gx[tid] * (T)p.enc_scale;
      } else if (p.encoding == 1) {
        // psi0 = v / |v| (real): dL/dv_k = (gpsi_k - psi0_k <gpsi, psi0>) / |v|, gpsi = 2 Re lambda_0
        T part[1] = {0};
        for (uint32_t k = tid; k < D; k += kWideThreads) part[0] += (T)2 * lam[k].x * psi[k].x;
        if (tid == 0) s_red[63] = 0;
        __syncthreads();
        block_accumulate<T, 1>(part, s_red, s_red + 63);
        const T dotp = s_red[63];
        for (uint32_t k = tid; k < (uint32_t)p.n_features; k += kWideThreads)
          gin[k] = ((T)2 * lam[k].x - psi[k].x * dotp) * amp_inv;
      }
    }
    __syncthreads();
  }
  for (int i = tid; i < n_rot * 8; i += kWideThreads) k_partials[(size_t)blockIdx.x * n_rot * 8 + i] = s_k[i];
}

// ---------------------------------------------------------------------------
// U^T of a weight-only StronglyEntanglingLayers circuit for n = 11, 12 (the eval-mode QConv2d route of C4): one
// workgroup per column j evolves |j> IN PLACE in row j of the output (complex128), so that row j of `ut` is column j
// of U and no workspace is needed.  ut[j * D + k] = <k|U|j>.
// ---------------------------------------------------------------------------
template <int UNUSED = 0>  // (a template only for linkage: the header is included by two translation units)
__global__ __launch_bounds__(kWideThreads) void wide_unitary_kernel(const double* __restrict__ angles,
                                                                    V2<double>* __restrict__ ut, int n, int sel_layers,
                                                                    int use_cnot) {
  using C = V2<double>;
  const int tid = threadIdx.x;
  const uint32_t D = 1u << n;
  for (uint32_t j = blockIdx.x; j < D; j += gridDim.x) {
    C* psi = ut + (size_t)j * D;
    for (uint32_t k = tid; k < D; k += kWideThreads) psi[k] = C{k == j ? 1.0 : 0.0, 0.0};
    __syncthreads();
    for (int s = 0; s < sel_layers; ++s) {
      for (int w = 0; w < n; ++w) {
        const double* a = angles + ((size_t)s * n + w) * 3;
        double c, sn, ca, sa, cb, sb;
        sincos(0.5 * a[1], &sn, &c);
        sincos(0.5 * (a[0] + a[2]), &sa, &ca);
        sincos(0.5 * (a[0] - a[2]), &sb, &cb);
        const C u00{ca * c, -sa * c}, u01{-cb * sn, -sb * sn}, u10{cb * sn, -sb * sn}, u11{ca * c, sa * c};
        const int q = n - 1 - w;
        const uint32_t bit = 1u << q;
        for (uint32_t t = tid; t < D / 2; t += kWideThreads) {
          const uint32_t i0 = ((t >> q) << (q + 1)) | (t & (bit - 1)), i1 = i0 | bit;
          const C a0 = psi[i0], a1 = psi[i1];
          psi[i0] = wmul<double>(u00, a0) + wmul<double>(u01, a1);
          psi[i1] = wmul<double>(u10, a0) + wmul<double>(u11, a1);
        }
        __syncthreads();
      }
      const int rr = s % (n - 1) + 1;
      if (!use_cnot) {
        const uint32_t dmask = D - 1u;
        for (uint32_t k = tid; k < D; k += kWideThreads) {
          const uint32_t rot = ((k << rr) | (k >> (n - rr))) & dmask;
          if (__popc(k & rot) & 1) psi[k] = C{-psi[k].x, -psi[k].y};
        }
        __syncthreads();
      } else {
        for (int i = 0; i < n; ++i) {
          const int qc = n - 1 - i, qt = n - 1 - (i + rr) % n;
          for (uint32_t k = tid; k < D; k += kWideThreads) {
            if (((k >> qc) & 1u) && !((k >> qt) & 1u)) {
              const uint32_t k1 = k | (1u << qt);
              const C tmp = psi[k];
              psi[k] = psi[k1];
              psi[k1] = tmp;
            }
          }
          __syncthreads();
        }
      }
    }
  }
}

}  // namespace qiddm
