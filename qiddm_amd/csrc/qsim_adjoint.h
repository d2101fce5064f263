// qsim_adjoint.h -- reverse-mode (adjoint) differentiation of the fused circuits, n <= 10.
//
// What PennyLane's diff_method="backprop" (reference nn/qdense.py:37, 246-alt, 419; nn/qconv.py:46)
// obtains by keeping every intermediate state on the autograd tape, done the statevector way:
//   forward once -> psi_final;   lambda = diag(g_eff) psi_final   (g_eff = upstream gradient folded on
//   the measurement: d(sum_k g_k p_k) or d(sum_w g_w <Z_w>));  then walk the circuit backwards,
//   un-applying every gate to BOTH vectors.  For a gate U(theta) on one wire
//        dL/dtheta = 2 Re sum_{a,b} (dU/dtheta)_{ab} K_{ab},   K_{ab} = sum_pairs conj(lambda_a) psi_b
//   with psi taken before and lambda after the gate.  The kernel accumulates the four complex K_{ab} of
//   every Rot gate over all samples (they share the weights); the host contracts them with the analytic
//   dRot/d(phi, theta, omega) in float64.  Per-sample input gradients (RZ / RY data encoding, amplitude
//   embedding) are finished in the kernel.  Cost ~5 forward passes, independent of the number of
//   parameters (a parameter-shift sweep costs 2 per parameter).
#pragma once
#include "qsim_fused.h"

namespace qiddm {

template <typename T>
__device__ __forceinline__ V2<T> neg_i(V2<T> a) {
  return V2<T>{a.y, -a.x};
}
// acc + conj(lam) * psi
template <typename T>
__device__ __forceinline__ V2<T> cjfma(V2<T> lam, V2<T> psi, V2<T> acc) {
  return cfma<T>(lam, psi, neg_i<T>(psi), acc);
}

// Sum 8 per-lane values over the whole wave; value idx ends up (fully reduced) in the lane whose logical
// number is < 8 with idx = 4*b0 + 2*b1 + b2 and is added to acc[idx] (LDS, private to the wave).
// Halving exchange for the first three steps: 7 + 3 exchanged values instead of 48.
// STORE: write the totals instead of adding them (a slot group that is filled once per use needs no zeroing and no
// read-modify-write round trip)
template <typename T, bool STORE = false>
__device__ __forceinline__ void wave_reduce8_into(const T (&v)[8], int lane, int llane, T* acc) {
  const bool b0 = llane & 1, b1 = llane & 2, b2 = llane & 4;
  T w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const T keep = b0 ? v[4 + i] : v[i];
    const T send = b0 ? v[i] : v[4 + i];
    w[i] = keep + xlane<1>(send, lane);
  }
  T u[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const T keep = b1 ? w[2 + i] : w[i];
    const T send = b1 ? w[i] : w[2 + i];
    u[i] = keep + xlane<2>(send, lane);
  }
  T t = (b2 ? u[1] : u[0]) + xlane<4>(b2 ? u[0] : u[1], lane);
  t += xlane<8>(t, lane);
  t += xlane<16>(t, lane);
  t += xlane<32>(t, lane);
  if (llane < 8) {
    T* slot = acc + (b0 ? 4 : 0) + (b1 ? 2 : 0) + (b2 ? 1 : 0);
    if constexpr (STORE) *slot = t;
    else *slot += t;
  }
}

constexpr int kFoldedAdjointMaxQubits = 9;

struct AdjointScalars {
  int64_t gin_ld;   // row stride of grad_inputs
  int32_t want_inputs;
  int32_t pad_;
};

// where the upstream gradient dL/d(out) of a sample comes from
template <typename T>
struct RowGrad {  // a row of a (batch, 2^N | N) matrix
  const T* __restrict__ row;
  __device__ __forceinline__ T prob(int k, T) const { return row[k]; }
  __device__ __forceinline__ T expz(int w) const { return row[w]; }
};
// the quantum convolution's post-processing y[co] = clamp(p[2 co] * D/2, 0, 1) (reference nn/qconv.py:58-69)
// differentiated in place: dL/dp_k is read from the (B, C_out, Ho, Wo) gradient of y at this pixel, for even
// k < 2 C_out and where the clamp lets it through (torch: min <= value <= max), and is zero elsewhere
template <typename T>
struct ConvGrad {
  const double* __restrict__ gy;  // at (b, 0, oi, oj)
  int64_t ch_stride;              // Ho * Wo
  int C_out;
  T post_scale;                   // D / 2
  __device__ __forceinline__ T prob(int k, T p2) const {
    if ((k & 1) != 0 || (k >> 1) >= C_out || p2 * post_scale > (T)1) return (T)0;
    return (T)gy[(int64_t)(k >> 1) * ch_stride] * post_scale;
  }
  __device__ __forceinline__ T expz(int) const { return (T)0; }
};

template <typename T, int N>
struct AdjointEngine {
  using E = Engine<T, N>;
  using L = typename E::L;
  using C = V2<T>;
  static constexpr int LB = L::LB, R = L::R, LPS = L::LPS;

  E fwd;    // gate table U (or, folded, the per-layer tables)
  E dag;    // same engine reading the dagger table U^dagger
  T* kacc;  // this wave's K accumulators [n_rot][8] in LDS (folded: gradient sums [n_layers][2][kFoldSlots])

  // ---- folded reverse sweep (CZ circuits, no RY data encoding) ----------------------------------------------
  // The forward is  a <- D_l a ; a <- RY^l a  per layer (qsim_fused.h).  Walking back, with (psi, lambda) right after
  // the layer's RYs:
  //   d/dtheta_w = Re <lambda| (-iY_w) |psi>  = sum_pairs Re(conj(lambda_1) psi_0 - conj(lambda_0) psi_1)
  //                (invariant under un-applying the other wires' RYs: taken wire by wire, then RY_w^dagger on both)
  //   d/dalpha_w = Im <lambda| Z_w |psi>      = sum_k z_w(k) Im(conj(lambda_k) psi_k), right after D_l;
  //                alpha^l_w = phi^l_w + omega^{l-1}_w (+ the data angle at a block start), so the SAME number is
  //                d/dphi^l_w, d/domega^{l-1}_w and (times the encoding scale) a term of d/dx_w
  // then D_l^* on both vectors.  Two 8-value wave reductions per layer instead of one per gate, no dagger table,
  // no dRot contraction afterwards.  The diagonal dropped after the last RY layer has zero gradient.
  static constexpr int kFoldSlots = N <= 8 ? 8 : 16;

  template <int J>
  __device__ __forceinline__ T ry_back_pairs(C (&psi)[R], C (&lam)[R], T c, T s) const {
    // (re, im) products accumulated as a pair: the operands are register pairs already, so these are plain packed
    // fmas -- written out as a scalar sum the SLP vectoriser packs them too, but with ~14 moves per wire to pair them up
    C g2 = C{(T)0, (T)0};
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if ((r & J) == 0) {
        const C p0 = psi[r], p1 = psi[r | J], l0 = lam[r], l1 = lam[r | J];
        g2 = __builtin_elementwise_fma(l1, p0, g2);
        g2 = __builtin_elementwise_fma(-l0, p1, g2);
        // RY^dagger = [[c, s], [-s, c]]
        psi[r] = __builtin_elementwise_fma(bcast<T>(s), p1, bcast<T>(c) * p0);
        psi[r | J] = __builtin_elementwise_fma(bcast<T>(-s), p0, bcast<T>(c) * p1);
        lam[r] = __builtin_elementwise_fma(bcast<T>(s), l1, bcast<T>(c) * l0);
        lam[r | J] = __builtin_elementwise_fma(bcast<T>(-s), l0, bcast<T>(c) * l1);
      }
    }
    return g2.x + g2.y;
  }
  template <int W>
  __device__ __forceinline__ void ry_back_wires(C (&psi)[R], C (&lam)[R], const typename E::FoldedLayer& f,
                                                T (&gth)[16]) const {
    if constexpr (W < N) {
      constexpr int Q = N - 1 - W;
      T c = f.ry[W].x;
      const T s = f.ry[W].y;
      // ordering token (see qsim_wide_cz_adjoint.h: undo_down_to): the previous wire's reduction is finished before this
      // wire's un-application overwrites the values it reads, instead of being parked with a copy of them
      if constexpr (W > 0) c = fma(gth[W - 1], (T)0, c);
      if constexpr (E::template kind_of<W>() == E::kReg) {
        gth[W] = ry_back_pairs<(1 << (Q >= LB ? Q - LB : 0))>(psi, lam, c, s);
      } else if constexpr (E::template kind_of<W>() == E::kSwap) {
        fwd.template swap_reg0_with_lane_bit<Q>(psi);
        fwd.template swap_reg0_with_lane_bit<Q>(lam);
        // after the swap every lane holds complete pairs, but each pair is held by ONE of the two lanes that
        // exchanged: the sum over lanes counts every pair once
        gth[W] = ry_back_pairs<1>(psi, lam, c, s);
        fwd.template swap_reg0_with_lane_bit<Q>(psi);
        fwd.template swap_reg0_with_lane_bit<Q>(lam);
      } else {
        const bool hi = (fwd.llane >> Q) & 1;
        const T sg = hi ? s : -s;  // the forward coefficient of the partner in this lane's row
        C acc2 = C{(T)0, (T)0};
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const C pp = xlane2<(1 << Q), T>(psi[r], fwd.lane);
          const C lp = xlane2<(1 << Q), T>(lam[r], fwd.lane);
          acc2 = __builtin_elementwise_fma(lam[r], pp, acc2);  // Re(conj(lambda_own) psi_partner) = .x + .y
          psi[r] = __builtin_elementwise_fma(bcast<T>(-sg), pp, bcast<T>(c) * psi[r]);
          lam[r] = __builtin_elementwise_fma(bcast<T>(-sg), lp, bcast<T>(c) * lam[r]);
        }
        const T acc = acc2.x + acc2.y;
        gth[W] = hi ? acc : -acc;
      }
      ry_back_wires<W + 1>(psi, lam, f, gth);
    }
  }

  // state preparation + all layers of one round on the folded tables (fwd.s_gates -> the round's first layer)
  template <typename Src>
  __device__ __forceinline__ void forward_round_folded(const KScalars& p, const Src& amp_src, const T (&xs)[N],
                                                       C (&psi)[R], C (&dx)[R], T (&cs)[N], T (&sn)[N], T& amp_inv,
                                                       int n_layers_total) const {
    const int lane = fwd.lane, sub = fwd.sub;
    amp_inv = 1;
    if (p.encoding == 1) {
      T n2 = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int k = (r << LB) | sub;
        T v = (T)p.pad_with;
        if (k < p.n_features) v = amp_src(k) + (T)p.enc_offset;
        psi[r] = C{v, (T)0};
        n2 += v * v;
      }
      n2 = group_sum<T, LB>(n2, lane);
      amp_inv = (T)1 / qsqrt(n2);
#pragma unroll
      for (int r = 0; r < R; ++r) psi[r] = C{psi[r].x * amp_inv, (T)0};
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) psi[r] = C{(T)0, (T)0};
      psi[0] = C{sub == 0 ? (T)1 : (T)0, (T)0};
    }
    if (p.encoding == 2) {
      fwd.half_angle_sincos(xs, cs, sn);
      fwd.rz_diagonal(cs, sn, dx);
    }
    fwd.folded_round(p, psi, dx, 0, n_layers_total);
  }

  // gacc: this wave's [layers][2][kFoldSlots] accumulators of the round (theta block, then alpha block)
  // (register diet for R >= 8: the register-bit phases are read from LDS where they are used and the data diagonal
  //  is rebuilt from (cs, sn) at the block starts instead of staying live through the whole sweep)
  __device__ __forceinline__ void reverse_round_folded(const KScalars& p, C (&psi)[R], C (&lam)[R],
                                                       const T (&cs)[N], const T (&sn)[N], T (&gx)[N], T* gacc) const {
    const int layers = p.n_blocks * p.sel_layers;
    const int llane = fwd.llane;
#pragma unroll
    for (int w = 0; w < N; ++w) gx[w] = 0;
    for (int li = layers - 1; li >= 0; --li) {
      const int s = li % p.sel_layers;
      typename E::FoldedLayer f;
      fwd.load_folded_light(li, (N > 1 && li > 0) ? ((li - 1) % p.sel_layers) % (N - 1) : -1, f);
      T gth[16], gal[16];
#pragma unroll
      for (int w = 0; w < 16; ++w) {
        gth[w] = 0;
        gal[w] = 0;
      }
      ry_back_wires<0>(psi, lam, f, gth);
      // d/dalpha_w right after the diagonal
      T t[R], tsum = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        t[r] = lam[r].x * psi[r].y - lam[r].y * psi[r].x;
        tsum += t[r];
      }
#pragma unroll
      for (int w = 0; w < N; ++w) {
        const int q = N - 1 - w;
        if (q >= LB) {
          T acc = 0;
#pragma unroll
          for (int r = 0; r < R; ++r) acc += ((r >> (q >= LB ? q - LB : 0)) & 1) ? -t[r] : t[r];
          gal[w] = acc;
        } else {
          gal[w] = ((llane >> q) & 1) ? -tsum : tsum;
        }
      }
      if (s == 0 && p.encoding == 2) {
#pragma unroll
        for (int w = 0; w < N; ++w) gx[w] += gal[w];  // the data angle sits in the same diagonal
      }
      {
        T* dst = gacc + (size_t)li * 2 * kFoldSlots;
        T v8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = gth[j];
        wave_reduce8_into<T>(v8, fwd.lane, llane, dst);
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = gal[j];
        wave_reduce8_into<T>(v8, fwd.lane, llane, dst + kFoldSlots);
        if constexpr (N > 8) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v8[j] = gth[8 + j];
          wave_reduce8_into<T>(v8, fwd.lane, llane, dst + 8);
#pragma unroll
          for (int j = 0; j < 8; ++j) v8[j] = gal[8 + j];
          wave_reduce8_into<T>(v8, fwd.lane, llane, dst + kFoldSlots + 8);
        }
      }
      // D_l^* on both vectors
      const bool upload = s == 0 && p.encoding == 2;
      if (upload) {
        C dxl[R];
        fwd.rz_diagonal(cs, sn, dxl);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          psi[r] = cmul2<T>(dxl[r], psi[r], neg_i<T>(psi[r]));
          lam[r] = cmul2<T>(dxl[r], lam[r], neg_i<T>(lam[r]));
        }
      }
      const C* thi_p = reinterpret_cast<const C*>(fwd.s_gates + (size_t)li * E::S::kFoldStride) + N + LPS;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        C ph = f.tlo;
        if constexpr (R > 1) {
          const C th = thi_p[r];
          ph = cmul2<T>(th, ph, times_i<T>(ph));
        }
        const uint32_t sb = ((f.cz >> r) & 1u) << 31;
        const C a = cmul2<T>(ph, psi[r], neg_i<T>(psi[r]));  // conj(ph) * psi
        const C b = cmul2<T>(ph, lam[r], neg_i<T>(lam[r]));
        psi[r] = C{flip_sign(a.x, sb), flip_sign(a.y, sb)};
        lam[r] = C{flip_sign(b.x, sb), flip_sign(b.y, sb)};
      }
    }
  }

  // K_{ab} += conj(lambda_a) psib_b for one in-register pair, then both vectors are un-applied
  template <int J>
  __device__ __forceinline__ void step_regs(C (&psi)[R], C (&lam)[R], const C* m, T (&k)[8]) const {
    const C* lo = m;
    const C* hi = m + 4;
    C k00{0, 0}, k01{0, 0}, k10{0, 0}, k11{0, 0};
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if ((r & J) == 0) {
        const C a0 = psi[r], a1 = psi[r | J];
        const C b0 = cfma<T>(a1, lo[2], lo[3], cmul2<T>(a0, lo[0], lo[1]));
        const C b1 = cfma<T>(a0, hi[2], hi[3], cmul2<T>(a1, hi[0], hi[1]));
        const C l0 = lam[r], l1 = lam[r | J];
        k00 = cjfma<T>(l0, b0, k00);
        k01 = cjfma<T>(l0, b1, k01);
        k10 = cjfma<T>(l1, b0, k10);
        k11 = cjfma<T>(l1, b1, k11);
        psi[r] = b0;
        psi[r | J] = b1;
        lam[r] = cfma<T>(l1, lo[2], lo[3], cmul2<T>(l0, lo[0], lo[1]));
        lam[r | J] = cfma<T>(l0, hi[2], hi[3], cmul2<T>(l1, hi[0], hi[1]));
      }
    }
    k[0] = k00.x; k[1] = k00.y; k[2] = k01.x; k[3] = k01.y;
    k[4] = k10.x; k[5] = k10.y; k[6] = k11.x; k[7] = k11.y;
  }
  // lane-bit version: this lane holds component a = its bit; `h` = the matching half of U^dagger
  template <int Q>
  __device__ __forceinline__ void step_lane(C (&psi)[R], C (&lam)[R], const C* h, C& s_own, C& s_x) const {
    s_own = C{0, 0};
    s_x = C{0, 0};
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const C own = psi[r];
      const C par = xlane2<(1 << Q), T>(own, fwd.lane);
      const C b = cfma<T>(par, h[2], h[3], cmul2<T>(own, h[0], h[1]));  // psi before the gate, own component
      const C l_own = lam[r];
      const C l_par = xlane2<(1 << Q), T>(l_own, fwd.lane);
      s_own = cjfma<T>(l_own, b, s_own);  // K_{aa}
      s_x = cjfma<T>(l_par, b, s_x);      // K_{(1-a) a}
      psi[r] = b;
      lam[r] = cfma<T>(l_par, h[2], h[3], cmul2<T>(l_own, h[0], h[1]));
    }
  }

  // un-apply the Rot gate on wire W of the layer starting at gate0 and bank its K
  template <int W>
  __device__ __forceinline__ void rot_steps_back(C (&psi)[R], C (&lam)[R], int gate0) const {
    if constexpr (W < N) {
      constexpr int Q = N - 1 - W;
      C m[8];
      dag.template load_gate<W>(gate0, m);
      T k[8];
      if constexpr (E::template kind_of<W>() == E::kReg) {
        step_regs<(1 << (Q >= LB ? Q - LB : 0))>(psi, lam, m, k);
      } else if constexpr (E::template kind_of<W>() == E::kSwap) {
        fwd.template swap_reg0_with_lane_bit<Q>(psi);
        fwd.template swap_reg0_with_lane_bit<Q>(lam);
        step_regs<1>(psi, lam, m, k);
        fwd.template swap_reg0_with_lane_bit<Q>(psi);
        fwd.template swap_reg0_with_lane_bit<Q>(lam);
      } else {
        C s_own, s_x;
        step_lane<Q>(psi, lam, m, s_own, s_x);
        const bool hi = (fwd.llane >> Q) & 1;
        const T z = 0;
        k[0] = hi ? z : s_own.x; k[1] = hi ? z : s_own.y;   // K00
        k[2] = hi ? s_x.x : z;   k[3] = hi ? s_x.y : z;     // K01 = conj(lam_0) psib_1: held by bit-1 lanes
        k[4] = hi ? z : s_x.x;   k[5] = hi ? z : s_x.y;     // K10
        k[6] = hi ? s_own.x : z; k[7] = hi ? s_own.y : z;   // K11
      }
      wave_reduce8_into<T>(k, fwd.lane, fwd.llane, kacc + (size_t)(gate0 + W) * 8);
      rot_steps_back<W + 1>(psi, lam, gate0);
    }
  }

  // inverse entangler ring on one vector
  __device__ __forceinline__ void ring_back(C (&a)[R], int ri, bool use_cnot) const {
    if constexpr (N > 1) {
      if (!use_cnot) {
        fwd.ring(a, ri, false);  // CZ ring is its own inverse
      } else {
        const uint32_t lane_term = fwd.s_cn_lane[ri * kWave + fwd.llane];
#pragma unroll
        for (int r = 0; r < R; ++r) fwd.s_slab[(r << LB) | fwd.sub] = a[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int r = 0; r < R; ++r) a[r] = fwd.s_slab[lane_term ^ fwd.s_cn_reg[ri * R + r]];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
  }

  // per-sample RY(x_w) un-application + d/dx_w (accumulated per lane into gx[w])
  template <int W>
  __device__ __forceinline__ void ry_steps_back(C (&psi)[R], C (&lam)[R], const T (&cs)[N], const T (&sn)[N],
                                                T (&gx)[N]) const {
    if constexpr (W < N) {
      constexpr int Q = N - 1 - W;
      const T c = cs[W], s = sn[W], z = 0;
      if constexpr (Q >= LB) {
        // RY^dagger = [[c, s], [-s, c]]
        const C m[8] = {C{c, z}, C{z, c}, C{s, z}, C{z, s}, C{c, z}, C{z, c}, C{-s, z}, C{z, -s}};
        T k[8];
        step_regs<(1 << (Q >= LB ? Q - LB : 0))>(psi, lam, m, k);
        // 2 Re sum (dRY/dtheta)_{ab} K_{ab},  dRY/dtheta = 1/2 [[-s, -c], [c, -s]]
        gx[W] += -s * (k[0] + k[6]) + c * (k[4] - k[2]);
      } else {
        const bool hi = (fwd.llane >> Q) & 1;
        const T sp = hi ? -s : s;  // partner coefficient of RY^dagger for this lane's row
        const C h[4] = {C{c, z}, C{z, c}, C{sp, z}, C{z, sp}};
        C s_own, s_x;
        step_lane<Q>(psi, lam, h, s_own, s_x);
        gx[W] += -s * s_own.x + (hi ? -c : c) * s_x.x;
      }
      ry_steps_back<W + 1>(psi, lam, cs, sn, gx);
    }
  }

  // ---- one round, forward: |0..0> or the embedded row -> psi_final.  Leaves what the reverse sweep needs
  // (half-angle sin/cos, the RZ diagonal, 1/|v| of the embedding) in the caller's registers.
  template <typename Src>
  __device__ __forceinline__ void forward_round(const KScalars& p, const Src& amp_src, const T (&xs)[N], C (&psi)[R],
                                                C (&dx)[R], T (&cs)[N], T (&sn)[N], T& amp_inv) const {
    const bool use_cnot = p.imprimitive == 0;
    const int lane = fwd.lane, sub = fwd.sub;
    amp_inv = 1;
    if (p.encoding == 1) {
      T n2 = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int k = (r << LB) | sub;
        T v = (T)p.pad_with;
        if (k < p.n_features) v = amp_src(k) + (T)p.enc_offset;
        psi[r] = C{v, (T)0};
        n2 += v * v;
      }
      n2 = group_sum<T, LB>(n2, lane);
      amp_inv = (T)1 / qsqrt(n2);
#pragma unroll
      for (int r = 0; r < R; ++r) psi[r] = C{psi[r].x * amp_inv, (T)0};
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) psi[r] = C{(T)0, (T)0};
      psi[0] = C{sub == 0 ? (T)1 : (T)0, (T)0};
    }
    if (p.encoding >= 2) fwd.half_angle_sincos(xs, cs, sn);
    if (p.encoding == 2) fwd.rz_diagonal(cs, sn, dx);
    C gate_m[8];
    fwd.template load_gate<0>(0, gate_m);
    const int n_rot = p.n_blocks * p.sel_layers * N;
    for (int blk = 0; blk < p.n_blocks; ++blk) {
      if (p.encoding == 2) {
#pragma unroll
        for (int r = 0; r < R; ++r) psi[r] = cmul2<T>(dx[r], psi[r], times_i<T>(psi[r]));
      } else if ((p.encoding == 3 && blk == 0) || p.encoding == 4) {
        fwd.template ry_layer<0>(psi, cs, sn);
      }
      for (int s = 0; s < p.sel_layers; ++s) {
        const int gate0 = (blk * p.sel_layers + s) * N;
        fwd.rot_layer(psi, gate0, gate0 + N < n_rot ? gate0 + N : 0, gate_m);
        if constexpr (N > 1) fwd.ring(psi, s % (N - 1), use_cnot);
      }
    }
  }

  // <Z_w> of every wire, broadcast to all lanes of the sample
  __device__ __forceinline__ void measure_expz(const C (&psi)[R], T (&result)[N]) const {
    T pr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) pr[r] = psi[r].x * psi[r].x + psi[r].y * psi[r].y;
#pragma unroll
    for (int w = 0; w < N; ++w) {
      const int q = N - 1 - w;
      T acc = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) acc += ((((r << LB) | fwd.sub) >> q) & 1) ? -pr[r] : pr[r];
      result[w] = group_sum<T, LB>(acc, fwd.lane);
    }
  }

  // lambda = diag(sum_w g_w z_w(k)) psi   (upstream gradient on the <Z_w> read-out)
  __device__ __forceinline__ void seed_expz(const T (&gw)[N], const C (&psi)[R], C (&lam)[R]) const {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int k = (r << LB) | fwd.sub;
      T g = 0;
#pragma unroll
      for (int w = 0; w < N; ++w) g += ((k >> (N - 1 - w)) & 1) ? -gw[w] : gw[w];
      lam[r] = C{g * psi[r].x, g * psi[r].y};
    }
  }

  // ---- one round, backwards: un-applies every gate to psi and lambda, banks K of the Rot gates in kacc and
  // the per-lane partial input gradients in gx (angle encodings)
  __device__ __forceinline__ void reverse_round(const KScalars& p, C (&psi)[R], C (&lam)[R], const C (&dx)[R],
                                                const T (&cs)[N], const T (&sn)[N], T (&gx)[N]) const {
    const bool use_cnot = p.imprimitive == 0;
    const int llane = fwd.llane;
#pragma unroll
    for (int w = 0; w < N; ++w) gx[w] = 0;
    for (int blk = p.n_blocks - 1; blk >= 0; --blk) {
      for (int s = p.sel_layers - 1; s >= 0; --s) {
        if constexpr (N > 1) {
          ring_back(psi, s % (N - 1), use_cnot);
          ring_back(lam, s % (N - 1), use_cnot);
        }
        rot_steps_back<0>(psi, lam, (blk * p.sel_layers + s) * N);
      }
      if (p.encoding == 2) {
        // d/dx_w of RZ_w(x_w): sum_k z_w(k) Im(conj(lambda_k) psi_k), psi right after the encoding layer
        T t[R], tsum = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          t[r] = lam[r].x * psi[r].y - lam[r].y * psi[r].x;
          tsum += t[r];
        }
#pragma unroll
        for (int w = 0; w < N; ++w) {
          const int q = N - 1 - w;
          if (q >= LB) {
            T acc = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) acc += ((r >> (q >= LB ? q - LB : 0)) & 1) ? -t[r] : t[r];
            gx[w] += acc;
          } else {
            gx[w] += ((llane >> q) & 1) ? -tsum : tsum;
          }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
          psi[r] = cmul2<T>(dx[r], psi[r], neg_i<T>(psi[r]));  // conj(dx) * psi
          lam[r] = cmul2<T>(dx[r], lam[r], neg_i<T>(lam[r]));
        }
      } else if ((p.encoding == 3 && blk == 0) || p.encoding == 4) {
        ry_steps_back<0>(psi, lam, cs, sn, gx);
      }
    }
  }

  // ---- one sample (group): forward, then the reverse sweep ------------------------------------------
  // g_src: upstream gradient of this sample (RowGrad / ConvGrad).
  // gin_row: where this sample's input gradient goes (may be null).
  template <typename Src, typename GSrc>
  __device__ __forceinline__ void run(const KScalars& p, const Src& amp_src, T (&xs)[N],
                                      const GSrc& g_src, T* __restrict__ gin_row, bool valid,
                                      bool folded = false) const {
    const int lane = fwd.lane, sub = fwd.sub;
    C psi[R], dx[R];
    T cs[N], sn[N];
    T amp_inv;
    if (folded)
      forward_round_folded(p, amp_src, xs, psi, dx, cs, sn, amp_inv, p.n_blocks * p.sel_layers);
    else
      forward_round(p, amp_src, xs, psi, dx, cs, sn, amp_inv);
    // ---- lambda = diag(g_eff) psi_final ----------------------------------------------------------------
    C lam[R];
    if (p.measure == 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const T g = valid ? g_src.prob((r << LB) | sub, psi[r].x * psi[r].x + psi[r].y * psi[r].y) : (T)0;
        lam[r] = C{g * psi[r].x, g * psi[r].y};
      }
    } else {
      T gw[N];
#pragma unroll
      for (int w = 0; w < N; ++w) gw[w] = valid ? g_src.expz(w) : (T)0;
      seed_expz(gw, psi, lam);
    }
    T gx[N];
    if (folded)
      reverse_round_folded(p, psi, lam, cs, sn, gx, kacc);
    else
      reverse_round(p, psi, lam, dx, cs, sn, gx);
    // ---- input gradients -----------------------------------------------------------------------------------------
    if (gin_row != nullptr) {
      if (p.encoding >= 2) {
        T v = 0;
#pragma unroll
        for (int w = 0; w < N; ++w) {
          const T tot = group_sum<T, LB>(gx[w], lane) * (T)p.enc_scale;
          v = fma((T)(sub == w ? 1 : 0), tot, v);
        }
        if (valid && sub < N) gin_row[sub] = v;
      } else if (p.encoding == 1) {
        // psi0 = v / |v| (real): dL/dv_k = (gpsi_k - psi0_k <gpsi, psi0>) / |v|, gpsi = 2 Re lambda_0
        T dotp = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) dotp += (T)2 * lam[r].x * psi[r].x;
        dotp = group_sum<T, LB>(dotp, lane);
        if (valid) {
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const int k = (r << LB) | sub;
            if (k < p.n_features) gin_row[k] = ((T)2 * lam[r].x - psi[r].x * dotp) * amp_inv;
          }
        }
      }
    }
  }
};

// ---------------------------------------------------------------------------
// kernel: grid-strided over samples; every block writes its K partial slab [n_rot][8]
// ---------------------------------------------------------------------------
// CONV: the samples are the output pixels of a quantum convolution -- features read from the image through
// PatchSrc (the unfold never exists), upstream gradient read from dL/dy through ConvGrad; `inputs` / `gout` unused.
template <typename T, int N, bool CONV>
__global__ __launch_bounds__(4 * kWave) void adjoint_kernel(const T* __restrict__ inputs,
                                                            const T* __restrict__ table,
                                                            const T* __restrict__ gout,
                                                            T* __restrict__ k_partials,
                                                            T* __restrict__ grad_inputs, const KScalars p,
                                                            const AdjointScalars ad,
                                                            const double* __restrict__ img,
                                                            const double* __restrict__ gy, const ConvScalars cv) {
  using E = Engine<T, N>;
  using L = typename E::L;
  using S = Smem<T, N>;
  constexpr int LB = L::LB, SPW = L::SPW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int n_rot = p.n_blocks * p.sel_layers * N;
  const int waves = blockDim.x >> 6;
  AdjointEngine<T, N> adj;
  adj.fwd.carve(smem_raw, n_rot);
  // after the forward engine's region: dagger gate table, then the per-wave K accumulators
  unsigned char* extra = smem_raw + S::bytes(n_rot, p.imprimitive == 0, waves);
  T* dag_gates = reinterpret_cast<T*>(extra);
  T* kall = dag_gates + (size_t)n_rot * kLdsGateReals;
  // CZ circuits: folded tables (appended to the gate table by qiddm_prepare_gates), per-layer gradient sums
  // (n <= 9: with 16 amplitudes per lane the folded sweep spills -- measured 3x slower at n = 10)
  const bool folded = p.fold != 0 && N >= 2 && N <= kFoldedAdjointMaxQubits;
  const int n_layers = n_rot / N;
  const int acc_len = folded ? n_layers * 2 * AdjointEngine<T, N>::kFoldSlots : n_rot * 8;
  if (folded) {
    adj.fwd.fill_folded_from_table(table + (size_t)n_rot * kVariants * kGateReals, n_layers);
  } else {
    adj.fwd.fill_gates_from_table(table, n_rot, -1, 0);
    for (int g = threadIdx.x; g < n_rot; g += blockDim.x) {
      const T* u = table + (size_t)g * kVariants * kGateReals;
      // U^dagger: (u00*, u10*; u01*, u11*)
      E::put_gate(dag_gates + (size_t)g * kLdsGateReals, u[0], -u[1], u[4], -u[5], u[2], -u[3], u[6], -u[7]);
    }
  }
  for (int i = threadIdx.x; i < waves * acc_len; i += blockDim.x) kall[i] = 0;
  adj.fwd.fill_rings(p.imprimitive == 0);
  __syncthreads();
  adj.dag = adj.fwd;
  adj.dag.s_gates = dag_gates;
  const int wave = threadIdx.x >> 6;
  adj.kacc = kall + (size_t)wave * acc_len;
  const int sub = adj.fwd.sub;
  const int swave = adj.fwd.llane >> LB;

  const int64_t groups = (p.batch + SPW - 1) / SPW;
  for (int64_t grp = (int64_t)blockIdx.x * waves + wave; grp < groups; grp += (int64_t)gridDim.x * waves) {
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
    // an out-of-range slot (sub-wave layouts only) repeats the last sample with a zero upstream
    // gradient: it still takes part in the wave-wide exchanges but adds nothing to K
    T* gin_row = ad.want_inputs ? grad_inputs + sample * ad.gin_ld : nullptr;
    if constexpr (CONV) {
      const int64_t pixels = (int64_t)cv.Ho * cv.Wo;
      const int64_t b = sample / pixels;
      const int pix = (int)(sample - b * pixels);
      const int oi = pix / cv.Wo, oj = pix - oi * cv.Wo;
      const PatchSrc<T> src{img + (size_t)b * cv.C * cv.H * cv.W, cv.H, cv.W, cv.kh, cv.kw, oi - cv.ph, oj - cv.pw};
      const ConvGrad<T> g_src{gy + (size_t)b * cv.C_out * pixels + pix, pixels, cv.C_out, (T)cv.post_scale};
      adj.run(p, src, xs, g_src, gin_row, valid, folded);
    } else {
      adj.run(p, RowSrc<T>{inputs + sample * p.in_ld}, xs, RowGrad<T>{gout + sample * p.g_ld}, gin_row, valid,
              folded);
    }
  }
  __syncthreads();
  // block partial: sum the waves' accumulators in a fixed order (deterministic); slab stride n_rot * 8 either way
  for (int i = threadIdx.x; i < acc_len; i += blockDim.x) {
    T tot = 0;
    for (int w = 0; w < waves; ++w) tot += kall[(size_t)w * acc_len + i];
    k_partials[(size_t)blockIdx.x * n_rot * 8 + i] = tot;
  }
}

// dL/dx of the quantum convolution from the per-pixel feature gradients (M = B Ho Wo rows of F = C kh kw):
// the transpose of torch.nn.Unfold as a gather -- every input element sums the kh*kw patch entries it appeared in,
// in a fixed order (deterministic; no atomics).
template <typename T>
__global__ __launch_bounds__(256) void qconv_fold_kernel(const T* __restrict__ gfeat, double* __restrict__ gx,
                                                         int64_t total, const ConvScalars cv) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int j = (int)(idx % cv.W);
  int64_t t = idx / cv.W;
  const int i = (int)(t % cv.H);
  t /= cv.H;
  const int c = (int)(t % cv.C);
  const int64_t b = t / cv.C;
  const int F = cv.C * cv.kh * cv.kw;
  double acc = 0;
  for (int di = 0; di < cv.kh; ++di) {
    const int oi = i - di + cv.ph;
    if (oi < 0 || oi >= cv.Ho) continue;
    for (int dj = 0; dj < cv.kw; ++dj) {
      const int oj = j - dj + cv.pw;
      if (oj < 0 || oj >= cv.Wo) continue;
      acc += (double)gfeat[((b * cv.Ho + oi) * cv.Wo + oj) * F + (c * cv.kh + di) * cv.kw + dj];
    }
  }
  gx[idx] = acc;
}

// 2 Re sum_ab (dU/dangle)_ab K_ab for the three angles of one Rot gate (float64); k = K00 K01 K10 K11 as (re, im)
//   U00 =  e^{-ia} c   U01 = -e^{ib} s   U10 = e^{-ib} s   U11 = e^{ia} c,  a = (phi+omega)/2, b = (phi-omega)/2
__device__ __forceinline__ void rot_grad_from_k(const double (&k)[8], double phi, double theta, double omega,
                                                double* __restrict__ out3) {
  double c, s, ca, sa, cb, sb;
  sincos(0.5 * theta, &s, &c);
  sincos(0.5 * (phi + omega), &sa, &ca);
  sincos(0.5 * (phi - omega), &sb, &cb);
  // Re(z * K) for z = (zr, zi), K = (kr, ki): zr*kr - zi*ki
  auto re_mul = [](double zr, double zi, double kr, double ki) { return zr * kr - zi * ki; };
  const double u00r = ca * c, u00i = -sa * c, u01r = -cb * s, u01i = -sb * s;
  const double u10r = cb * s, u10i = -sb * s, u11r = ca * c, u11i = sa * c;
  // multiply by -i/2: (x + iy)(-i/2) = (y - ix)/2 ; by +i/2: (-y + ix)/2
  auto mi = [&](double xr, double xi, double kr, double ki) { return re_mul(0.5 * xi, -0.5 * xr, kr, ki); };
  auto pi_ = [&](double xr, double xi, double kr, double ki) { return re_mul(-0.5 * xi, 0.5 * xr, kr, ki); };
  const double d_phi = mi(u00r, u00i, k[0], k[1]) + pi_(u01r, u01i, k[2], k[3]) + mi(u10r, u10i, k[4], k[5]) +
                       pi_(u11r, u11i, k[6], k[7]);
  const double d_omega = mi(u00r, u00i, k[0], k[1]) + mi(u01r, u01i, k[2], k[3]) + pi_(u10r, u10i, k[4], k[5]) +
                         pi_(u11r, u11i, k[6], k[7]);
  // d/dtheta: c -> -s/2, s -> c/2
  const double t00r = ca * (-0.5 * s), t00i = -sa * (-0.5 * s), t01r = -cb * (0.5 * c), t01i = -sb * (0.5 * c);
  const double t10r = cb * (0.5 * c), t10i = -sb * (0.5 * c), t11r = ca * (-0.5 * s), t11i = sa * (-0.5 * s);
  const double d_theta = re_mul(t00r, t00i, k[0], k[1]) + re_mul(t01r, t01i, k[2], k[3]) +
                         re_mul(t10r, t10i, k[4], k[5]) + re_mul(t11r, t11i, k[6], k[7]);
  out3[0] = 2.0 * d_phi;
  out3[1] = 2.0 * d_theta;
  out3[2] = 2.0 * d_omega;
}

// ---------------------------------------------------------------------------
// finalize: sum the per-workgroup K slabs in a fixed order and contract with the analytic
// dRot/d(phi, theta, omega):   dL/dangle = 2 Re sum_ab (dU/dangle)_ab K_ab      (float64)
//   U00 =  e^{-ia} c   U01 = -e^{ib} s   U10 = e^{-ib} s   U11 = e^{ia} c,  a = (phi+omega)/2, b = (phi-omega)/2
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kWave) void adjoint_finalize_kernel(const T* __restrict__ k_partials,
                                                                 int64_t n_partials, int64_t n_rot,
                                                                 const double* __restrict__ angles,
                                                                 double* __restrict__ grad_angles) {
  // one wavefront per gate: lanes stride over the workgroup slabs, then a fixed-order butterfly
  const int64_t g = blockIdx.x;
  if (g >= n_rot) return;
  const int lane = threadIdx.x;
  double k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t pidx = lane; pidx < n_partials; pidx += kWave) {
    const T* src = k_partials + (pidx * n_rot + g) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) k[i] += (double)src[i];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) k[i] = group_sum<double, 6>(k[i], lane);
  if (lane != 0) return;
  rot_grad_from_k(k, angles[g * 3 + 0], angles[g * 3 + 1], angles[g * 3 + 2], grad_angles + g * 3);
}

// folded slabs: [layer][2][slots] sums of d/dtheta and d/dalpha per wire; alpha^l = phi^l + omega^(l-1), so
//   d/dphi^l = A[l],  d/dtheta^l = Th[l],  d/domega^l = A[l+1] inside the round (0 for a round's last layer)
template <typename T>
__global__ __launch_bounds__(kWave) void adjoint_finalize_folded_kernel(const T* __restrict__ partials,
                                                                        int64_t n_partials, int64_t slab_stride,
                                                                        int n, int layers_per_round, int slots,
                                                                        int64_t n_rot, double* __restrict__ grad_angles) {
  const int64_t g = blockIdx.x;
  if (g >= n_rot) return;
  const int lane = threadIdx.x;
  const int layer = (int)(g / n), w = (int)(g - (int64_t)layer * n);
  const bool has_next = (layer % layers_per_round) + 1 < layers_per_round;
  double th = 0, al = 0, an = 0;
#pragma unroll 4
  for (int64_t pidx = lane; pidx < n_partials; pidx += kWave) {
    const T* src = partials + pidx * slab_stride + (size_t)layer * 2 * slots;
    th += (double)src[w];
    al += (double)src[slots + w];
    if (has_next) an += (double)src[2 * slots + slots + w];
  }
  th = group_sum<double, 6>(th, lane);
  al = group_sum<double, 6>(al, lane);
  an = group_sum<double, 6>(an, lane);
  if (lane == 0) {
    grad_angles[g * 3 + 0] = al;
    grad_angles[g * 3 + 1] = th;
    grad_angles[g * 3 + 2] = an;
  }
}

}  // namespace qiddm
