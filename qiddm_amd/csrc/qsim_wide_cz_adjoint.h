// qsim_wide_cz_adjoint.h -- reverse-mode gradients of the CZ-entangled circuit family for 11 <= n <= 16 qubits, on the
// pass structure of qsim_wide_cz.h.
//
// Replaces ``diff_method="backprop"`` (torch autograd through default.qubit.torch, reference nn/qdense.py:37, 419) for the
// wide configurations -- one QNode round of BASELINE config 5's 16-qubit QIDDM_LL-style circuit.  The per-gate wide
// adjoint (qsim_adjoint_wide.h) sweeps both vectors once per GATE between workgroup barriers; here the reverse sweep
// mirrors the forward: with the state right after pass p of the forward in one slab (psi) and the cotangent
// lambda = dL/d(psi*) in the other, reverse pass p undoes, on its tile in registers,
//
//     [real RY of layer p+1 on local positions 0..9]  .  [diagonal D^{p+1}]  .  [real RY of layer p on positions P..9]
//
// on BOTH vectors and takes the gradients from the values it already holds:
//     d/dtheta_w = Re <lambda| (-iY_w) |psi>   right after the RY  (pairs along the wire's bit: registers, or the DPP /
//                                               permlane partner the un-application fetches anyway)
//     d/dalpha_w = Im <lambda| Z_w |psi>        at the diagonal (signed sums over the index bits, as for <Z_w>)
// alpha^l = phi^l + omega^{l-1} (+ x at a block start), so d/dalpha is at once d/dphi^l, d/domega^{l-1} and a term of
// d/dx.  Layer 0 acts on |0..0>: its RY gradients are  Re (RY^dagger lambda)[e_w]  -- weighted sums of Re lambda with
// the known product-state factors -- taken in the last reverse pass without a further sweep.  A round of L layers costs
// L - 1 forward sweeps of one slab and L - 1 reverse sweeps of two, instead of ~6n sweeps of two per layer.
//
// Output: per-workgroup slabs of per-layer angle-gradient sums in the layout adjoint_finalize_folded_kernel reads
// ([layer][theta | alpha][16 wires], slab stride n_rot * 8), summed over the workgroup's samples in a fixed order, and
// grad_inputs (B, n) per sample.
#pragma once
#include "qsim_wide_cz.h"

namespace qiddm {

template <typename T>
struct WideCzAdjSmem {
  // [ry][ua][ux] as the forward, then doubles: acc [L][2][16], wave rows [waves][48], gin [16], g [16], xs [16]
  __host__ __device__ static size_t bytes(int64_t layers, int n) {
    return (size_t)layers * n * 2 * 2 * sizeof(T) + 16 * 2 * sizeof(T) +
           ((size_t)layers * 32 + kWideMaxWaves * 48 + 16 + 16 + 16) * sizeof(double);
  }
};

template <typename T, int N>
struct WideCzAdj : WideCz<T, N> {
  using B = WideCz<T, N>;
  using G = typename B::G;
  using C = typename B::C;
  static constexpr int R = 16, NB = B::NB, NT = B::NT, P = B::P;

  // ---- un-apply the real RY on local position POS on psi (a) and lambda (l); th += this lane's share of
  //      Re <l| (-iY) |a>, evaluated before the un-application ---------------------------------------------------
  template <int J>
  __device__ __forceinline__ void undo_pairs(C (&a)[R], C (&l)[R], T c, T s, T& th) const {
    // two FMA chains (no product temporaries: written as sums of products the scheduler hoists every multiply and the
    // kernel needs 500 registers), each pair un-applied right after its contribution
    C t2 = C{(T)0, (T)0};  // (re, im) products as one packed chain: the operands are register pairs already
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if ((r & J) == 0) {
        const C a0 = a[r], a1 = a[r | J], l0 = l[r], l1 = l[r | J];
        t2 = __builtin_elementwise_fma(l1, a0, t2);
        t2 = __builtin_elementwise_fma(-l0, a1, t2);
        a[r] = __builtin_elementwise_fma(bcast<T>(s), a1, bcast<T>(c) * a0);
        a[r | J] = __builtin_elementwise_fma(bcast<T>(-s), a0, bcast<T>(c) * a1);
        l[r] = __builtin_elementwise_fma(bcast<T>(s), l1, bcast<T>(c) * l0);
        l[r | J] = __builtin_elementwise_fma(bcast<T>(-s), l0, bcast<T>(c) * l1);
      }
    }
    th += t2.x + t2.y;
  }
  template <int POS>
  __device__ __forceinline__ void undo_pos(C (&a)[R], C (&l)[R], T c, T s, T& th) const {
    if constexpr (POS == 0) {
      undo_pairs<1>(a, l, c, s, th);
    } else if constexpr (POS <= 4) {
      constexpr int LBIT = POS - 1;
      const bool hi = (this->llane >> LBIT) & 1;
      const T sg = hi ? -s : s;   // RY^dagger = [[c, s], [-s, c]]
      C tc2 = C{(T)0, (T)0};
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const C pa = xlane2<(1 << LBIT), T>(a[r], this->lane);
        const C pl = xlane2<(1 << LBIT), T>(l[r], this->lane);
        tc2 = __builtin_elementwise_fma(l[r], pa, tc2);   // own lambda with the partner's psi
        a[r] = __builtin_elementwise_fma(bcast<T>(sg), pa, bcast<T>(c) * a[r]);
        l[r] = __builtin_elementwise_fma(bcast<T>(sg), pl, bcast<T>(c) * l[r]);
      }
      const T tc = tc2.x + tc2.y;
      th += hi ? tc : -tc;
    } else if constexpr (POS <= 6) {
      constexpr int LBIT = POS - 1;
      this->eng.template swap_reg0_with_lane_bit<LBIT>(a);
      this->eng.template swap_reg0_with_lane_bit<LBIT>(l);
      undo_pairs<1>(a, l, c, s, th);
      this->eng.template swap_reg0_with_lane_bit<LBIT>(a);
      this->eng.template swap_reg0_with_lane_bit<LBIT>(l);
    } else {
      undo_pairs<(1 << (POS - 6))>(a, l, c, s, th);
    }
  }
  // positions 9 down to FROM of `layer`; th[pos] accumulates
  template <int SET, int POS, int FROM>
  __device__ __forceinline__ void undo_down_to(C (&a)[R], C (&l)[R], int layer, T (&th)[10]) const {
    if constexpr (POS >= FROM) {
      const C cs = this->template ry_coeff<SET, POS>(layer);
      T c = cs.x;
      // Ordering token.  The reduction th[POS + 1] is off the critical path, so the scheduler parks it -- with the 64
      // registers of pre-update psi / lambda it reads -- behind the next positions' updates, position after position,
      // until the kernel needs 500 registers.  Making this position's cosine depend on it (times zero: exact) finishes
      // each reduction before the next un-application starts.
      if constexpr (POS < 9) c = fma(th[POS + 1], (T)0, c);
      undo_pos<POS>(a, l, c, cs.y, th[POS]);
      undo_down_to<SET, POS - 1, FROM>(a, l, layer, th);
    }
  }
  // un-apply on lambda only (the turnaround pass: psi returns to what the slab already holds)
  template <int SET, int POS, int FROM>
  __device__ __forceinline__ void grad_and_undo_lambda(C (&a)[R], C (&l)[R], int layer, T (&th)[10]) const {
    // a is a scratch copy of psi: undoing it too keeps the pair formulas of undo_pos valid position after position
    undo_down_to<SET, POS, FROM>(a, l, layer, th);
  }

  // ---- signed sums of c_k = Im(conj(lambda_k) psi_k) over the index bits: d/dalpha of every wire ---------------
  struct Signed {
    T tot = 0;
    T reg[4] = {0, 0, 0, 0};
    T tile[NB > 0 ? NB : 1];
  };
  __device__ __forceinline__ static T signed_token(const Signed& m) {
    T v = m.tot + m.reg[0] + m.reg[1] + m.reg[2] + m.reg[3];
#pragma unroll
    for (int i = 0; i < NB; ++i) v += m.tile[i];
    return v;
  }
  __device__ __forceinline__ void alpha_tile(const C (&a)[R], const C (&l)[R], uint32_t t, Signed& m) const {
    T ck[R];
#pragma unroll
    for (int r = 0; r < R; ++r) ck[r] = l[r].x * a[r].y - l[r].y * a[r].x;
    T tot = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) tot += ck[r];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      T sgn = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) sgn += ((r >> rb) & 1) ? -ck[r] : ck[r];
      m.reg[rb] += sgn;
    }
    m.tot += tot;
#pragma unroll
    for (int i = 0; i < NB; ++i) m.tile[i] += ((t >> i) & 1u) ? -tot : tot;
  }
  // `token`: a value computed from the signed sums of alpha_tile; the phase is made to depend on it (times zero: exact)
  // so that those reductions are finished -- not parked with a copy of psi / lambda -- before the vectors are overwritten
  template <int SET>
  __device__ __forceinline__ void undo_diag(C (&a)[R], C (&l)[R], const typename B::Diag& d, uint32_t t,
                                            T token = (T)0) const {
    C tt = C{(T)1, (T)0};
#pragma unroll
    for (int i = 0; i < NB; ++i) tt = wide_cmul<T>(tt, wide_sel<T>((t >> i) & 1u, d.ut[i]));
    tt = wide_cmul<T>(tt, d.pl);
    tt.x = fma(token, (T)0, tt.x);
    const uint32_t kt = B::template tile_index_bits<SET>(t);
    const uint32_t x = kt | d.kl;
    const uint32_t y = B::rotl_n(kt, d.range) | d.rl;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      C ph = wide_cmul<T>(tt, d.pr[r]);
      const uint32_t sb = (uint32_t)(__popc((x | B::template reg_index_bits<SET>(r)) & (y | d.rr_reg[r])) & 1) << 31;
      ph = C{flip_sign(ph.x, sb), flip_sign(-ph.y, sb)};   // conj(phase) * sign
      a[r] = wide_cmul<T>(ph, a[r]);
      l[r] = wide_cmul<T>(ph, l[r]);
    }
  }

  // ---- wave row of the pass's sums: [0,16) theta of `cur`, [16,32) alpha of `cur`, [32,48) theta of `prev` ----------
  template <int SET>
  __device__ __forceinline__ void put_theta(double* __restrict__ row, int base, const T (&th)[10], int from) const {
#pragma unroll
    for (int pos = 0; pos < 10; ++pos) {
      if (pos >= from) {
        const T v = group_sum<T, 6>(th[pos], this->lane);
        if (this->lane == 0) row[base + (N - 1 - G::local_bit(SET, pos))] = (double)v;
      }
    }
  }
  template <int SET>
  __device__ __forceinline__ void put_signed(double* __restrict__ row, int base, const Signed& m) const {
#pragma unroll
    for (int j = 1; j <= 6; ++j) {
      const T v = group_sum<T, 6>(((this->llane >> (j - 1)) & 1) ? -m.tot : m.tot, this->lane);
      if (this->lane == 0) row[base + (N - 1 - G::local_bit(SET, j))] = (double)v;
    }
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      const int pos = rb == 0 ? 0 : 6 + rb;
      const T v = group_sum<T, 6>(m.reg[rb], this->lane);
      if (this->lane == 0) row[base + (N - 1 - G::local_bit(SET, pos))] = (double)v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const T v = group_sum<T, 6>(m.tile[i], this->lane);
      if (this->lane == 0) row[base + (N - 1 - G::tile_bit(SET, i))] = (double)v;
    }
  }

  // ---- turnaround: psi_L = [RY of the last layer on positions P..9] psi;  lambda_L = g_eff psi_L;  theta gradients of
  //      those positions;  lambda un-applied back to the stage the psi slab holds ----------------------------------------
  template <int SET>
  __device__ __forceinline__ void turnaround(const C* __restrict__ psi, C* __restrict__ lam, int layer_last,
                                             const KScalars& p, const T* __restrict__ g_row,
                                             const double* __restrict__ s_g, double* __restrict__ row) const {
    T th[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) th[i] = 0;
    // <Z> read-out: g_eff(k) = sum_w g_w (1 - 2 b_w(k)) = G0 - 2 (lane part + register part + tile part)
    T g0 = 0, g_lane = 0, g_reg[R], g_tb[NB > 0 ? NB : 1];
    if (p.measure == 1) {
#pragma unroll
      for (int w = 0; w < N; ++w) g0 += (T)s_g[w];
#pragma unroll
      for (int j = 1; j <= 6; ++j)
        g_lane += ((this->llane >> (j - 1)) & 1) ? (T)s_g[N - 1 - G::local_bit(SET, j)] : (T)0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        T v = 0;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
          const int pos = rb == 0 ? 0 : 6 + rb;
          v += ((r >> rb) & 1) ? (T)s_g[N - 1 - G::local_bit(SET, pos)] : (T)0;
        }
        g_reg[r] = v;
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) g_tb[i] = (T)s_g[N - 1 - G::tile_bit(SET, i)];
    }
    for (uint32_t t = (uint32_t)this->wave; t < (uint32_t)NT; t += (uint32_t)this->waves) {
      C a[R], l[R];
      this->template load_tile<SET>(psi, t, a);
      this->template ry_from<SET, P>(a, layer_last);
      if (p.measure == 1) {
        T g_tile = 0;
#pragma unroll
        for (int i = 0; i < NB; ++i) g_tile += ((t >> i) & 1u) ? g_tb[i] : (T)0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const T ge = g0 - (T)2 * (g_lane + g_reg[r] + g_tile);
          l[r] = C{ge * a[r].x, ge * a[r].y};
        }
      } else {
#pragma unroll
        for (int r1 = 0; r1 < R / 2; ++r1) {
          using T2 = V2<T>;
          const T2 gg = *reinterpret_cast<const T2*>(g_row + this->template pair_index<SET>(t, r1));
          l[2 * r1] = C{gg.x * a[2 * r1].x, gg.x * a[2 * r1].y};
          l[2 * r1 + 1] = C{gg.y * a[2 * r1 + 1].x, gg.y * a[2 * r1 + 1].y};
        }
      }
      undo_down_to<SET, 9, P>(a, l, layer_last, th);
      this->template store_tile<SET>(lam, t, l);
    }
    put_theta<SET>(row, 0, th, P);
  }

  // ---- reverse pass p >= 1 (set SET = p & 1): undo layer p+1 on 0..9, D^{p+1}, layer p on P..9 ----------------------
  template <int SET>
  __device__ __forceinline__ void reverse_pass(C* __restrict__ psi, C* __restrict__ lam, int layer_base, int pidx,
                                               const KScalars& p, double* __restrict__ row) const {
    const int lcur = layer_base + pidx + 1, lprev = layer_base + pidx;
    typename B::Diag d;
    const int s_prev = pidx % p.sel_layers;
    this->template build_diag<SET>(lcur, p.encoding == 2 && ((pidx + 1) % p.sel_layers == 0), (s_prev % (N - 1)) + 1, d);
    T th_cur[10], th_prev[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) th_cur[i] = th_prev[i] = 0;
    Signed m;
#pragma unroll
    for (int i = 0; i < NB; ++i) m.tile[i] = 0;
    for (uint32_t t = (uint32_t)this->wave; t < (uint32_t)NT; t += (uint32_t)this->waves) {
      C a[R], l[R];
      this->template load_tile<SET>(psi, t, a);
      this->template load_tile<SET>(lam, t, l);
      undo_down_to<SET, 9, 0>(a, l, lcur, th_cur);
      alpha_tile(a, l, t, m);
      undo_diag<SET>(a, l, d, t, signed_token(m) + th_cur[0]);
      undo_down_to<SET, 9, P>(a, l, lprev, th_prev);
      this->template store_tile<SET>(psi, t, a);
      this->template store_tile<SET>(lam, t, l);
    }
    put_theta<SET>(row, 0, th_cur, 0);
    put_signed<SET>(row, 16, m);
    put_theta<SET>(row, 32, th_prev, P);
  }

  // ---- last reverse pass (p = 0, set A): undo layer 1 on 0..9, D^1; layer 0 analytically from lambda ------------------
  __device__ __forceinline__ void reverse_pass0(const C* __restrict__ psi, const C* __restrict__ lam, int layer_base,
                                                const KScalars& p, double* __restrict__ row) const {
    constexpr int SET = 0;
    const int lcur = layer_base + 1;
    typename B::Diag d;
    this->template build_diag<SET>(lcur, p.encoding == 2 && (1 % p.sel_layers == 0), 1, d);   // ring of layer 0: range 1
    T th_cur[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) th_cur[i] = 0;
    Signed m;
#pragma unroll
    for (int i = 0; i < NB; ++i) m.tile[i] = 0;
    // product-state factors of layer 0: f_w(b) = b ? sin : cos of theta_w / 2; derivative selector dsel_w(b) = b ? cos : -sin
    // (row 1 of RY^dagger).  weight of wire w at index k: prod_{w' != w} f_{w'}(b_{w'}) * dsel_w(b_w)
    T fl = 1;                 // lane part of the plain product
    T fl_ex[6];               // lane part with lane wire j replaced by its selector
    C cs_lane[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      cs_lane[j] = this->s_ry[layer_base * N + (N - 1 - G::local_bit(SET, j + 1))];
      fl *= ((this->llane >> j) & 1) ? cs_lane[j].y : cs_lane[j].x;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      T v = 1;
#pragma unroll
      for (int j2 = 0; j2 < 6; ++j2) {
        const bool b = (this->llane >> j2) & 1;
        v *= j2 == j ? (b ? cs_lane[j2].x : -cs_lane[j2].y) : (b ? cs_lane[j2].y : cs_lane[j2].x);
      }
      fl_ex[j] = v;
    }
    C cs_reg[4], cs_tile[NB > 0 ? NB : 1];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
      cs_reg[rb] = wide_uniform2<T>(this->s_ry[layer_base * N + (N - 1 - G::local_bit(SET, rb == 0 ? 0 : 6 + rb))]);
#pragma unroll
    for (int i = 0; i < NB; ++i) cs_tile[i] = wide_uniform2<T>(this->s_ry[layer_base * N + (N - 1 - G::tile_bit(SET, i))]);
    T acc_common = 0, acc_reg[4] = {0, 0, 0, 0}, acc_tile[NB > 0 ? NB : 1];
#pragma unroll
    for (int i = 0; i < NB; ++i) acc_tile[i] = 0;
    for (uint32_t t = (uint32_t)this->wave; t < (uint32_t)NT; t += (uint32_t)this->waves) {
      C a[R], l[R];
      this->template load_tile<SET>(psi, t, a);
      this->template load_tile<SET>(lam, t, l);
      undo_down_to<SET, 9, 0>(a, l, lcur, th_cur);
      alpha_tile(a, l, t, m);
      undo_diag<SET>(a, l, d, t, signed_token(m) + th_cur[0]);
      // layer 0: d/dtheta_w = Re (RY^dagger lambda)[e_w].  The register bits are contracted level by level: after
      // level k, pl[] holds the plain contraction of bits 0..k and ex[j][] the one with bit j's factor pair replaced
      // by its selector pair (-sin, cos) -- five weighted sums for 82 FMAs and no tables
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
      const T s_plain = pl[0];
      T ft = 1;
#pragma unroll
      for (int i = 0; i < NB; ++i) ft *= ((t >> i) & 1u) ? cs_tile[i].y : cs_tile[i].x;
      acc_common += ft * s_plain;
#pragma unroll
      for (int exi = 0; exi < 4; ++exi) acc_reg[exi] += ft * ex[exi][0];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        T u = 1;
#pragma unroll
        for (int i2 = 0; i2 < NB; ++i2) {
          const bool b = (t >> i2) & 1u;
          u *= i2 == i ? (b ? cs_tile[i2].x : -cs_tile[i2].y) : (b ? cs_tile[i2].y : cs_tile[i2].x);
        }
        acc_tile[i] += u * s_plain;
      }
    }
    put_theta<SET>(row, 0, th_cur, 0);
    put_signed<SET>(row, 16, m);
    // layer 0 thetas -> slots [32, 48)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const T v = group_sum<T, 6>(fl_ex[j] * acc_common, this->lane);
      if (this->lane == 0) row[32 + (N - 1 - G::local_bit(SET, j + 1))] = (double)v;
    }
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      const T v = group_sum<T, 6>(fl * acc_reg[rb], this->lane);
      if (this->lane == 0) row[32 + (N - 1 - G::local_bit(SET, rb == 0 ? 0 : 6 + rb))] = (double)v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const T v = group_sum<T, 6>(fl * acc_tile[i], this->lane);
      if (this->lane == 0) row[32 + (N - 1 - G::tile_bit(SET, i))] = (double)v;
    }
  }
};

// ---------------------------------------------------------------------------------------------------------
// the kernel: one QNode round (L >= 2 layers).  ws: gridDim.x PAIRS of slabs (psi, lambda); partials: gridDim.x slabs
// of `slab_stride` elements, [layer][theta | alpha][16]; grad_inputs (B, gin_ld) or null.
// ---------------------------------------------------------------------------------------------------------
template <typename T, int N>
__global__ __launch_bounds__(4 * kWave) void wide_cz_adjoint_kernel(const T* __restrict__ inputs,
                                                                     const T* __restrict__ tail,
                                                                     const T* __restrict__ gout,
                                                                     T* __restrict__ partials, int64_t slab_stride,
                                                                     T* __restrict__ grad_inputs, int64_t gin_ld,
                                                                     V2<T>* __restrict__ ws, const KScalars p) {
  using W = WideCzAdj<T, N>;
  using C = V2<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int L = p.n_blocks * p.sel_layers;
  C* s_ry = reinterpret_cast<C*>(smem_raw);
  C* s_ua = s_ry + (size_t)L * N;
  C* s_ux = s_ua + (size_t)L * N;
  double* s_acc = reinterpret_cast<double*>(s_ux + 16);   // [L][2][16]
  double* s_wv = s_acc + (size_t)L * 32;                  // [waves][48]
  double* s_gin = s_wv + kWideMaxWaves * 48;              // [16]
  double* s_g = s_gin + 16;                               // [16] upstream gradient of <Z_w>
  double* s_xs = s_g + 16;                                // [16]

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
  w.s_ux = s_ux;
  w.n_layers_round = L;
  for (int i = tid; i < L * 2 * N; i += blockDim.x) {
    const int l = i / (2 * N), e = i - l * 2 * N;
    const C v = C{tail[2 * (size_t)i], tail[2 * (size_t)i + 1]};
    if (e < N) s_ry[l * N + e] = v;
    else s_ua[l * N + (e - N)] = v;
  }
  for (int i = tid; i < L * 32; i += blockDim.x) s_acc[i] = 0.0;
  C* psi = ws + (size_t)blockIdx.x * ((size_t)2 << N);
  C* lam = psi + ((size_t)1 << N);
  double* row = s_wv + w.wave * 48;

  // after a pass: sum the waves' rows in a fixed order into the per-layer accumulators (and the per-sample input
  // gradient when `cur`'s diagonal carried the data re-upload)
  auto combine = [&](int l_cur, bool has_cur, bool cur_block_start, int l_prev) {
    __syncthreads();
    if (tid < 48) {
      double v = 0.0;
      for (int wv = 0; wv < w.waves; ++wv) v += s_wv[wv * 48 + tid];
      const int kind = tid >> 4, wire = tid & 15;
      if (wire < N) {
        if (kind == 0 && has_cur) s_acc[(l_cur * 2 + 0) * 16 + wire] += v;
        if (kind == 1 && has_cur) {
          s_acc[(l_cur * 2 + 1) * 16 + wire] += v;
          if (cur_block_start) s_gin[wire] += v;
        }
        if (kind == 2) s_acc[(l_prev * 2 + 0) * 16 + wire] += v;
      }
    }
    __syncthreads();
  };

  for (int64_t sample = blockIdx.x; sample < p.batch; sample += gridDim.x) {
    const T* g_row = gout + sample * p.g_ld;
    __syncthreads();
    if (tid < 16) {
      s_gin[tid] = 0.0;
      s_g[tid] = (p.measure == 1 && tid < N) ? (double)g_row[tid] : 0.0;
      s_xs[tid] = (p.encoding == 2 && tid < N) ? (double)inputs[sample * p.in_ld + tid] * p.enc_scale : 0.0;
    }
    __syncthreads();
    if (tid < N) {
      double s, c;
      sincos(0.5 * s_xs[tid], &s, &c);
      s_ux[tid] = C{(T)c, (T)s};
    }
    // ---- forward: passes 0 .. L-2 leave the state after pass L-2 in the psi slab ---------------------------------
    for (int li = 0; li + 1 < L; ++li) {
      __syncthreads();
      w.run_pass(psi, 0, li, p, nullptr, nullptr);
    }
    // ---- turnaround (pass L-1) -----------------------------------------------------------------------------------
    __syncthreads();
    for (int i = w.lane; i < 48; i += kWave) row[i] = 0.0;
    if (((L - 1) & 1) == 0) w.template turnaround<0>(psi, lam, L - 1, p, g_row, s_g, row);
    else w.template turnaround<1>(psi, lam, L - 1, p, g_row, s_g, row);
    combine(L - 1, true, false, 0);   // kind 0 only: theta of the last layer (rows 16.. are zero)
    // ---- reverse passes L-2 .. 0 ---------------------------------------------------------------------------------
    for (int pidx = L - 2; pidx >= 0; --pidx) {
      for (int i = w.lane; i < 48; i += kWave) row[i] = 0.0;
      const bool bs = p.encoding == 2 && ((pidx + 1) % p.sel_layers == 0);
      if (pidx == 0) w.reverse_pass0(psi, lam, 0, p, row);
      else if ((pidx & 1) == 0) w.template reverse_pass<0>(psi, lam, 0, pidx, p, row);
      else w.template reverse_pass<1>(psi, lam, 0, pidx, p, row);
      combine(pidx + 1, true, bs, pidx);
    }
    if (grad_inputs != nullptr && tid < N) grad_inputs[sample * gin_ld + tid] = (T)(s_gin[tid] * p.enc_scale);
  }
  __syncthreads();
  T* __restrict__ slab = partials + (size_t)blockIdx.x * slab_stride;
  for (int i = tid; i < L * 32; i += blockDim.x) slab[i] = (T)s_acc[i];
}

}  // namespace qiddm
