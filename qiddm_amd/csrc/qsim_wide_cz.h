// qsim_wide_cz.h -- forward of the CZ-entangled circuit family for 11 <= n <= 16 qubits on gfx950.
//
// Replaces the PennyLane executions of the re-uploading RZ / StronglyEntanglingLayers(CZ) circuits
// (reference nn/qdense.py:249-265, 1403-1421, 1599-1617) at the wide configurations: BASELINE config 5,
// QIDDM_LL-style (2352, 16, 6, 2) -- 65 536 amplitudes, 512 KiB of complex64 per sample.  This is the one
// configuration of the path whose statevector cannot live on chip, so every sweep of the slab is physical
// memory traffic and the HBM roofline is the bound that matters.
//
// What the generic tiled engine (qsim_tiled.h) spends and this kernel does not:
//   * general complex 2x2 gates.  Rot = RZ(omega) RY(theta) RZ(phi) and the CZ ring is diagonal, so a layer is
//       [diagonal D^l] then [real RY on every wire];   D^l = RZ(phi^l) . CZring^{l-1} . RZ(omega^{l-1}) . RZ(x)
//     (the data re-upload RZ(x) only at a block start).  The diagonal is ONE complex multiply per amplitude by a
//     phase that factors over the index bits: (per-lane part) x (per-register part) x (per-tile part), built
//     once per pass; a real RY costs 2 packed ops per amplitude instead of 4, and its coefficients are
//     wave-uniform scalars.
//   * an interpreted pass program.  The pass structure is compile time (template on n and on the local-bit
//     set), every gate position is a template parameter.
//   * the first sweep of every round: layer 0 acts on |0..0>, the result is a real product state and is
//     GENERATED in registers by the first pass (no load, and the round needs L - 1 sweeps for L layers).
//   * exposed memory latency.  Tiles are double-buffered in registers (the loads of tile t+1 are in flight while
//     tile t is computed), every access is a 16-byte-per-lane vector access (two consecutive amplitudes per lane),
//     and the workgroup is 8 waves at n >= 14.
//
// Layout.  One workgroup owns a sample; its slab of 2^n complex amplitudes lives in a per-workgroup workspace.
// A pass sweeps the slab in TILES of 2^10 amplitudes: 10 "local" index bits are free inside a tile, the other
// n - 10 select the tile.  Two local-bit sets alternate (NB = n - 10, P = 10 - NB):
//     set A: local = bits 0..9                      tile = bits 10..n-1
//     set B: local = bits 0..P-1 and 10..n-1        tile = bits P..9
// In BOTH sets the bits that the other set does not have are local positions P..9, so a pass is always
//     [finish layer l on positions P..9] . D^{l+1} . [layer l+1 on positions 0..9]  -> one sweep per layer.
// Inside a tile, local position 0 is register bit 0 (the two amplitudes of a 16-byte access), positions 1..6 are
// the six lane bits (logical lane numbering of qsim_fused.h: DPP / permlane partners), positions 7..9 are register
// bits 1..3: 16 amplitudes per lane.
#pragma once
#include "qsim_fused.h"

namespace qiddm {

constexpr int kWideMaxWaves = 8;

template <int N>
struct WideGeom {
  static constexpr int NB = N - 10;
  static constexpr int NT = 1 << NB;
  static constexpr int P = 10 - NB;
  __host__ __device__ static constexpr int local_bit(int set, int j) { return (set == 0 || j < P) ? j : j + NB; }
  __host__ __device__ static constexpr int tile_bit(int set, int i) { return set == 0 ? 10 + i : P + i; }
  // element index of (tile t, 10-bit local index u)
  __host__ __device__ static constexpr uint32_t index(int set, uint32_t t, uint32_t u) {
    return set == 0 ? ((t << 10) | u) : ((u & ((1u << P) - 1u)) | (t << P) | ((u >> P) << 10));
  }
};

template <typename T>
struct WideCzSmem {
  // [ry: L*n C][ua: L*n C][ux: 16 C][res: 16 double][red: waves*16 double][xs: 16 double]
  __host__ __device__ static size_t bytes(int64_t layers_all, int n) {
    return (size_t)layers_all * n * 2 * 2 * sizeof(T) + 16 * 2 * sizeof(T) + (16 + kWideMaxWaves * 16 + 16) * sizeof(double);
  }
};

template <typename T>
__device__ __forceinline__ T wide_uniform(T v) {
  if constexpr (sizeof(T) == 4) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
  } else {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
  }
}
template <typename T>
__device__ __forceinline__ V2<T> wide_uniform2(V2<T> v) {
  return V2<T>{wide_uniform<T>(v.x), wide_uniform<T>(v.y)};
}
template <typename T>
__device__ __forceinline__ V2<T> wide_cmul(V2<T> a, V2<T> b) {
  return cmul2<T>(a, b, times_i<T>(b));
}
template <typename T>
__device__ __forceinline__ V2<T> wide_sel(uint32_t bit, V2<T> u) {  // bit ? u : conj(u)
  return V2<T>{u.x, bit ? u.y : -u.y};
}

template <typename T, int N>
struct WideCz {
  using G = WideGeom<N>;
  using C = V2<T>;
  using E = Engine<T, 10>;
  static constexpr int R = 16, NB = G::NB, NT = G::NT, P = G::P;
  struct __attribute__((aligned(2 * sizeof(V2<T>)))) Pair {
    C lo, hi;
  };

  E eng;
  int lane, llane, wave, waves;
  const C* s_ry;
  const C* s_ua;
  const C* s_ux;
  int n_layers_round;

  // ---- addressing ------------------------------------------------------------------------------------
  template <int SET>
  __device__ __forceinline__ uint32_t pair_index(uint32_t t, int r1) const {  // element index of a[2*r1]
    const uint32_t u = ((uint32_t)llane << 1) | ((uint32_t)r1 << 7);
    return G::index(SET, t, u);
  }
  template <int SET>
  __device__ __forceinline__ void load_tile(const C* __restrict__ slab, uint32_t t, C (&a)[R]) const {
#pragma unroll
    for (int r1 = 0; r1 < R / 2; ++r1) {
      const Pair v = *reinterpret_cast<const Pair*>(slab + pair_index<SET>(t, r1));
      a[2 * r1] = v.lo;
      a[2 * r1 + 1] = v.hi;
    }
  }
  template <int SET>
  __device__ __forceinline__ void store_tile(C* __restrict__ slab, uint32_t t, const C (&a)[R]) const {
#pragma unroll
    for (int r1 = 0; r1 < R / 2; ++r1) {
      Pair v;
      v.lo = a[2 * r1];
      v.hi = a[2 * r1 + 1];
      *reinterpret_cast<Pair*>(slab + pair_index<SET>(t, r1)) = v;
    }
  }

  // ---- real RY on local position POS with wave-uniform (c, s) --------------------------------------------
  template <int POS>
  __device__ __forceinline__ void ry_pos(C (&a)[R], T c, T s) const {
    if constexpr (POS == 0) {
      eng.template ry_pairs<1>(a, c, s);
    } else if constexpr (POS <= 4) {
      constexpr int LBIT = POS - 1;
      const T sg = ((llane >> LBIT) & 1) ? s : -s;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const C par = xlane2<(1 << LBIT), T>(a[r], lane);
        a[r] = __builtin_elementwise_fma(bcast<T>(sg), par, bcast<T>(c) * a[r]);
      }
    } else if constexpr (POS <= 6) {
      constexpr int LBIT = POS - 1;  // 4 or 5: exchange with register bit 0, rotate register-local, exchange back
      eng.template swap_reg0_with_lane_bit<LBIT>(a);
      eng.template ry_pairs<1>(a, c, s);
      eng.template swap_reg0_with_lane_bit<LBIT>(a);
    } else {
      eng.template ry_pairs<(1 << (POS - 6))>(a, c, s);
    }
  }
  template <int SET, int POS>
  __device__ __forceinline__ C ry_coeff(int layer) const {
    constexpr int w = N - 1 - G::local_bit(SET, POS);
    return wide_uniform2<T>(s_ry[layer * N + w]);
  }
  // RY layer `layer` on local positions [FROM, 10)
  template <int SET, int POS>
  __device__ __forceinline__ void ry_from(C (&a)[R], int layer) const {
    if constexpr (POS < 10) {
      const C cs = ry_coeff<SET, POS>(layer);
      ry_pos<POS>(a, cs.x, cs.y);
      ry_from<SET, POS + 1>(a, layer);
    }
  }

  // ---- the diagonal of layer `ld` (1 <= position of ld inside its round) ---------------------------------
  struct Diag {
    C pl;            // lane part of the phase
    C pr[R];         // register part (wave-uniform: lives in scalar registers for float)
    C ut[NB > 0 ? NB : 1];   // unit phases of the tile bits (wave-uniform)
    uint32_t kl, rl;  // this lane's index bits and their ring rotation
    uint32_t rr_reg[R];  // ring rotation of the register bits' index contribution (wave-uniform)
    int range;
  };
  __device__ __forceinline__ C unit_phase(int ld, int w, bool block_start) const {
    C u = s_ua[ld * N + w];
    if (block_start) u = wide_cmul<T>(u, s_ux[w]);
    return u;
  }
  __device__ __forceinline__ static uint32_t rotl_n(uint32_t k, int rr) {
    return ((k << rr) | (k >> (N - rr))) & ((1u << N) - 1u);
  }
  template <int SET>
  __device__ __forceinline__ static constexpr uint32_t reg_index_bits(int r) {
    return ((uint32_t)(r & 1) << G::local_bit(SET, 0)) | ((uint32_t)((r >> 1) & 1) << G::local_bit(SET, 7)) |
           ((uint32_t)((r >> 2) & 1) << G::local_bit(SET, 8)) | ((uint32_t)((r >> 3) & 1) << G::local_bit(SET, 9));
  }
  template <int SET>
  __device__ __forceinline__ void build_diag(int ld, bool block_start, int range, Diag& d) const {
    // lane part
    C pl = C{(T)1, (T)0};
#pragma unroll
    for (int j = 1; j <= 6; ++j) {
      const int w = N - 1 - G::local_bit(SET, j);
      pl = wide_cmul<T>(pl, wide_sel<T>((llane >> (j - 1)) & 1, unit_phase(ld, w, block_start)));
    }
    // register part (positions 0, 7, 8, 9 <-> register bits 0..3), by doubling
    C pr[R];
    pr[0] = C{(T)1, (T)0};
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      const int pos = rb == 0 ? 0 : 6 + rb;
      const C u = wide_uniform2<T>(unit_phase(ld, N - 1 - G::local_bit(SET, pos), block_start));
      const C uc = C{u.x, -u.y};
#pragma unroll
      for (int r = 0; r < (1 << rb); ++r) {
        pr[r | (1 << rb)] = wide_cmul<T>(pr[r], u);
        pr[r] = wide_cmul<T>(pr[r], uc);
      }
    }
    d.pl = pl;
#pragma unroll
    for (int r = 0; r < R; ++r) d.pr[r] = wide_uniform2<T>(pr[r]);
#pragma unroll
    for (int i = 0; i < NB; ++i)
      d.ut[i] = wide_uniform2<T>(unit_phase(ld, N - 1 - G::tile_bit(SET, i), block_start));
    // index bits of this lane (positions 1..6) and ring rotations
    uint32_t kl = 0;
#pragma unroll
    for (int j = 1; j <= 6; ++j) kl |= (uint32_t)((llane >> (j - 1)) & 1) << G::local_bit(SET, j);
    d.kl = kl;
    d.rl = rotl_n(kl, range);
#pragma unroll
    for (int r = 0; r < R; ++r) d.rr_reg[r] = rotl_n(reg_index_bits<SET>(r), range);
    d.range = range;
  }
  template <int SET>
  __device__ __forceinline__ static uint32_t tile_index_bits(uint32_t t) {
    return t << G::tile_bit(SET, 0);  // the tile bits are contiguous in both sets
  }
  template <int SET>
  __device__ __forceinline__ void apply_diag(C (&a)[R], const Diag& d, uint32_t t) const {
    C tt = C{(T)1, (T)0};
#pragma unroll
    for (int i = 0; i < NB; ++i) tt = wide_cmul<T>(tt, wide_sel<T>((t >> i) & 1u, d.ut[i]));
    tt = wide_cmul<T>(tt, d.pl);        // tile part x lane part
    const uint32_t kt = tile_index_bits<SET>(t);
    const uint32_t x = kt | d.kl;
    const uint32_t y = rotl_n(kt, d.range) | d.rl;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      C ph = wide_cmul<T>(tt, d.pr[r]);
      // CZ ring of the previous layer: sign = parity(popc(k & rotl(k, range)))
      const uint32_t sb = (uint32_t)(__popc((x | reg_index_bits<SET>(r)) & (y | d.rr_reg[r])) & 1) << 31;
      ph = C{flip_sign(ph.x, sb), flip_sign(ph.y, sb)};
      a[r] = wide_cmul<T>(ph, a[r]);
    }
  }

  // ---- the product state of a round's first layer, generated in registers ---------------------------------
  struct Init {
    T fl;        // lane part
    T fr[R];     // register part
    C ft[NB > 0 ? NB : 1];   // (cos, sin) of the tile bits' wires (wave-uniform)
  };
  template <int SET>
  __device__ __forceinline__ void build_init(int layer0, Init& in) const {
    T fl = (T)1;
#pragma unroll
    for (int j = 1; j <= 6; ++j) {
      const C cs = s_ry[layer0 * N + (N - 1 - G::local_bit(SET, j))];
      fl *= ((llane >> (j - 1)) & 1) ? cs.y : cs.x;
    }
    in.fl = fl;
    T fr[R];
    fr[0] = (T)1;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      const int pos = rb == 0 ? 0 : 6 + rb;
      const C cs = wide_uniform2<T>(s_ry[layer0 * N + (N - 1 - G::local_bit(SET, pos))]);
#pragma unroll
      for (int r = 0; r < (1 << rb); ++r) {
        fr[r | (1 << rb)] = fr[r] * cs.y;
        fr[r] = fr[r] * cs.x;
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) in.fr[r] = fr[r] * fl;
#pragma unroll
    for (int i = 0; i < NB; ++i) in.ft[i] = wide_uniform2<T>(s_ry[layer0 * N + (N - 1 - G::tile_bit(SET, i))]);
  }
  __device__ __forceinline__ void gen_tile(const Init& in, uint32_t t, C (&a)[R]) const {
    T ft = (T)1;
#pragma unroll
    for (int i = 0; i < NB; ++i) ft *= ((t >> i) & 1u) ? in.ft[i].y : in.ft[i].x;
#pragma unroll
    for (int r = 0; r < R; ++r) a[r] = C{in.fr[r] * ft, (T)0};
  }

  // ---- measurement accumulators ------------------------------------------------------------------------
  struct Meas {
    T tot = 0;
    T reg[4] = {0, 0, 0, 0};
    T tile[NB > 0 ? NB : 1];
  };
  template <int SET>
  __device__ __forceinline__ void measure_tile(const C (&a)[R], uint32_t t, int measure, T* __restrict__ out_row,
                                               Meas& m) const {
    T pr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) pr[r] = a[r].x * a[r].x + a[r].y * a[r].y;
    if (measure == 0) {
#pragma unroll
      for (int r1 = 0; r1 < R / 2; ++r1) {
        using T2 = V2<T>;
        *reinterpret_cast<T2*>(out_row + pair_index<SET>(t, r1)) = T2{pr[2 * r1], pr[2 * r1 + 1]};
      }
    } else {
      T tot = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) tot += pr[r];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        T sgn = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) sgn += ((r >> rb) & 1) ? -pr[r] : pr[r];
        m.reg[rb] += sgn;
      }
      m.tot += tot;
#pragma unroll
      for (int i = 0; i < NB; ++i) m.tile[i] += ((t >> i) & 1u) ? -tot : tot;
    }
  }
  // <Z_w> partial sums of this wave -> s_red[wave][w]
  template <int SET>
  __device__ __forceinline__ void reduce_measure(const Meas& m, double* __restrict__ s_red) const {
    double* row = s_red + wave * 16;
#pragma unroll
    for (int j = 1; j <= 6; ++j) {
      const T v = group_sum<T, 6>(((llane >> (j - 1)) & 1) ? -m.tot : m.tot, lane);
      if (lane == 0) row[N - 1 - G::local_bit(SET, j)] = (double)v;
    }
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      const int pos = rb == 0 ? 0 : 6 + rb;
      const T v = group_sum<T, 6>(m.reg[rb], lane);
      if (lane == 0) row[N - 1 - G::local_bit(SET, pos)] = (double)v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const T v = group_sum<T, 6>(m.tile[i], lane);
      if (lane == 0) row[N - 1 - G::tile_bit(SET, i)] = (double)v;
    }
  }

  // ---- one pass over the slab -----------------------------------------------------------------------------
  template <int SET, bool INIT, bool LAST>
  __device__ __forceinline__ void pass(C* __restrict__ slab, int layer_base, int li, const KScalars& p,
                                       T* __restrict__ out_row, double* __restrict__ s_red) const {
    const int lnext = layer_base + li + 1;
    Diag d;
    Init in;
    Meas m;
#pragma unroll
    for (int i = 0; i < NB; ++i) m.tile[i] = 0;
    if constexpr (INIT) build_init<SET>(layer_base, in);
    if constexpr (!LAST) {
      const int s_prev = li % p.sel_layers;                     // SEL position of layer li: its ring precedes D^{li+1}
      build_diag<SET>(lnext, p.encoding == 2 && ((li + 1) % p.sel_layers == 0), (s_prev % (N - 1)) + 1, d);
    }
    // complex64: the next tile's loads are in flight while this one is computed (register double buffer);
    // complex128 (the parity precision) has no registers to spare for that
    constexpr bool PREFETCH = sizeof(T) == 4 && !INIT;
    C a[R], nxt[PREFETCH ? R : 1];
    uint32_t t = (uint32_t)wave;
    if constexpr (PREFETCH) {
      if (t < (uint32_t)NT) load_tile<SET>(slab, t, nxt);
    }
    for (; t < (uint32_t)NT; t += (uint32_t)waves) {
      if constexpr (INIT) {
        gen_tile(in, t, a);
      } else {
        if constexpr (PREFETCH) {
#pragma unroll
          for (int r = 0; r < R; ++r) a[r] = nxt[r];
          if (t + (uint32_t)waves < (uint32_t)NT) load_tile<SET>(slab, t + (uint32_t)waves, nxt);
        } else {
          load_tile<SET>(slab, t, a);
        }
        ry_from<SET, P>(a, layer_base + li);
      }
      if constexpr (!LAST) {
        apply_diag<SET>(a, d, t);
        ry_from<SET, 0>(a, lnext);
        store_tile<SET>(slab, t, a);
      } else {
        measure_tile<SET>(a, t, p.measure, out_row, m);
      }
    }
    if constexpr (LAST) {
      if (p.measure == 1) reduce_measure<SET>(m, s_red);
    }
  }
  // li: index (inside the round) of the layer this pass FINISHES on positions P..9 (li == 0: the pass generates
  // layer 0 instead of loading); unless it is the round's last layer the pass goes on with D^{li+1} and layer li+1.
  __device__ __forceinline__ void run_pass(C* __restrict__ slab, int layer_base, int li, const KScalars& p,
                                           T* __restrict__ out_row, double* __restrict__ s_red) const {
    const bool last = li == n_layers_round - 1;
    if (li == 0) {
      if (last) pass<0, true, true>(slab, layer_base, li, p, out_row, s_red);
      else pass<0, true, false>(slab, layer_base, li, p, out_row, s_red);
    } else if ((li & 1) == 0) {
      if (last) pass<0, false, true>(slab, layer_base, li, p, out_row, s_red);
      else pass<0, false, false>(slab, layer_base, li, p, out_row, s_red);
    } else {
      if (last) pass<1, false, true>(slab, layer_base, li, p, out_row, s_red);
      else pass<1, false, false>(slab, layer_base, li, p, out_row, s_red);
    }
  }
};

// ---------------------------------------------------------------------------------------------------------
// the kernel.  grid.x strides over samples; ws: gridDim.x slabs of 2^N complex<T>; tail: the per-layer tables
// (wide_fold_entry) of all n_rounds * n_blocks * sel_layers layers.
// ---------------------------------------------------------------------------------------------------------
template <typename T, int N>
__global__ __launch_bounds__(kWideMaxWaves* kWave) void wide_cz_kernel(const T* __restrict__ inputs,
                                                                        const T* __restrict__ tail,
                                                                        T* __restrict__ out, V2<T>* __restrict__ ws,
                                                                        const KScalars p) {
  using W = WideCz<T, N>;
  using C = V2<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int layers_round = p.n_blocks * p.sel_layers;
  const int layers_all = p.n_rounds * layers_round;
  C* s_ry = reinterpret_cast<C*>(smem_raw);
  C* s_ua = s_ry + (size_t)layers_all * N;
  C* s_ux = s_ua + (size_t)layers_all * N;
  double* s_res = reinterpret_cast<double*>(s_ux + 16);
  double* s_red = s_res + 16;
  double* s_xs = s_red + kWideMaxWaves * 16;

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
  w.n_layers_round = layers_round;

  // per-layer tables: [layer][2N] complex, first N = ry, next N = ua
  for (int i = tid; i < layers_all * 2 * N; i += blockDim.x) {
    const int l = i / (2 * N), e = i - l * 2 * N;
    const C v = C{tail[2 * (size_t)i], tail[2 * (size_t)i + 1]};
    if (e < N) s_ry[l * N + e] = v;
    else s_ua[l * N + (e - N)] = v;
  }
  C* slab = ws + (size_t)blockIdx.x * ((size_t)1 << N);

  for (int64_t sample = blockIdx.x; sample < p.batch; sample += gridDim.x) {
    T* out_row = out + sample * p.out_ld;
    if (tid < N) s_xs[tid] = p.encoding == 2 ? (double)inputs[sample * p.in_ld + tid] * p.enc_scale : 0.0;
    for (int round = 0; round < p.n_rounds; ++round) {
      __syncthreads();  // s_xs of this round; the previous round's / sample's slab traffic has drained
      if (tid < N) {
        double s, c;
        sincos(0.5 * s_xs[tid], &s, &c);
        s_ux[tid] = C{(T)c, (T)s};
      }
      const int layer_base = round * layers_round;
      for (int li = 0; li < layers_round; ++li) {
        __syncthreads();  // the previous pass's stores (and s_ux) are visible to the whole workgroup
        w.run_pass(slab, layer_base, li, p, out_row, s_red);
      }
      // ---- finish the round's measurement; chain x <- out[:, 0:n] --------------------------------------
      __syncthreads();
      if (p.measure == 1) {
        if (tid < N) {
          double tot = 0.0;
          for (int wv = 0; wv < w.waves; ++wv) tot += s_red[wv * 16 + tid];
          s_res[tid] = tot;
          if (round + 1 < p.n_rounds) s_xs[tid] = tot * p.enc_scale;
          else out_row[tid] = (T)tot;
        }
      } else if (round + 1 < p.n_rounds) {
        if (tid < N) s_xs[tid] = (double)out_row[tid] * p.enc_scale;
      }
    }
    __syncthreads();
  }
}

}  // namespace qiddm
