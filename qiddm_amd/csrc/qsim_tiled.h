// qsim_tiled.h -- statevector engine for 11 <= n <= 16 qubits on gfx950.
//
// The 2^n-amplitude slab of a sample (64 KiB .. 512 KiB as complex64) no longer fits one
// wavefront's registers.  One WORKGROUP owns a sample; the slab lives in a per-workgroup
// workspace in global memory (it stays L2 / Infinity-Cache resident: the workgroup is the only
// reader and writer) and is processed in PASSES.  In a pass every wave repeatedly
//     loads a TILE of 2^10 amplitudes (10 "local" index bits free, the other n-10 fixed),
//     applies -- in registers, with the same packed-FMA / DPP / permlane gate code as the fused
//     n <= 10 kernel (Engine<T, 10>) -- every pending gate whose target bit is local,
//     stores the tile back.
// Two local-bit sets alternate:  A = bits {0..9},  B = bits {0..9-nb} u {10..n-1}  (nb = n-10);
// both keep the lowest >= 4 index bits local so every access is a full 64-byte segment.
// Gates of one Rot layer commute, so a layer needs exactly one switch A<->B; diagonal steps (the
// per-sample RZ data-encoding, CZ rings) are applied in whichever pass is open; a CNOT ring is a
// GF(2)-linear permutation of the index and is folded into the store addresses of the pass it ends.
// => CZ circuits: one sweep of the slab per SEL layer instead of 2n+ sweeps of the per-gate model.
//
// Replaces the same PennyLane executions as qsim_fused.h for the wide configurations
// (BASELINE configs 4 and 5: 12-qubit QConv2d, 16-qubit qdense).
#pragma once
#include "qsim_fused.h"

namespace qiddm {

constexpr int kTileBits = 10;
constexpr int kTiledMaxQubits = 16;
constexpr int kTiledMaxOps = 4096;
constexpr int kTiledMaxPasses = 768;
constexpr int kTiledWaves = 4;

struct PassHdr {
  uint8_t local[kTileBits];  // global bit position of tile bit 0..9 (ascending)
  uint8_t flags;             // 1: INIT (no load)   2: FINAL (measure instead of store)
  uint8_t cnot_range;        // != 0: store through the CNOT-ring index map of this range
  uint16_t op_begin, op_end;
};
enum : uint32_t { kOpRot = 1, kOpDiagX = 2, kOpCzRing = 3, kOpRyX = 4 };
__device__ __forceinline__ uint32_t make_op(uint32_t type, uint32_t tb, uint32_t arg) {
  return type | (tb << 4) | (arg << 8);
}

// CNOT(i, (i+rr) % n) for i = 0..n-1 in order, as a map of the amplitude index (wire w <-> bit n-1-w)
__device__ __forceinline__ uint32_t cnot_map(uint32_t k, int rr, int n) {
  for (int i = 0; i < n; ++i) {
    int t = i + rr;
    t = t >= n ? t - n : t;
    k ^= ((k >> (n - 1 - i)) & 1u) << (n - 1 - t);
  }
  return k;
}

struct TiledScalars {
  int32_t n;  // qubits (11..16)
  int32_t pad_;
};

// LDS layout of the tiled kernel
template <typename T>
struct TiledSmem {
  static constexpr size_t kProgBytes = kTiledMaxPasses * sizeof(PassHdr) + kTiledMaxOps * 4 + 64;
  static constexpr size_t kMiscBytes = 16 * 3 * sizeof(double) /* xs, cs, sn */ + 16 * sizeof(double) /* result */ +
                                       kTiledWaves * 32 * sizeof(double) /* cross-wave reduce */ + 64;
  __host__ __device__ static size_t gate_bytes(int64_t n_rot) { return (size_t)n_rot * kLdsGateReals * sizeof(T); }
  __host__ __device__ static size_t bytes(int64_t n_rot) { return gate_bytes(n_rot) + kProgBytes + kMiscBytes; }
};

// ---------------------------------------------------------------------------
// pass program: built once per workgroup by one thread (depends on the descriptor only)
// ---------------------------------------------------------------------------
struct ProgramBuilder {
  PassHdr* passes;
  uint32_t* ops;
  int n_pass = 0, n_ops = 0;
  int n, nb;
  int cur = 0;  // 0: set A, 1: set B

  __device__ int tile_bit(int set, int q) const {  // tile bit of global bit q in `set`, or -1
    if (set == 0) return q < kTileBits ? q : -1;
    const int low = kTileBits - nb;  // bits 0..low-1 stay local
    if (q < low) return q;
    if (q >= kTileBits) return low + (q - kTileBits);
    return -1;
  }
  __device__ void open(int set, uint8_t flags) {
    PassHdr& h = passes[n_pass];
    int t = 0;
    for (int q = 0; q < n; ++q)
      if (tile_bit(set, q) >= 0) h.local[t++] = (uint8_t)q;
    h.flags = flags;
    h.cnot_range = 0;
    h.op_begin = (uint16_t)n_ops;
    h.op_end = (uint16_t)n_ops;
    cur = set;
  }
  __device__ void close(int cnot_range) {
    passes[n_pass].op_end = (uint16_t)n_ops;
    passes[n_pass].cnot_range = (uint8_t)cnot_range;
    ++n_pass;
  }
  __device__ void emit(uint32_t op) { ops[n_ops++] = op; }

  // one layer of single-qubit gates on every wire: the wires local to the open pass first, then
  // one switch of the local set for the rest
  __device__ void layer(uint32_t type, int gate0) {
    uint32_t done = 0;
    for (int rep = 0; rep < 2; ++rep) {
      for (int w = 0; w < n; ++w) {
        const int tb = tile_bit(cur, n - 1 - w);
        if (tb >= 0 && !((done >> w) & 1u)) {
          emit(make_op(type, (uint32_t)tb, (uint32_t)(type == kOpRot ? gate0 + w : (w | (gate0 << 8)))));
          done |= 1u << w;
        }
      }
      if (done == (1u << n) - 1u) break;
      close(0);
      open(cur ^ 1, 0);
    }
  }

  __device__ void build(int n_, int encoding, int n_blocks, int sel_layers, bool use_cnot) {
    n = n_;
    nb = n - kTileBits;
    open(0, 1);
    for (int blk = 0; blk < n_blocks; ++blk) {
      if (encoding == 2) emit(make_op(kOpDiagX, 0, (uint32_t)blk));
      if ((encoding == 3 && blk == 0) || encoding == 4) layer(kOpRyX, blk);  // gate0 slot carries the block
      for (int s = 0; s < sel_layers; ++s) {
        layer(kOpRot, (blk * sel_layers + s) * n);
        const int range = (s % (n - 1)) + 1;
        if (!use_cnot) {
          emit(make_op(kOpCzRing, 0, (uint32_t)range));
        } else {
          close(range);
          open(cur, 0);
        }
      }
    }
    passes[n_pass].flags |= 2;
    close(0);
  }
};

// ---------------------------------------------------------------------------
// the kernel.  grid.x strides over samples, grid.y = parameter-shift replica (SHIFT).
// ws: gridDim.x * gridDim.y pairs of slabs of 2^n complex<T>.
// ---------------------------------------------------------------------------
template <typename T, bool SHIFT>
__global__ __launch_bounds__(kTiledWaves* kWave) void tiled_circuit_kernel(
    const T* __restrict__ inputs, const T* __restrict__ table, T* __restrict__ out,
    const T* __restrict__ gout, T* __restrict__ dots, V2<T>* __restrict__ ws, const KScalars p,
    const TiledScalars tp) {
  using E = Engine<T, kTileBits>;
  using C = V2<T>;
  constexpr int R = E::R;  // 16 amplitudes per lane
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int n = tp.n;
  const int nb = n - kTileBits;
  const int n_tiles = 1 << nb;
  const uint32_t dmask = (1u << n) - 1u;
  const int gates_per_round = p.n_blocks * p.sel_layers * n;
  const int n_rot = p.n_rounds * gates_per_round;
  const bool use_cnot = p.imprimitive == 0;

  // ---- LDS carve-up ---------------------------------------------------------------------------
  T* s_gates = reinterpret_cast<T*>(smem_raw);
  unsigned char* cursor = smem_raw + TiledSmem<T>::gate_bytes(n_rot);
  PassHdr* s_pass = reinterpret_cast<PassHdr*>(cursor);
  cursor += kTiledMaxPasses * sizeof(PassHdr);
  uint32_t* s_ops = reinterpret_cast<uint32_t*>(cursor);
  cursor += kTiledMaxOps * 4;
  int* s_counts = reinterpret_cast<int*>(cursor);  // [0] = n_pass
  cursor += 64;
  double* s_xs = reinterpret_cast<double*>(cursor);      // [16] input angles of the current round
  double* s_cs = s_xs + 16;                               // [16] cos(x/2)
  double* s_sn = s_cs + 16;                               // [16] sin(x/2)
  double* s_res = s_sn + 16;                              // [16] <Z_w> of the round / scalars
  double* s_red = s_res + 16;                             // [waves][32] cross-wave reduction

  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = tid >> 6;
  const int llane = logical_lane(lane);

  // ---- parameter-shift replica -----------------------------------------------------------------
  int shift_gate = -1, shift_var = 0, replica_local = 0;
  int sh_blk = -1, sh_wire = 0;
  T sh_sign = 0;
  if constexpr (SHIFT) {
    replica_local = blockIdx.y;
    const int rho = p.first_replica + replica_local;
    if (rho < 6 * n_rot) {
      shift_gate = rho / 6;
      shift_var = 1 + rho % 6;
    } else {
      const int q = rho - 6 * n_rot;
      sh_blk = q / (2 * n);
      sh_wire = (q >> 1) % n;
      sh_sign = (q & 1) ? (T)-1 : (T)1;
    }
  }

  // ---- stage the gate table and build the pass program ------------------------------------------
  E eng;
  eng.s_gates = s_gates;
  eng.lane = lane;
  eng.llane = llane;
  eng.sub = llane;
  for (int g = tid; g < n_rot; g += blockDim.x) {
    const int var = (g == shift_gate) ? shift_var : 0;
    const T* u = table + ((size_t)g * kVariants + var) * kGateReals;
    E::put_gate(s_gates + (size_t)g * kLdsGateReals, u[0], u[1], u[2], u[3], u[4], u[5], u[6], u[7]);
  }
  if (tid == 0) {
    ProgramBuilder pb;
    pb.passes = s_pass;
    pb.ops = s_ops;
    pb.build(n, p.encoding, p.n_blocks, p.sel_layers, use_cnot);
    s_counts[0] = pb.n_pass;
  }
  __syncthreads();
  const int n_pass = s_counts[0];

  // two slabs per workgroup: a pass that ends in a CNOT ring scatters through the index map and must
  // not overwrite amplitudes other tiles have not read yet -> it writes the other slab (ping-pong)
  C* slab0 = ws + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * ((size_t)2 << n);
  C* slab1 = slab0 + ((size_t)1 << n);

  for (int64_t sample = blockIdx.x; sample < p.batch; sample += gridDim.x) {
    const T* __restrict__ in_row = inputs + sample * p.in_ld;
    // ---- round-0 inputs ---------------------------------------------------------------------------
    if (tid < n) s_xs[tid] = p.encoding >= 2 ? (double)in_row[tid] * p.enc_scale : 0.0;
    T amp_inv = 1;
    if (p.encoding == 1) {  // amplitude embedding: norm over features + constant padding
      double part = 0.0;
      for (int k = tid; k < p.n_features; k += blockDim.x) {
        const double v = (double)in_row[k] + p.enc_offset;
        part += v * v;
      }
      part = group_sum<double, 6>(part, lane);
      if (lane == 0) s_red[wave * 32] = part;
      __syncthreads();
      double tot = 0.0;
      for (int w = 0; w < kTiledWaves; ++w) tot += s_red[w * 32];
      tot += p.pad_with * p.pad_with * (double)(((int64_t)1 << n) - p.n_features);
      amp_inv = (T)(1.0 / sqrt(tot));
      __syncthreads();
    }

    C* slab = slab0;      // current state
    C* slab_alt = slab1;  // target of permuting passes
    T dot_acc = 0;  // SHIFT + probs: sum_k g_k p_k of this lane
    for (int round = 0; round < p.n_rounds; ++round) {
      __syncthreads();  // s_xs of this round is complete
      if (tid < n && p.encoding >= 2) {
        double s, c;
        sincos(0.5 * s_xs[tid], &s, &c);
        s_cs[tid] = c;
        s_sn[tid] = s;
      }
      T ez_acc[kTiledMaxQubits];
#pragma unroll
      for (int w = 0; w < kTiledMaxQubits; ++w) ez_acc[w] = 0;

      for (int pi = 0; pi < n_pass; ++pi) {
        __syncthreads();  // previous pass's stores (and s_cs/s_sn) are visible to the whole workgroup
        const PassHdr hdr = s_pass[pi];
        // per-lane and per-register offsets of this pass's local bits
        uint32_t lane_off = 0;
#pragma unroll
        for (int b = 0; b < 6; ++b) lane_off |= (uint32_t)((llane >> b) & 1) << hdr.local[b];
        uint32_t local_mask = 0;
#pragma unroll
        for (int b = 0; b < kTileBits; ++b) local_mask |= 1u << hdr.local[b];
        // CNOT-ring store map F (GF(2)-linear): F(k) = F(base) ^ F(lane part) ^ xor_j F(register bit j)
        uint32_t f_lane = 0, f_reg[4] = {0, 0, 0, 0};
        if (hdr.cnot_range != 0) {
          f_lane = cnot_map(lane_off, hdr.cnot_range, n);
#pragma unroll
          for (int j = 0; j < 4; ++j) f_reg[j] = cnot_map(1u << hdr.local[6 + j], hdr.cnot_range, n);
        }

        for (int tile = wave; tile < n_tiles; tile += kTiledWaves) {
          // deposit the tile number into the non-local bit positions
          uint32_t base = 0;
          {
            int t = tile;
            for (int q = 0; q < n; ++q) {
              if (!((local_mask >> q) & 1u)) {
                base |= (uint32_t)(t & 1) << q;
                t >>= 1;
              }
            }
          }
          uint32_t kidx[R];
#pragma unroll
          for (int r = 0; r < R; ++r) {
            uint32_t ro = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) ro |= (uint32_t)((r >> j) & 1) << hdr.local[6 + j];
            kidx[r] = base | lane_off | ro;
          }
          C a[R];
          if (hdr.flags & 1) {  // INIT
            if (p.encoding == 1) {
#pragma unroll
              for (int r = 0; r < R; ++r) {
                T v = (T)p.pad_with;
                if ((int)kidx[r] < p.n_features) v = in_row[kidx[r]] + (T)p.enc_offset;
                a[r] = C{v * amp_inv, (T)0};
              }
            } else {
#pragma unroll
              for (int r = 0; r < R; ++r) a[r] = C{kidx[r] == 0 ? (T)1 : (T)0, (T)0};
            }
          } else {
#pragma unroll
            for (int r = 0; r < R; ++r) a[r] = slab[kidx[r]];
          }

          // ---- interpret the pass ---------------------------------------------------------------
          for (int oi = hdr.op_begin; oi < hdr.op_end; ++oi) {
            const uint32_t op = s_ops[oi];
            const uint32_t type = op & 15u, tb = (op >> 4) & 15u, arg = op >> 8;
            if (type == kOpRot || type == kOpRyX) {
              C m[8];
              if (type == kOpRot) {
                const C* gp = reinterpret_cast<const C*>(
                    s_gates + (size_t)(arg + round * gates_per_round) * kLdsGateReals);
                if (tb < 4) {
                  const C* hp = gp + (((llane >> tb) & 1) << 2);
#pragma unroll
                  for (int i = 0; i < 4; ++i) m[i] = hp[i];
                } else {
#pragma unroll
                  for (int i = 0; i < 8; ++i) m[i] = gp[i];
                }
              } else {
                const int ry_wire = (int)(arg & 255u), ry_blk = (int)(arg >> 8);
                T c = (T)s_cs[ry_wire], s = (T)s_sn[ry_wire];
                const T z = 0;
                if constexpr (SHIFT) {
                  if (ry_blk == sh_blk && ry_wire == sh_wire) {  // shifted RY input angle: +-pi/4 on the half angle
                    const T h = (T)0.70710678118654752440;
                    const T c2 = h * (c - sh_sign * s), s2 = h * (s + sh_sign * c);
                    c = c2;
                    s = s2;
                  }
                }
                if (tb < 4) {
                  const T sp = ((llane >> tb) & 1) ? s : -s;
                  m[0] = C{c, z}; m[1] = C{z, c}; m[2] = C{sp, z}; m[3] = C{z, sp};
                } else {
                  m[0] = C{c, z}; m[1] = C{z, c}; m[2] = C{-s, z}; m[3] = C{z, -s};
                  m[4] = C{c, z}; m[5] = C{z, c}; m[6] = C{s, z};  m[7] = C{z, s};
                }
              }
              switch (tb) {
                case 0: eng.template gate_lane<0>(a, m); break;
                case 1: eng.template gate_lane<1>(a, m); break;
                case 2: eng.template gate_lane<2>(a, m); break;
                case 3: eng.template gate_lane<3>(a, m); break;
                case 4:
                  eng.template swap_reg0_with_lane_bit<4>(a);
                  eng.template gate_regs<1>(a, m, m + 4);
                  eng.template swap_reg0_with_lane_bit<4>(a);
                  break;
                case 5:
                  eng.template swap_reg0_with_lane_bit<5>(a);
                  eng.template gate_regs<1>(a, m, m + 4);
                  eng.template swap_reg0_with_lane_bit<5>(a);
                  break;
                case 6: eng.template gate_regs<1>(a, m, m + 4); break;
                case 7: eng.template gate_regs<2>(a, m, m + 4); break;
                case 8: eng.template gate_regs<4>(a, m, m + 4); break;
                default: eng.template gate_regs<8>(a, m, m + 4); break;
              }
            } else if (type == kOpDiagX) {
              // per-sample RZ data encoding: phase(k) = prod_w exp(i (2 b_w - 1) x_w / 2); the lane- and
              // tile-dependent factors are folded first, then one complex multiply per amplitude and
              // register bit
              T fr = 1, fi = 0;
              for (int q = 0; q < n; ++q) {
                const bool is_reg = (q == hdr.local[6]) | (q == hdr.local[7]) | (q == hdr.local[8]) |
                                    (q == hdr.local[9]);
                if (is_reg) continue;
                const T c = (T)s_cs[n - 1 - q];
                T si = (T)s_sn[n - 1 - q];
                si = ((base | lane_off) >> q) & 1u ? si : -si;
                const T nr = fr * c - fi * si;
                fi = fr * si + fi * c;
                fr = nr;
              }
              C ph[R];
              ph[0] = C{fr, fi};
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const int w = n - 1 - hdr.local[6 + j];
                const T c = (T)s_cs[w], s = (T)s_sn[w];
#pragma unroll
                for (int r = 0; r < (1 << j); ++r) {
                  const C d = ph[r];
                  ph[r | (1 << j)] = C{d.x * c - d.y * s, d.x * s + d.y * c};
                  ph[r] = C{d.x * c + d.y * s, d.y * c - d.x * s};
                }
              }
#pragma unroll
              for (int r = 0; r < R; ++r) a[r] = cmul2<T>(ph[r], a[r], times_i<T>(a[r]));
              if constexpr (SHIFT) {
                if ((int)arg == sh_blk) {  // extra RZ(+-pi/2) on sh_wire
                  const int q = n - 1 - sh_wire;
                  const T h = (T)0.70710678118654752440;
#pragma unroll
                  for (int r = 0; r < R; ++r) {
                    const T si = ((kidx[r] >> q) & 1u) ? h * sh_sign : -h * sh_sign;
                    a[r] = C{a[r].x * h - a[r].y * si, a[r].x * si + a[r].y * h};
                  }
                }
              }
            } else {  // kOpCzRing
              const int rr = (int)arg;
#pragma unroll
              for (int r = 0; r < R; ++r) {
                const uint32_t k = kidx[r];
                const uint32_t rot = ((k << rr) | (k >> (n - rr))) & dmask;
                const uint32_t sb = (uint32_t)(__popc(k & rot) & 1) << 31;
                a[r] = C{flip_sign(a[r].x, sb), flip_sign(a[r].y, sb)};
              }
            }
          }

          // ---- store / measure ----------------------------------------------------------------------
          if (!(hdr.flags & 2)) {
            if (hdr.cnot_range == 0) {
#pragma unroll
              for (int r = 0; r < R; ++r) slab[kidx[r]] = a[r];
            } else {
              const uint32_t f_tile = cnot_map(base, hdr.cnot_range, n) ^ f_lane;
#pragma unroll
              for (int r = 0; r < R; ++r) {
                uint32_t k = f_tile;
#pragma unroll
                for (int j = 0; j < 4; ++j) k ^= ((r >> j) & 1) ? f_reg[j] : 0u;
                slab_alt[k] = a[r];
              }
            }
          } else {
            T pr[R];
#pragma unroll
            for (int r = 0; r < R; ++r) pr[r] = a[r].x * a[r].x + a[r].y * a[r].y;
            if (p.measure == 0) {
              if constexpr (SHIFT) {
#pragma unroll
                for (int r = 0; r < R; ++r) dot_acc += gout[sample * p.g_ld + kidx[r]] * pr[r];
              } else {
#pragma unroll
                for (int r = 0; r < R; ++r) out[sample * p.out_ld + kidx[r]] = pr[r];
              }
            } else {
#pragma unroll
              for (int w = 0; w < kTiledMaxQubits; ++w) {
                if (w < n) {
                  const int q = n - 1 - w;
                  T acc = 0;
#pragma unroll
                  for (int r = 0; r < R; ++r) acc += ((kidx[r] >> q) & 1u) ? -pr[r] : pr[r];
                  ez_acc[w] += acc;
                }
              }
            }
          }
        }  // tiles
        if (hdr.cnot_range != 0 && !(hdr.flags & 2)) {  // the permuted state now lives in the other slab
          C* t = slab;
          slab = slab_alt;
          slab_alt = t;
        }
      }    // passes

      // ---- finish the round's measurement ----------------------------------------------------------
      if (p.measure == 1) {
        __syncthreads();
#pragma unroll
        for (int w = 0; w < kTiledMaxQubits; ++w) {
          const T v = group_sum<T, 6>(ez_acc[w], lane);
          if (lane == 0) s_red[wave * 32 + w] = (double)v;
        }
        __syncthreads();
        if (tid < n) {
          double tot = 0.0;
          for (int w = 0; w < kTiledWaves; ++w) tot += s_red[w * 32 + tid];
          s_res[tid] = tot;
        }
        __syncthreads();
      }
      if (round + 1 < p.n_rounds) {  // chain: x <- out[:, 0:n]
        __syncthreads();
        if (tid < n) {
          const double v = p.measure == 1 ? s_res[tid] : (double)out[sample * p.out_ld + tid];
          s_xs[tid] = v * p.enc_scale;
        }
      }
    }  // rounds

    // ---- epilogue -------------------------------------------------------------------------------------
    if constexpr (!SHIFT) {
      if (p.measure == 1 && tid < n) out[sample * p.out_ld + tid] = (T)s_res[tid];
    } else {
      double tot = 0.0;
      if (p.measure == 0) {
        const T v = group_sum<T, 6>(dot_acc, lane);
        __syncthreads();
        if (lane == 0) s_red[wave * 32] = (double)v;
        __syncthreads();
        for (int w = 0; w < kTiledWaves; ++w) tot += s_red[w * 32];
      } else {
        for (int w = 0; w < n; ++w) tot += (double)gout[sample * p.g_ld + w] * s_res[w];
      }
      if (tid == 0) dots[(int64_t)replica_local * p.batch + sample] = (T)tot;
    }
    __syncthreads();
  }  // samples
}

}  // namespace qiddm
