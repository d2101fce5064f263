// ubench_layer.hip -- the layer of dense_quad8_kernel (qsim_lean.h) taken apart: the same instruction sequence with
// one ingredient removed or changed per variant, 13 layers per "step", s_memtime around `iters` steps, one 256-thread
// workgroup per CU on all 256 CUs.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../qiddm_amd/csrc -o ubench_layer ubench_layer.hip && ./ubench_layer
#include "qsim_lean.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace qiddm;
typedef float v4f __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

enum { kFull = 0, kNoFetch, kNoFetchNoBarrier, kNoExchange, kNoSwaps, kNoDpp, kNoPhase, kFetchAfterBarrier, kOnlyExchange, kRuntimeLoop, kRuntimeLoopBranch,
       kCount };
static const char* kNames[] = {
    "full layer as shipped (fetch at top, derive after swap 5)",
    "coefficients fixed in registers (no table fetch / derive)",
    "  ... and no s_barrier (racy values): barrier + skew",
    "  ... and no exchange at all (in-wave chain alone: phase, 4 dpp, 2 swaps)",
    "  ... full but without the two permlane RYs",
    "  ... full but without the four DPP RYs",
    "  ... full but without the phase multiply",
    "full layer, table fetch placed behind the partner reads instead of the top",
    "exchange alone (write, barrier, 3 reads, 4 pk)",
    "full layer, layer count a kernel argument (loop by two, not unrolled)",
    "  ... plus the never-taken block-start test of the nets without re-upload"};

template <int VAR>
__global__ __launch_bounds__(256) void layer_loop(float* out, unsigned long long* ticks, int iters, const float* tab,
                                                  int n_layers, int next_upload_arg) {
  using T = float;
  using C = V2<T>;
  constexpr int LAYERS = 13;
  __shared__ C s_ph[LAYERS * 256];
  __shared__ __attribute__((aligned(16))) T s_un[LAYERS * 8];
  __shared__ C s_slab[2 * 4 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int i = tid; i < LAYERS * 256; i += 256) s_ph[i] = C{tab[(i * 2) % 509] * 0.9f + 0.05f, tab[(i * 2 + 1) % 509] * 0.3f};
  for (int i = tid; i < LAYERS * 8; i += 256) s_un[i] = tab[i % 509] * 0.4f;
  T pm[8];
  const uint32_t kbase = ((uint32_t)wv << 6) | (uint32_t)logical_lane(lane);
#pragma unroll
  for (int q = 0; q < 8; ++q) pm[q] = ((kbase >> q) & 1u) ? (T)1 : (T)-1;
  C a{0.01f * (float)(tid + 1), 0.02f};
  int par = 0;
  struct Raw { C ph; v4f lo, hi; };
  auto fetch = [&](Raw& r, int l) {
    r.ph = s_ph[l * 256 + tid];
    const v4f* u = reinterpret_cast<const v4f*>(s_un + l * 8);
    r.lo = u[0];
    r.hi = u[1];
  };
  auto derive = [&](LeanLayer<T>& c, const Raw& r) {
    c.ph = r.ph;
    c.ts[0] = r.lo.x * pm[0]; c.ts[1] = r.lo.y * pm[1]; c.ts[2] = r.lo.z * pm[2]; c.ts[3] = r.lo.w * pm[3];
    c.t4 = r.hi.x; c.t5 = r.hi.y;
    c.k1 = r.hi.z * pm[6]; c.k2 = r.hi.w * pm[7]; c.k3 = c.k1 * c.k2;
  };
  LeanLayer<T> ca, cb;
  Raw raw;
  fetch(raw, 0);
  derive(ca, raw);
  cb = ca;
  __syncthreads();
  int next_upload = next_upload_arg;
  auto layer = [&](const LeanLayer<T>& cur, LeanLayer<T>& nxt, int li) {
    constexpr bool tables = VAR == kFull || VAR == kFetchAfterBarrier || VAR >= kRuntimeLoop;
    constexpr bool full = VAR == kFull || VAR >= kRuntimeLoop;
    if constexpr (full) {
      fetch(raw, li + 1 < (VAR >= kRuntimeLoop ? n_layers : LAYERS) ? li + 1 : li);
      __builtin_amdgcn_sched_barrier(0);
    }
    C phv = cur.ph;
    if constexpr (VAR == kRuntimeLoopBranch) {
      if (li == next_upload) {
        next_upload += 2;
        asm volatile("" ::: "memory");
        phv = cmul2<T>(C{0.6f, 0.8f}, phv, times_i<T>(phv));
      }
    }
    if constexpr (VAR != kNoPhase && VAR != kOnlyExchange) a = cmul2<T>(phv, a, times_i<T>(a));
    if constexpr (VAR != kNoDpp && VAR != kOnlyExchange) {
      ry_t_dpp4(a, cur.ts[0], cur.ts[1], cur.ts[2], cur.ts[3]);
    }
    if constexpr (VAR != kNoSwaps && VAR != kOnlyExchange) ry_t_swap<5, T>(a, cur.t5);
    if constexpr (full) derive(nxt, raw);
    if constexpr (VAR != kNoSwaps && VAR != kOnlyExchange) ry_t_swap<4, T>(a, cur.t4);
    if constexpr (VAR != kNoExchange) {
      C* buf = s_slab + (size_t)par * (4 * 64);
      par ^= 1;
      buf[wv * 64 + lane] = a;
      if constexpr (VAR == kNoFetchNoBarrier) __builtin_amdgcn_s_waitcnt(0xc07f);
      else __syncthreads();
      const C p1 = buf[(wv ^ 1) * 64 + lane], p2 = buf[(wv ^ 2) * 64 + lane], p3 = buf[(wv ^ 3) * 64 + lane];
      if constexpr (VAR == kFetchAfterBarrier) {
        fetch(raw, li + 1 < LAYERS ? li + 1 : li);
        derive(nxt, raw);
      }
      const C o = __builtin_elementwise_fma(bcast<T>(cur.k1), p1, a);
      const C t = __builtin_elementwise_fma(bcast<T>(cur.k3), p3, bcast<T>(cur.k2) * p2);
      a = o + t;
    }
    (void)tables;
  };
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    int li = 0;
    if constexpr (VAR >= kRuntimeLoop) {
      for (; li + 1 < n_layers; li += 2) {
        layer(ca, cb, li);
        layer(cb, ca, li + 1);
      }
      if (li < n_layers) layer(ca, cb, li);
    } else {
      for (; li + 1 < LAYERS; li += 2) {
        layer(ca, cb, li);
        layer(cb, ca, li + 1);
      }
      if (li < LAYERS) layer(ca, cb, li);
    }
    a = a * C{0.5f, 0.5f};   // keep the values bounded
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) ticks[blockIdx.x] = t1 - t0;
  out[(size_t)blockIdx.x * 256 + tid] = a.x + a.y;
}

template <int VAR>
static void run(float* out, unsigned long long* ticks, const float* tab, int iters) {
  const int blocks = 256;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(layer_loop<VAR>, dim3(blocks), dim3(256), 0, 0, out, ticks, iters, tab, 13, 0x7fffffff);
    CHECK(hipDeviceSynchronize());
  }
  std::vector<unsigned long long> h(blocks);
  CHECK(hipMemcpy(h.data(), ticks, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double sum = 0;
  for (auto v : h) sum += (double)v;
  printf("%-76s %7.1f ticks/layer\n", kNames[VAR], sum / blocks / ((double)iters * 13));
}

int main() {
  float *out, *tab;
  unsigned long long* ticks;
  CHECK(hipMalloc(&out, 256 * 256 * sizeof(float)));
  CHECK(hipMalloc(&ticks, 256 * sizeof(unsigned long long)));
  CHECK(hipMalloc(&tab, 512 * sizeof(float)));
  std::vector<float> h(512);
  for (int i = 0; i < 512; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.0f;
  CHECK(hipMemcpy(tab, h.data(), 512 * sizeof(float), hipMemcpyHostToDevice));
  const int iters = 400;
  run<kFull>(out, ticks, tab, iters);
  run<kFetchAfterBarrier>(out, ticks, tab, iters);
  run<kNoFetch>(out, ticks, tab, iters);
  run<kNoFetchNoBarrier>(out, ticks, tab, iters);
  run<kNoExchange>(out, ticks, tab, iters);
  run<kNoSwaps>(out, ticks, tab, iters);
  run<kNoDpp>(out, ticks, tab, iters);
  run<kNoPhase>(out, ticks, tab, iters);
  run<kOnlyExchange>(out, ticks, tab, iters);
  run<kRuntimeLoop>(out, ticks, tab, iters);
  run<kRuntimeLoopBranch>(out, ticks, tab, iters);
  return 0;
}
