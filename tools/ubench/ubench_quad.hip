// ubench_quad.hip -- cost model for the four-wavefront sampler's layer (tools/ubench/README in DESIGN.md section 4):
// dependent-chain latencies of the instruction shapes a layer is made of, measured the way the kernel runs them --
// one 256-thread workgroup per CU on all 256 CUs (one wavefront per SIMD), s_memtime around the loop.
//   hipcc -O3 --offload-arch=gfx950 -o ubench_quad ubench_quad.hip && ./ubench_quad
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

__device__ __forceinline__ v2f bc(float v) { return v2f{v, v}; }
__device__ __forceinline__ float dpp_f(float v, const int ctrl) { return v; }
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ v2f dpp2(v2f a) { return v2f{dpp<CTRL>(a.x), dpp<CTRL>(a.y)}; }

enum { kPkFma = 0, kRyDppPk, kRyDppScalar, kRyDppFused, kRyPermlane, kLdsExchange, kLdsExchangeNoBarrier, kPhase, kQuad4, kCount };
static const char* kNames[] = {"dependent v_pk_fma_f32",
                               "RY via 2 dpp mov + pk_mul + pk_fma (quad_perm)",
                               "RY via 2 dpp mov + 2 mul + 2 fma (scalar f32)",
                               "RY via 2 mul + 2 v_fmac_f32_dpp (fused operand)",
                               "RY via permlane32 swap pair (swap, 2 mul, 2 fma, swap)",
                               "wave-bit exchange: ds_write_b64, s_barrier, 3 ds_read_b64, 5 pk",
                               "same without s_barrier (LDS round trip alone; racy values)",
                               "phase multiply (2 pk, complex)",
                               "two DPP bits merged: 3 quad_perm fetches + 4-term real 4x4"};

template <int VAR>
__global__ __launch_bounds__(256) void chain_kernel(float* out, unsigned long long* ticks, int iters, const float* coef) {
  __shared__ v2f slab[2][4][64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  v2f a = v2f{0.001f * (float)(tid + 1), 0.002f * (float)(tid + 3)};
  const float c = coef[0], s = coef[1], sg = (lane & 1) ? s : -s;
  const float k0 = coef[2], k1 = coef[3], k2 = coef[4], k3 = coef[5];
  int par = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if constexpr (VAR == kPkFma) {
        a = __builtin_elementwise_fma(bc(c), a, bc(s));
      } else if constexpr (VAR == kRyDppPk) {
        const v2f p = dpp2<0xB1>(a);
        a = __builtin_elementwise_fma(bc(sg), p, bc(c) * a);
      } else if constexpr (VAR == kRyDppScalar) {
        const float px = dpp<0xB1>(a.x), py = dpp<0xB1>(a.y);
        a = v2f{fmaf(sg, px, c * a.x), fmaf(sg, py, c * a.y)};
      } else if constexpr (VAR == kRyDppFused) {
        float tx = c * a.x, ty = c * a.y;
        asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                     : "+v"(tx) : "v"(a.x), "v"(sg));
        asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                     : "+v"(ty) : "v"(a.y), "v"(sg));
        a = v2f{tx, ty};
      } else if constexpr (VAR == kRyPermlane) {
        unsigned lo = __float_as_uint(a.x), hi = __float_as_uint(a.y);
        auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
        const float l = __uint_as_float(r[0]), h = __uint_as_float(r[1]);
        const float nl = fmaf(-s, h, c * l), nh = fmaf(s, l, c * h);
        auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(nl), __float_as_uint(nh), false, false);
        a = v2f{__uint_as_float(q[0]), __uint_as_float(q[1])};
      } else if constexpr (VAR == kLdsExchange || VAR == kLdsExchangeNoBarrier) {
        slab[par][wv][lane] = a;
        const v2f own = bc(k0) * a;
        if constexpr (VAR == kLdsExchange) __syncthreads();
        else __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        const v2f p1 = slab[par][wv ^ 1][lane], p2 = slab[par][wv ^ 2][lane], p3 = slab[par][wv ^ 3][lane];
        par ^= 1;
        const v2f o = __builtin_elementwise_fma(bc(k1), p1, own);
        const v2f t = __builtin_elementwise_fma(bc(k3), p3, bc(k2) * p2);
        a = o + t;
      } else if constexpr (VAR == kPhase) {
        const v2f ph = v2f{c, s};
        a = __builtin_elementwise_fma(bc(ph.y), v2f{-a.y, a.x}, bc(ph.x) * a);
      } else if constexpr (VAR == kQuad4) {
        const v2f x1 = dpp2<0xB1>(a), x2 = dpp2<0x4E>(a), x3 = dpp2<0x1B>(a);
        const v2f o = __builtin_elementwise_fma(bc(k1), x1, bc(k0) * a);
        const v2f t = __builtin_elementwise_fma(bc(k3), x3, bc(k2) * x2);
        a = o + t;
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) ticks[blockIdx.x] = t1 - t0;
  out[(size_t)blockIdx.x * 256 + tid] = a.x + a.y;
}

template <int VAR>
static void run(float* out, unsigned long long* ticks, const float* coef, int iters) {
  const int blocks = 256;
  hipLaunchKernelGGL(chain_kernel<VAR>, dim3(blocks), dim3(256), 0, 0, out, ticks, iters, coef);
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(chain_kernel<VAR>, dim3(blocks), dim3(256), 0, 0, out, ticks, iters, coef);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(blocks);
  CHECK(hipMemcpy(h.data(), ticks, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double sum = 0;
  for (auto v : h) sum += (double)v;
  const double per = sum / blocks / ((double)iters * 8);
  printf("%-72s %8.1f ticks/op   (%.1f ns/op wall)\n", kNames[VAR], per, ms * 1e6 / ((double)iters * 8));
}

int main() {
  float *out, *coef;
  unsigned long long* ticks;
  CHECK(hipMalloc(&out, 256 * 256 * sizeof(float)));
  CHECK(hipMalloc(&ticks, 256 * sizeof(unsigned long long)));
  CHECK(hipMalloc(&coef, 8 * sizeof(float)));
  const float h[8] = {0.8f, 0.6f, 0.64f, 0.48f, 0.48f, 0.36f, 0, 0};
  CHECK(hipMemcpy(coef, h, sizeof(h), hipMemcpyHostToDevice));
  const int iters = 2000;
  run<kPkFma>(out, ticks, coef, iters);
  run<kPhase>(out, ticks, coef, iters);
  run<kRyDppPk>(out, ticks, coef, iters);
  run<kRyDppScalar>(out, ticks, coef, iters);
  run<kRyDppFused>(out, ticks, coef, iters);
  run<kQuad4>(out, ticks, coef, iters);
  run<kRyPermlane>(out, ticks, coef, iters);
  run<kLdsExchange>(out, ticks, coef, iters);
  run<kLdsExchangeNoBarrier>(out, ticks, coef, iters);
  return 0;
}
