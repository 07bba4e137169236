// Microbenchmark: cost of the LSTM gate nonlinearity block (lstm_gates5) for one wave per
// SIMD, in shader cycles per call, for candidate formulations:
//   V0 current (argument scaling on the VALU, scalar fp32 ops)
//   V1 exp2 arguments pre-scaled on the host (no argument multiplies; -|x| source modifiers)
//   V2 V1 + unit pairs as float2 (v_pk_add/mul/fma_f32)
//   VN feedback only (the loop overhead to subtract)
// hipcc -O3 --offload-arch=gfx950 -I open_l2o_amd/csrc scripts/microbench/gates_variants.hip -o /tmp/gates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int kNT = 5;
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }

__device__ __forceinline__ void gates_v0(const f32x4 (&acc)[kNT], float (&c)[kNT], float (&h)[kNT]) {
  constexpr float kL2E = 1.4426950408889634f;
  float e_i[kNT], E_j[kNT], e_f[kNT], e_o[kNT], ij[kNT], rf[kNT], cn[kNT], E_c[kNT], ro[kNT];
#pragma unroll
  for (int t = 0; t < kNT; ++t) e_i[t] = ex2(acc[t][0] * -kL2E);
#pragma unroll
  for (int t = 0; t < kNT; ++t) E_j[t] = ex2(__builtin_fabsf(acc[t][1]) * (-2.0f * kL2E));
#pragma unroll
  for (int t = 0; t < kNT; ++t) e_f[t] = ex2(__builtin_fmaf(acc[t][2], -kL2E, -kL2E));
#pragma unroll
  for (int t = 0; t < kNT; ++t) e_o[t] = ex2(acc[t][3] * -kL2E);
#pragma unroll
  for (int t = 0; t < kNT; ++t) ij[t] = rcp((1.0f + e_i[t]) * (1.0f + E_j[t]));
#pragma unroll
  for (int t = 0; t < kNT; ++t) rf[t] = rcp(1.0f + e_f[t]);
#pragma unroll
  for (int t = 0; t < kNT; ++t)
    cn[t] = __builtin_fmaf(rf[t], c[t], __builtin_copysignf((1.0f - E_j[t]) * ij[t], acc[t][1]));
#pragma unroll
  for (int t = 0; t < kNT; ++t) E_c[t] = ex2(__builtin_fabsf(cn[t]) * (-2.0f * kL2E));
#pragma unroll
  for (int t = 0; t < kNT; ++t) ro[t] = rcp((1.0f + E_c[t]) * (1.0f + e_o[t]));
#pragma unroll
  for (int t = 0; t < kNT; ++t) { c[t] = cn[t]; h[t] = __builtin_copysignf((1.0f - E_c[t]) * ro[t], cn[t]); }
}

// acc already holds: [0] -log2e*i  [1] 2log2e*j  [2] -log2e*(f+1)  [3] -log2e*o
__device__ __forceinline__ void gates_v1(const f32x4 (&acc)[kNT], float (&c)[kNT], float (&h)[kNT]) {
  constexpr float k2 = 2.0f * 1.4426950408889634f;
  float e_i[kNT], E_j[kNT], e_f[kNT], e_o[kNT], ij[kNT], rf[kNT], cn[kNT], E_c[kNT], ro[kNT];
#pragma unroll
  for (int t = 0; t < kNT; ++t) e_i[t] = ex2(acc[t][0]);
#pragma unroll
  for (int t = 0; t < kNT; ++t) E_j[t] = ex2(-__builtin_fabsf(acc[t][1]));
#pragma unroll
  for (int t = 0; t < kNT; ++t) e_f[t] = ex2(acc[t][2]);
#pragma unroll
  for (int t = 0; t < kNT; ++t) e_o[t] = ex2(acc[t][3]);
#pragma unroll
  for (int t = 0; t < kNT; ++t) ij[t] = rcp((1.0f + e_i[t]) * (1.0f + E_j[t]));
#pragma unroll
  for (int t = 0; t < kNT; ++t) rf[t] = rcp(1.0f + e_f[t]);
#pragma unroll
  for (int t = 0; t < kNT; ++t)
    cn[t] = __builtin_fmaf(rf[t], c[t], __builtin_copysignf((1.0f - E_j[t]) * ij[t], acc[t][1]));
#pragma unroll
  for (int t = 0; t < kNT; ++t) E_c[t] = ex2(-__builtin_fabsf(cn[t] * k2));
#pragma unroll
  for (int t = 0; t < kNT; ++t) ro[t] = rcp((1.0f + E_c[t]) * (1.0f + e_o[t]));
#pragma unroll
  for (int t = 0; t < kNT; ++t) { c[t] = cn[t]; h[t] = __builtin_copysignf((1.0f - E_c[t]) * ro[t], cn[t]); }
}

__device__ __forceinline__ f32x2 mk2(float a, float b) { f32x2 r; r.x = a; r.y = b; return r; }
template <int T0>
__device__ __forceinline__ void gates_pair(const f32x4 (&acc)[kNT], float (&c)[kNT], float (&h)[kNT]) {
  constexpr int T1 = T0 + 1;
  constexpr float k2 = 2.0f * 1.4426950408889634f;
  const f32x2 one = mk2(1.0f, 1.0f);
  const f32x2 e_i = mk2(ex2(acc[T0][0]), ex2(acc[T1][0]));
  const f32x2 E_j = mk2(ex2(-__builtin_fabsf(acc[T0][1])), ex2(-__builtin_fabsf(acc[T1][1])));
  const f32x2 e_f = mk2(ex2(acc[T0][2]), ex2(acc[T1][2]));
  const f32x2 e_o = mk2(ex2(acc[T0][3]), ex2(acc[T1][3]));
  const f32x2 dij = (one + e_i) * (one + E_j);
  const f32x2 df = one + e_f;
  const f32x2 ij = mk2(rcp(dij.x), rcp(dij.y));
  const f32x2 rf = mk2(rcp(df.x), rcp(df.y));
  f32x2 tj = (one - E_j) * ij;
  tj = mk2(__builtin_copysignf(tj.x, acc[T0][1]), __builtin_copysignf(tj.y, acc[T1][1]));
  const f32x2 cp = mk2(c[T0], c[T1]);
  const f32x2 cn = __builtin_elementwise_fma(rf, cp, tj);
  const f32x2 ca = cn * mk2(k2, k2);
  const f32x2 E_c = mk2(ex2(-__builtin_fabsf(ca.x)), ex2(-__builtin_fabsf(ca.y)));
  const f32x2 dro = (one + E_c) * (one + e_o);
  const f32x2 ro = mk2(rcp(dro.x), rcp(dro.y));
  const f32x2 hh = (one - E_c) * ro;
  c[T0] = cn.x; c[T1] = cn.y;
  h[T0] = __builtin_copysignf(hh.x, cn.x); h[T1] = __builtin_copysignf(hh.y, cn.y);
}
__device__ __forceinline__ void gates_v2(const f32x4 (&acc)[kNT], float (&c)[kNT], float (&h)[kNT]) {
  gates_pair<0>(acc, c, h);
  gates_pair<2>(acc, c, h);
  // the odd unit: scalar
  constexpr float k2 = 2.0f * 1.4426950408889634f;
  const float e_i = ex2(acc[4][0]), E_j = ex2(-__builtin_fabsf(acc[4][1])), e_f = ex2(acc[4][2]), e_o = ex2(acc[4][3]);
  const float ij = rcp((1.0f + e_i) * (1.0f + E_j)), rf = rcp(1.0f + e_f);
  const float cn = __builtin_fmaf(rf, c[4], __builtin_copysignf((1.0f - E_j) * ij, acc[4][1]));
  const float E_c = ex2(-__builtin_fabsf(cn * k2));
  const float ro = rcp((1.0f + E_c) * (1.0f + e_o));
  c[4] = cn; h[4] = __builtin_copysignf((1.0f - E_c) * ro, cn);
}

template <int V>
__global__ __launch_bounds__(256) void k_bench(float* out, long long* cyc, int iters) {
  const int lane = threadIdx.x;
  f32x4 acc[kNT];
  float c[kNT], h[kNT];
#pragma unroll
  for (int t = 0; t < kNT; ++t) {
    c[t] = 0.01f * (lane + t); h[t] = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] = 0.001f * (lane * 4 + r + t) - 0.1f;
  }
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma nounroll
  for (int it = 0; it < iters; ++it) {
    if (V == 0) gates_v0(acc, c, h);
    if (V == 1) gates_v1(acc, c, h);
    if (V == 2) gates_v2(acc, c, h);
    if (V == 3) {
#pragma unroll
      for (int t = 0; t < kNT; ++t) { h[t] = acc[t][0] + c[t]; c[t] = acc[t][1] * c[t]; }
    }
#pragma unroll
    for (int t = 0; t < kNT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] = __builtin_fmaf(h[(t + r) % kNT], 0.37f, acc[t][r] * 0.5f);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < kNT; ++t) s += h[t] + c[t];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int V>
void run(const char* name, int waves_per_simd, float* out, long long* cyc, float* base) {
  const int iters = 2000;
  const dim3 grid(256 * waves_per_simd), block(256);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k_bench<V>, grid, block, 0, 0, out, cyc, iters);
  hipEventRecord(a);
  hipLaunchKernelGGL(k_bench<V>, grid, block, 0, 0, out, cyc, iters);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, a, b);
  long long cy; hipMemcpy(&cy, cyc, 8, hipMemcpyDeviceToHost);
  const float ns = ms * 1e6f / iters;
  if (V == 3) *base = ns;
  printf("%-28s waves/SIMD=%d: %8.1f ns/iter (%7.1f net of loop)   s_memtime %lld ticks/iter\n", name,
         waves_per_simd, ns, ns - *base, cy / iters);
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&cyc, 8);
  for (int wps = 1; wps <= 2; ++wps) {
    float base = 0.f;
    run<3>("feedback only", wps, out, cyc, &base);
    run<0>("V0 current", wps, out, cyc, &base);
    run<1>("V1 prescaled", wps, out, cyc, &base);
    run<2>("V2 prescaled+packed", wps, out, cyc, &base);
  }
  return 0;
}
