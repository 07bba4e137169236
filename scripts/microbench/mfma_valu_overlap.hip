// Microbenchmark: do v_mfma_f32_16x16x4_f32 and f32 VALU / transcendental work of the SAME
// wave (or of a co-resident wave) overlap on gfx950?  Prints cycles per loop iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: MFMA only, 1: VALU only, 2: interleaved, 3: TRANS only, 4: MFMA+TRANS interleaved
__global__ void k(float* out, int iters) {
  f32x4 acc[5];
  for (int t = 0; t < 5; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  float v[6];
  for (int i = 0; i < 6; ++i) v[i] = a + i;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 10; ++u) {
      if (MODE == 0 || MODE == 2 || MODE == 4)
        acc[u % 5] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u % 5], 0, 0, 0);
      if (MODE == 1 || MODE == 2) {
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = __builtin_fmaf(v[i], b, a);
      }
      if (MODE == 3 || MODE == 4) {
#pragma unroll
        for (int i = 0; i < 3; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]) ;
#pragma unroll
        for (int i = 3; i < 6; ++i) v[i] = __builtin_fmaf(v[i], b, a);
      }
      if (MODE == 2 || MODE == 4) __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int t = 0; t < 5; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  for (int i = 0; i < 6; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (iters * 10.0f);
}

template <int MODE>
void run(const char* name, int threads, float* d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  float cyc; hipMemcpy(&cyc, d, 4, hipMemcpyDeviceToHost);
  printf("%-28s threads/block=%4d (waves/SIMD=%d): %.3f ms, %.1f ns per group, s_memtime ticks/group=%.1f\n", name, threads,
         threads / 256, ms, ms * 1e6 / (iters * 10.0), cyc);
}

int main() {
  float* d; hipMalloc(&d, 256 * 1024 * 4);
  for (int th : {256, 512}) {
    if (th == 256) {
      run<0>("MFMA only (1/group)", 256, d); run<1>("VALU only (6 fma/group)", 256, d); run<2>("MFMA + 6 fma interleaved", 256, d);
      run<3>("3 exp + 3 fma /group", 256, d); run<4>("MFMA + 3exp+3fma interleaved", 256, d);
    } else {
      run<0>("MFMA only (1/group)", 512, d); run<1>("VALU only (6 fma/group)", 512, d); run<2>("MFMA + 6 fma interleaved", 512, d);
      run<3>("3 exp + 3 fma /group", 512, d); run<4>("MFMA + 3exp+3fma interleaved", 512, d);
    }
  }
  return 0;
}
