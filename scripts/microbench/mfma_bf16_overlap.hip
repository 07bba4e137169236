// Does v_mfma_f32_16x16x32_bf16 overlap with f32 VALU / transcendental work on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int NV>   // 0: MFMA only, 1: VALU only, 2: interleaved, 3: exp+fma only, 4: MFMA + exp/fma
__global__ void k(float* out, int iters) {
  f32x4 acc[5];
  for (int t = 0; t < 5; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  bf16x8 av, bv;
  for (int i = 0; i < 8; ++i) { av[i] = (__bf16)(a + i); bv[i] = (__bf16)(b * i); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 10; ++u) {
      if (MODE == 0 || MODE == 2 || MODE == 4)
        acc[u % 5] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[u % 5], 0, 0, 0);
      if (MODE == 1 || MODE == 2) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = __builtin_fmaf(v[i], b, a);
      }
      if (MODE == 3 || MODE == 4) {
#pragma unroll
        for (int i = 0; i < NV / 2; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]);
#pragma unroll
        for (int i = NV / 2; i < NV; ++i) v[i] = __builtin_fmaf(v[i], b, a);
      }
      if (MODE == 2 || MODE == 4) __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
  for (int t = 0; t < 5; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NV>
void run(const char* name, int threads, float* d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(threads), 0, 0, d, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s NV=%d waves/SIMD=%d: %.3f ms, %.2f ns per group\n", name, NV, threads / 256, ms, ms * 1e6 / (iters * 10.0));
}

int main() {
  float* d; hipMalloc(&d, 256 * 1024 * 4);
  run<0, 4>("bf16 MFMA only (1/group)", 256, d);
  run<1, 4>("VALU only (NV fma/group)", 256, d);
  run<2, 4>("bf16 MFMA + NV fma", 256, d);
  run<1, 8>("VALU only (NV fma/group)", 256, d);
  run<2, 8>("bf16 MFMA + NV fma", 256, d);
  run<3, 4>("exp+fma only", 256, d);
  run<4, 4>("bf16 MFMA + exp/fma", 256, d);
  run<3, 8>("exp+fma only", 256, d);
  run<4, 8>("bf16 MFMA + exp/fma", 256, d);
  run<0, 4>("bf16 MFMA only (1/group)", 512, d);
  run<1, 8>("VALU only (NV fma/group)", 512, d);
  run<2, 8>("bf16 MFMA + NV fma", 512, d);
  run<3, 8>("exp+fma only", 512, d);
  run<4, 8>("bf16 MFMA + exp/fma", 512, d);
  return 0;
}
