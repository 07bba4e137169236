// Do MFMA (f32 16x16x4 / bf16 16x16x32) and f32 VALU / transcendental instructions of the
// same wave overlap on gfx950?  Instruction order is pinned with asm volatile.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4a __attribute__((ext_vector_type(4)));

#define MF32(acc) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MBF(acc) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(av), "v"(bv))
#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(a))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))

// MODE: 0 f32-MFMA only; 1 bf16-MFMA only; 2 NV fma only; 3 NV/2 exp + NV/2 fma only;
//       4 f32-MFMA + fma; 5 f32-MFMA + exp/fma; 6 bf16-MFMA + fma; 7 bf16-MFMA + exp/fma
template <int MODE, int NV>
__global__ void k(float* out, int iters) {
  f32x4 acc[5];
  for (int t = 0; t < 5; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  f32x4 av = {a, b, a, b}, bv = {b, a, b, a};   // 8 bf16 bit patterns, values irrelevant
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 10; ++u) {
      if (MODE == 0 || MODE == 4 || MODE == 5) MF32(acc[u % 5]);
      if (MODE == 1 || MODE == 6 || MODE == 7) MBF(acc[u % 5]);
      if (MODE == 2 || MODE == 4 || MODE == 6) {
#pragma unroll
        for (int i = 0; i < NV; ++i) FMA(v[i]);
      }
      if (MODE == 3 || MODE == 5 || MODE == 7) {
#pragma unroll
        for (int i = 0; i < NV / 2; ++i) { EXP(v[i]); FMA(v[NV / 2 + i]); }
      }
    }
  }
  asm volatile("s_nop 15\n s_nop 15");
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
  printf("%-30s NV=%d waves/SIMD=%d: %7.3f ms, %6.2f ns per group\n", name, NV, threads / 256, ms, ms * 1e6 / (iters * 10.0));
}

template <int W> void all(float* d) {
  run<0, 6>("f32 MFMA only", W, d);
  run<1, 6>("bf16 MFMA only", W, d);
  run<2, 6>("6 fma only", W, d);
  run<3, 6>("3 exp + 3 fma only", W, d);
  run<4, 6>("f32 MFMA + 6 fma", W, d);
  run<5, 6>("f32 MFMA + 3exp/3fma", W, d);
  run<6, 6>("bf16 MFMA + 6 fma", W, d);
  run<7, 6>("bf16 MFMA + 3exp/3fma", W, d);
  run<2, 4>("4 fma only", W, d);
  run<6, 4>("bf16 MFMA + 4 fma", W, d);
  run<3, 4>("2 exp + 2 fma only", W, d);
  run<7, 4>("bf16 MFMA + 2exp/2fma", W, d);
}
int main() {
  float* d; hipMalloc(&d, 256 * 1024 * 4);
  all<256>(d);
  all<512>(d);
  return 0;
}
