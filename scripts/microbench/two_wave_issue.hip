// Microbenchmark: what a SECOND wave on a SIMD buys, per instruction class (the question behind k_unroll_lds' two waves
// per SIMD).  One workgroup per CU, 256 threads (one wave per SIMD) or 512 (two), every wave issuing a
// stream of INDEPENDENT instructions; reported: nanoseconds per instruction PER SIMD (wall clock over the launch, so no
// assumption about which clock s_memtime counts), for
//   fma | exp | mfma (16x16x32 bf16, 4 accumulators) | the gate-block mix (2 fma : 1 exp) | MIXED waves: wave 0 of a SIMD
//   issues MFMAs, wave 1 FMAs (do the matrix pipe and the VALU overlap across waves?)
// Build: hipcc -O3 --offload-arch=gfx950 scripts/microbench/two_wave_issue.hip -o build/two_wave_issue
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

template <int KIND>
__device__ __forceinline__ void body(float (&r)[16], f4 (&acc)[4], b8 av, b8 bv) {
#pragma unroll
  for (int rnd = 0; rnd < 4; ++rnd) {
    if (KIND == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 15]));
    } else if (KIND == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
    } else if (KIND == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[i & 3], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 15; i += 3) {
        asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i + 1]) : "v"(r[i + 2]));
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i + 2]) : "v"(r[i + 1]));
      }
      asm volatile("v_exp_f32 %0, %0" : "+v"(r[15]));
    }
  }
}

// KA: what the first wave of a SIMD runs, KB: what the second runs (512 threads only)
template <int KA, int KB>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  float r[16];
  for (int i = 0; i < 16; ++i) r[i] = 0.001f * (threadIdx.x + i);
  f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  b8 av, bv;
  for (int i = 0; i < 8; ++i) { av[i] = (__bf16)(0.01f * i); bv[i] = (__bf16)(0.02f * i); }
  const bool second = threadIdx.x >= 256;     // waves 4..7 = the second wave of each SIMD
  if (!second) {
#pragma nounroll
    for (int it = 0; it < iters; ++it) body<KA>(r, acc, av, bv);
  } else {
#pragma nounroll
    for (int it = 0; it < iters; ++it) body<KB>(r, acc, av, bv);
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += r[i];
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int KA, int KB>
void run(const char* name, float* out, int threads) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KA, KB>), dim3(256), dim3(threads), 0, 0, out, iters);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<KA, KB>), dim3(256), dim3(threads), 0, 0, out, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double per_wave = (double)iters * 64;                     // instructions per wave
  printf("%-58s %3d threads  %8.3f ms   %6.3f ns per instruction per wave   %6.3f ns per instruction per SIMD\n", name, threads, ms,
         ms * 1e6 / per_wave, ms * 1e6 / (per_wave * (threads / 256)));
}

int main() {
  float* out;
  (void)hipMalloc(&out, 256 * 512 * 4);
  run<0, 0>("v_fma_f32", out, 256);            run<0, 0>("v_fma_f32", out, 512);
  run<1, 1>("v_exp_f32", out, 256);            run<1, 1>("v_exp_f32", out, 512);
  run<2, 2>("v_mfma_f32_16x16x32_bf16", out, 256); run<2, 2>("v_mfma_f32_16x16x32_bf16", out, 512);
  run<3, 3>("gate mix (1 exp : 2 fma)", out, 256); run<3, 3>("gate mix (1 exp : 2 fma)", out, 512);
  run<2, 0>("MIXED: first wave MFMA, second wave v_fma", out, 512);
  run<2, 3>("MIXED: first wave MFMA, second wave gate mix", out, 512);
  run<0, 1>("MIXED: first wave v_fma, second wave v_exp", out, 512);
  return 0;
}
