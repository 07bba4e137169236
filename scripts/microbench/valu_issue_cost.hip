// Microbenchmark: the ISSUE cost (cycles per instruction, one wave per SIMD, independent operands -- no dependency
// stalls) of the instruction classes the fused unroll's tile-step is made of.  These are the per-class weights of the
// work-based roofline figure of bench.py (`roofline.frac_work`: sum over classes of count x issue cost / measured
// cycles per tile-step), so that the figure does not depend on a hand-waved "quarter rate".
//   v_fma_f32, v_pk_fma_f32, v_exp_f32, v_rcp_f32, v_cvt_pk_bf16_f32, v_permlane32_swap, v_mfma_f32_16x16x32_bf16
// Build: hipcc -O3 --offload-arch=gfx950 scripts/microbench/valu_issue_cost.hip -o build/valu_issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int KIND>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(float* out, long long* cyc, int iters) {
  float r[16];
  for (int i = 0; i < 16; ++i) r[i] = 0.001f * (threadIdx.x + i);
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef __bf16 b8 __attribute__((ext_vector_type(8)));
  f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  f2 p[8];
  for (int i = 0; i < 8; ++i) { p[i].x = r[i]; p[i].y = r[i + 8]; }
  b8 av, bv;
  for (int i = 0; i < 8; ++i) { av[i] = (__bf16)(0.01f * i); bv[i] = (__bf16)(0.02f * i); }
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma nounroll
  for (int it = 0; it < iters; ++it) {
    // 64 independent instructions per trip (16 registers x 4 rounds; a round reuses a register 16 instructions later)
#pragma unroll
    for (int rnd = 0; rnd < 4; ++rnd) {
      if (KIND == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 15]));
      } else if (KIND == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
      } else if (KIND == 2) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
      } else if (KIND == 3) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
      } else if (KIND == 4) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 15]));
      } else if (KIND == 5) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r[i]), "+v"(r[i + 1]));
          asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(r[i]), "+v"(r[i + 1]));
        }
      } else if (KIND == 6) {      // 16 MFMAs over 4 accumulators (the kernels' pattern: consecutive MFMAs, different C)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[i & 3], 0, 0, 0);
      } else {                     // 7: v_exp with one plain VALU in between (mixed stream, as in the gate blocks)
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
          asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i + 1]) : "v"(r[i]));
        }
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += r[i];
  for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND>
void run(const char* name, float* out, long long* cyc, int per_trip) {
  const int iters = 2000;
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
  (void)hipDeviceSynchronize();
  long long cy; (void)hipMemcpy(&cy, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-44s %6.2f cycles per instruction\n", name, (double)cy / iters / per_trip);
}

int main() {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 8);
  run<0>("v_fma_f32", out, cyc, 64);
  run<1>("v_pk_fma_f32 (two fp32 FMAs)", out, cyc, 64);
  run<2>("v_exp_f32", out, cyc, 64);
  run<3>("v_rcp_f32", out, cyc, 64);
  run<4>("v_cvt_pk_bf16_f32", out, cyc, 64);
  run<5>("v_permlane{32,16}_swap_b32", out, cyc, 64);
  run<6>("v_mfma_f32_16x16x32_bf16 (4 accumulators)", out, cyc, 64);
  run<7>("v_exp_f32 + v_fma_f32 alternating (per pair /2)", out, cyc, 64);
  return 0;
}
