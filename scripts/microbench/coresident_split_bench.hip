// Microbenchmark / prototype: does a SECOND instruction stream per SIMD shorten the LSTM
// tile-step?  One "layer unit" of the fused unroll's critical path is
//     20 packed bf16x3 MFMAs (a chunk: 5 M-tiles x 4)  ->  gate nonlinearities of the lane's 5 units  ->  split5(h)
// (per step the kernels run two of these back to back, plus the GEMV / exchange phases).  Three forms, all with every
// SIMD of the chip busy, cycles per layer unit from s_memtime on wave 0 and ns from HIP events:
//
//   one_wave   : the production shape -- one wave per SIMD owns the whole tile (5 M-tiles, 5 units per lane)
//   split_3_2  : the co-resident half-tile split of DESIGN 8.1 -- TWO waves per SIMD (waves w and w + 4 of a 512-thread
//                workgroup), M-tiles {0,1,2} / {3,4}: 12 / 8 MFMAs and the gate math of 3 / 2 units per lane; every
//                layer the waves swap their new h through LDS (ds_write, workgroup barrier, ds_read) because the next
//                chunk's B operand needs all 5 units of the lane group; each wave then splits the full 5-vector
//   split_nosync: the same two-wave split WITHOUT the exchange (wrong results; an upper bound on what the split could
//                gain if the swap were free)
//
// Build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize -I open_l2o_amd/csrc \
//        scripts/microbench/coresident_split_bench.hip -o build/coresident_split_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "l2o_lstm_bx3.h"
using namespace l2o;
using bx::u32x4;

// gate nonlinearities of ONE unit (same algebra as bx::gates5's tail)
__device__ __forceinline__ void gate1(const f32x4& a, float& c, float& h) {
  constexpr float k2 = 2.0f * 1.4426950408889634f;
  const float e_i = fast_exp2(a[0]), E_j = fast_exp2(-__builtin_fabsf(a[1]));
  const float e_f = fast_exp2(a[2]), e_o = fast_exp2(a[3]);
  const float ij = fast_rcp((1.0f + e_i) * (1.0f + E_j)), rf = fast_rcp(1.0f + e_f);
  const float cn = __builtin_fmaf(rf, c, __builtin_copysignf((1.0f - E_j) * ij, a[1]));
  const float E_c = fast_exp2(-__builtin_fabsf(cn * k2));
  const float ro = fast_rcp((1.0f + E_c) * (1.0f + e_o));
  c = cn;
  h = __builtin_copysignf((1.0f - E_c) * ro, cn);
}

// ---- one wave per SIMD: the production shape ----------------------------------------------------------------
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_one_wave(const unsigned* frag, float* out, long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  u32x4 a[kNT][bx::kPack];
#pragma unroll
  for (int t = 0; t < kNT; ++t)
#pragma unroll
    for (int j = 0; j < bx::kPack; ++j) a[t][j] = *reinterpret_cast<const u32x4*>(frag + ((t * bx::kPack + j) * 64 + lane) * 4);
  float h[kNT], c[kNT];
#pragma unroll
  for (int t = 0; t < kNT; ++t) { h[t] = 0.01f * (lane + t); c[t] = 0.02f * t; }
  bx::BOp<true> b;
  bx::split5<true>(h, 0u, b);
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma nounroll
  for (int it = 0; it < iters; ++it) {
    f32x4 acc[kNT];
#pragma unroll
    for (int t = 0; t < kNT; ++t) acc[t] = f32x4{0.1f, -0.2f, 0.3f, 0.05f};
#pragma unroll
    for (int n = 0; n < kNT * bx::kPack; ++n) acc[n % kNT] = bx::mfma_bf(a[n % kNT][n / kNT], b.m[n / kNT], acc[n % kNT]);
    bx::gates5(acc, c, h);
    bx::split5<true>(h, 0u, b);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = h[0] + h[4] + c[2];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// ---- two waves per SIMD, M-tiles {0,1,2} / {3,4} -------------------------------------------------------------
template <bool SYNC>
__global__ __launch_bounds__(512) void k_split(const unsigned* frag, float* out, long long* cyc, int iters) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int role = wv >> 2;                      // waves w and w + 4 share SIMD w % 4 (round-robin wave placement)
  const int pairw = wv & 3;                      // which tile of the workgroup
  __shared__ float hx[4][kNT][64];               // [tile][unit slice t][lane]: the h values of the tile's lane groups
  u32x4 a[3][bx::kPack];                         // role 0: M-tiles 0..2, role 1: M-tiles 3, 4 (third entry unused)
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int j = 0; j < bx::kPack; ++j) {
      const int tt = role == 0 ? t : (t < 2 ? 3 + t : 4);
      a[t][j] = *reinterpret_cast<const u32x4*>(frag + ((tt * bx::kPack + j) * 64 + lane) * 4);
    }
  float h[kNT], c[kNT];
#pragma unroll
  for (int t = 0; t < kNT; ++t) { h[t] = 0.01f * (lane + t); c[t] = 0.02f * t; }
  bx::BOp<true> b;
  bx::split5<true>(h, 0u, b);
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma nounroll
  for (int it = 0; it < iters; ++it) {
    if (role == 0) {                             // (uniform per wave)
      f32x4 acc[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) acc[t] = f32x4{0.1f, -0.2f, 0.3f, 0.05f};
#pragma unroll
      for (int n = 0; n < 3 * bx::kPack; ++n) acc[n % 3] = bx::mfma_bf(a[n % 3][n / 3], b.m[n / 3], acc[n % 3]);
      f32x4 acc5[kNT] = {acc[0], acc[1], acc[2], acc[2], acc[2]};
      bx::gates_pair<0>(acc5, c, h);
      gate1(acc[2], c[2], h[2]);
      if (SYNC) { hx[pairw][0][lane] = h[0]; hx[pairw][1][lane] = h[1]; hx[pairw][2][lane] = h[2]; }
    } else {
      f32x4 acc[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[t] = f32x4{0.1f, -0.2f, 0.3f, 0.05f};
#pragma unroll
      for (int n = 0; n < 2 * bx::kPack; ++n) acc[n % 2] = bx::mfma_bf(a[n % 2][n / 2], b.m[n / 2], acc[n % 2]);
      f32x4 acc5[kNT] = {acc[0], acc[0], acc[0], acc[0], acc[1]};
      bx::gates_pair<3>(acc5, c, h);
      if (SYNC) { hx[pairw][3][lane] = h[3]; hx[pairw][4][lane] = h[4]; }
    }
    if (SYNC) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (LDS traffic only)
      if (role == 0) { h[3] = hx[pairw][3][lane]; h[4] = hx[pairw][4][lane]; }
      else { h[0] = hx[pairw][0][lane]; h[1] = hx[pairw][1][lane]; h[2] = hx[pairw][2][lane]; }
    }
    bx::split5<true>(h, 0u, b);
    if (SYNC) asm volatile("s_barrier" ::: "memory");                        // the slots are rewritten next layer
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + tid] = h[0] + h[4] + c[2];
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// ---- round 4, second pass: TWO (512 threads) or THREE (768) FULL waves per SIMD, each owning a whole tile with its fragments
// in registers (122 registers per lane): does the SIMD interleave two independent copies of the production stream?
// (k_unroll_lds runs two waves per SIMD and takes 7 200 cycles per step where one wave takes 4 900.)
template <int THREADS>
__global__ __launch_bounds__(THREADS)
void k_full_waves(const unsigned* frag, float* out, long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  u32x4 a[kNT][bx::kPack];
#pragma unroll
  for (int t = 0; t < kNT; ++t)
#pragma unroll
    for (int j = 0; j < bx::kPack; ++j) a[t][j] = *reinterpret_cast<const u32x4*>(frag + ((t * bx::kPack + j) * 64 + lane) * 4);
  float h[kNT], c[kNT];
#pragma unroll
  for (int t = 0; t < kNT; ++t) { h[t] = 0.01f * (lane + t); c[t] = 0.02f * t; }
  bx::BOp<true> b;
  bx::split5<true>(h, 0u, b);
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma nounroll
  for (int it = 0; it < iters; ++it) {
    f32x4 acc[kNT];
#pragma unroll
    for (int t = 0; t < kNT; ++t) acc[t] = f32x4{0.1f, -0.2f, 0.3f, 0.05f};
#pragma unroll
    for (int n = 0; n < kNT * bx::kPack; ++n) acc[n % kNT] = bx::mfma_bf(a[n % kNT][n / kNT], b.m[n / kNT], acc[n % kNT]);
    bx::gates5(acc, c, h);
    bx::split5<true>(h, 0u, b);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = h[0] + h[4] + c[2];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <class K>
void run(const char* name, K kern, int threads, const unsigned* frag, float* out, long long* cyc) {
  const int iters = 4000;
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, frag, out, cyc, iters);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, frag, out, cyc, iters);
  (void)hipEventRecord(b);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  long long cy; (void)hipMemcpy(&cy, cyc, 8, hipMemcpyDeviceToHost);
  hipFuncAttributes fa;
  (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern));
  printf("%-14s %8.1f ns / layer unit   %6lld cycles   (%d registers per lane)\n", name, ms * 1e6f / iters, cy / iters, fa.numRegs);
}

int main() {
  const size_t nw = kNT * bx::kPack * 64 * 4;
  std::vector<unsigned> h(nw);
  // bf16 pairs with small exponents: 0x3c00..0x3d7f -> |w| ~ 0.008 .. 0.06, alternating sign
  for (size_t i = 0; i < nw; ++i) {
    const unsigned lo = 0x3c00u + (unsigned)((i * 2654435761u >> 9) % 0x180u), hi = 0xbc00u + (unsigned)((i * 40503u >> 3) % 0x180u);
    h[i] = lo | (hi << 16);
  }
  unsigned* frag; float* out; long long* cyc;
  (void)hipMalloc(&frag, nw * 4); (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&cyc, 8);
  (void)hipMemcpy(frag, h.data(), nw * 4, hipMemcpyHostToDevice);
  run("one_wave", k_one_wave, 256, frag, out, cyc);
  run("split_3_2", k_split<true>, 512, frag, out, cyc);
  run("split_nosync", k_split<false>, 512, frag, out, cyc);
  printf("-- full waves (every wave runs the whole layer unit; ns and cycles are per unit of ONE wave) --\n");
  run("full_x1", k_full_waves<256>, 256, frag, out, cyc);
  run("full_x2", k_full_waves<512>, 512, frag, out, cyc);
  run("full_x3", k_full_waves<768>, 768, frag, out, cyc);
  run("full_x4", k_full_waves<1024>, 1024, frag, out, cyc);
  return 0;
}
