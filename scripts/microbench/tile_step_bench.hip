// Microbenchmark: shader cycles per optimizer-network evaluation of one 16-coordinate tile
// (no HBM traffic: state stays in registers), one wave per SIMD on every CU.
//   fp32 : l2o::lstm_tile_step   (v_mfma_f32_16x16x4_f32 path)
//   bx3  : l2o::bx::tile_step    (bf16x3 split on v_mfma_f32_16x16x32_bf16)
// hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize \
//       -I open_l2o_amd/csrc scripts/microbench/tile_step_bench.hip -o build/tile_step_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "l2o_lstm_bx3.h"
using namespace l2o;

template <int PRE, bool BX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_bench(const float* wpack, float* out, long long* cyc, int iters) {
  const int lane = threadIdx.x & 63, q = lane >> 4;
  TileState s;
#pragma unroll
  for (int t = 0; t < kNT; ++t) { s.h1[t] = 0.01f * lane; s.c1[t] = 0.02f * t; s.h2[t] = 0.f; s.c2[t] = 0.f; }
  float acc = 0.f, g = 0.001f * lane;
  long long t0, t1;
  if (BX) {
    bx::NetWB<PRE> w;
    bx::load_netw<PRE>(w, wpack, lane);
    t0 = __builtin_amdgcn_s_memtime();
#pragma nounroll
    for (int it = 0; it < iters; ++it) {
      const float d = bx::tile_step<PRE>(w, s, g, g * 0.5f, q);
      acc += d; g = __builtin_fmaf(d, 0.01f, g);
    }
    t1 = __builtin_amdgcn_s_memtime();
  } else {
    NetW<PRE> w;
    load_netw<PRE>(w, wpack, lane);
    t0 = __builtin_amdgcn_s_memtime();
#pragma nounroll
    for (int it = 0; it < iters; ++it) {
      const float d = lstm_tile_step<PRE>(w, s, g, g * 0.5f, q);
      acc += d; g = __builtin_fmaf(d, 0.01f, g);
    }
    t1 = __builtin_amdgcn_s_memtime();
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + s.h1[0] + s.c2[4];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int PRE, bool BX>
void run(const char* name, const float* wpack, float* out, long long* cyc) {
  const int iters = 2000;
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL((k_bench<PRE, BX>), dim3(256), dim3(256), 0, 0, wpack, out, cyc, iters);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL((k_bench<PRE, BX>), dim3(256), dim3(256), 0, 0, wpack, out, cyc, iters);
  (void)hipEventRecord(b);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  long long cy; (void)hipMemcpy(&cy, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-24s %8.1f ns / tile-step   %6lld s_memtime ticks\n", name, ms * 1e6f / iters, cy / iters);
}

int main() {
  const size_t nw = 1 << 16;
  std::vector<float> h(nw);
  for (size_t i = 0; i < nw; ++i) h[i] = 0.05f * (float)((i * 2654435761u >> 8) % 200) / 200.0f - 0.025f;
  // bf16 section: any bit pattern is a finite small number if we keep exponents small -> reuse floats' high halves
  float* wpack; float* out; long long* cyc;
  (void)hipMalloc(&wpack, nw * 4); (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 8);
  (void)hipMemcpy(wpack, h.data(), nw * 4, hipMemcpyHostToDevice);
  run<L2O_PRE_IDENTITY, false>("DM identity  fp32 MFMA", wpack, out, cyc);
  run<L2O_PRE_IDENTITY, true>("DM identity  bf16x3", wpack, out, cyc);
  run<L2O_PRE_LOGSIGN, true>("DM logsign   bf16x3", wpack, out, cyc);
  run<L2O_PRE_FC_ELU, false>("RNNProp      fp32 MFMA", wpack, out, cyc);
  run<L2O_PRE_FC_ELU, true>("RNNProp      bf16x3", wpack, out, cyc);
  return 0;
}
