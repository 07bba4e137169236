// Microbenchmark: the optimizer-network evaluation of a SEQUENCE of independent 16-coordinate tiles by one
// wave (what k_cwlstm_step and k_unroll_cu do), plain (bx::tile_step per tile) vs software-pipelined across
// tiles (bx::TilePipe: the next tile's two recurrent MFMA chunks are issued underneath this tile's gate blocks).
// Result (profiles/archive_r01_r03/r01_k_microbench_tile_pipe.txt): 0 % (RNNProp) to 6 % (DM nets) faster than the plain sequence
// for any interleaving pattern (-DL2O_PIPE_PAT=0/1/2) -- the matrix pipe does not run beside this gate math.
// hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize \
//       -I open_l2o_amd/csrc scripts/microbench/tile_pipe_bench.hip -o build/tile_pipe_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "l2o_lstm_bx3.h"
using namespace l2o;

namespace l2o {
namespace bx {
// A wave that evaluates the network for a SEQUENCE of independent tiles (k_cwlstm_step, k_unroll_cu),
// software-pipelined across tiles: the two recurrent chunks of the NEXT tile (L2B from its h2, L1H from its h1:
// 60 of a tile's 90 / 120 MFMAs) depend on nothing this tile computes, so they are issued underneath this tile's
// two gate blocks (one MFMA per three VALU / transcendental instructions in program order -- a wave issues in
// order, the matrix pipe runs beside the vector pipe), and the next tile's operand splits underneath this tile's
// input chunk.  prime() issues the first tile's chunks (nothing to hide them under); step() returns the Linear
// output of the current tile and leaves the pipe primed for `nxt` (next = false: last tile of the sequence).
#ifndef L2O_PIPE_PAT
#define L2O_PIPE_PAT 1
#endif
template <int PRE>
struct TilePipe {
  BOp b1n, b2n;
  f32x4 acc1n[kNT], acc2n[kNT];

  __device__ __forceinline__ void prime(const NetWB<PRE>& w, const TileState& s, int q) {
    const unsigned one = q == 0 ? 0x3f800000u : 0u;
    split5(s.h2, one, b2n);
    issue<PRE, kChL2B, 0, kChunkMfmas, true>(w, b2n, acc2n);
    split5(s.h1, one, b1n);
    issue<PRE, kChL1H, 0, kChunkMfmas, true>(w, b1n, acc1n);
  }

  __device__ __forceinline__ float step(const NetWB<PRE>& w, TileState& s, float in0, float in1, int q,
                                        const TileState& nxt, bool next) {
    const unsigned one = q == 0 ? 0x3f800000u : 0u;
    f32x4 acc1[kNT], acc2[kNT];
#pragma unroll
    for (int t = 0; t < kNT; ++t) { acc1[t] = acc1n[t]; acc2[t] = acc2n[t]; }
    // ---- this tile's input chunk; the next tile's operand splits ride underneath it
    if (PRE == L2O_PRE_FC_ELU) {
      float fc[kNT];
#pragma unroll
      for (int t = 0; t < kNT; ++t)
        fc[t] = eluf_(__builtin_fmaf(w.fcw1[t], in1, __builtin_fmaf(w.fcw0[t], in0, w.fcb[t])));
      BOp bf;
      split5(fc, 0u, bf);
      issue<PRE, kChL1X, 0, kChunkMfmas, false>(w, bf, acc1);
    } else {
#pragma unroll
      for (int t = 0; t < kNT; ++t) {
        acc1[t] += w.win0[t] * in0;
        if (PRE == L2O_PRE_LOGSIGN) acc1[t] += w.win1[t] * in1;
      }
    }
    if (next) { split5(nxt.h2, one, b2n); split5(nxt.h1, one, b1n); }
    // ---- layer-1 gates  ||  the next tile's L2B chunk
    __builtin_amdgcn_sched_barrier(0);
    if (next) issue<PRE, kChL2B, 0, kChunkMfmas, true>(w, b2n, acc2n);
    if (L2O_PIPE_PAT == 2) __builtin_amdgcn_sched_barrier(0);
    gates5(acc1, s.c1, s.h1);
    if (next && L2O_PIPE_PAT == 1) {
#pragma unroll
      for (int i = 0; i < kChunkMfmas; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);    // three VALU / transcendental
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    BOp b1;
    split5(s.h1, one, b1);
    issue<PRE, kChL2A, 0, kChunkMfmas, false>(w, b1, acc2);
    // ---- layer-2 gates  ||  the next tile's L1H chunk
    __builtin_amdgcn_sched_barrier(0);
    if (next) issue<PRE, kChL1H, 0, kChunkMfmas, true>(w, b1n, acc1n);
    if (L2O_PIPE_PAT == 2) __builtin_amdgcn_sched_barrier(0);
    gates5(acc2, s.c2, s.h2);
    if (next && L2O_PIPE_PAT == 1) {
#pragma unroll
      for (int i = 0; i < kChunkMfmas; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    float d0 = s.h2[0] * w.wl[0], d1 = s.h2[1] * w.wl[1];
    d0 = __builtin_fmaf(s.h2[2], w.wl[2], d0);
    d1 = __builtin_fmaf(s.h2[3], w.wl[3], d1);
    d0 = __builtin_fmaf(s.h2[4], w.wl[4], d0);
    const float d = quad_q_sum(d0 + d1);
    return d + w.bl;
  }
};

}  // namespace bx
}  // namespace l2o

template <int PRE, int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_bench(const float* wpack, float* out, long long* cyc, int iters) {
  const int lane = threadIdx.x & 63, q = lane >> 4;
  TileState sA, sB;
#pragma unroll
  for (int t = 0; t < kNT; ++t) {
    sA.h1[t] = 0.01f * lane; sA.c1[t] = 0.02f * t; sA.h2[t] = 0.f; sA.c2[t] = 0.f;
    sB.h1[t] = 0.01f * lane + 1; sB.c1[t] = 0.02f * t; sB.h2[t] = 0.1f; sB.c2[t] = 0.f;
  }
  float acc = 0.f, g = 0.001f * lane;
  bx::NetWB<PRE> w;
  bx::load_netw<PRE>(w, wpack, lane);
#pragma unroll
  for (int ch = 0; ch < bx::NetWB<PRE>::NCH; ++ch)
#pragma unroll
    for (int t5 = 0; t5 < kNT; ++t5)
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) asm volatile("" : "+a"(w.a[ch][t5][s3]));
  long long t0, t1;
  t0 = __builtin_amdgcn_s_memtime();
  if (MODE == 0) {
#pragma nounroll
    for (int it = 0; it < iters; it += 2) {
      float d = bx::tile_step<PRE>(w, sA, g, g * 0.5f, q);
      acc += d; g = __builtin_fmaf(d, 0.01f, g);
      d = bx::tile_step<PRE>(w, sB, g, g * 0.5f, q);
      acc += d; g = __builtin_fmaf(d, 0.01f, g);
    }
  } else {
    bx::TilePipe<PRE> pipe;
    pipe.prime(w, sA, q);
#pragma nounroll
    for (int it = 0; it < iters; it += 2) {
      float d = pipe.step(w, sA, g, g * 0.5f, q, sB, true);
      acc += d; g = __builtin_fmaf(d, 0.01f, g);
      d = pipe.step(w, sB, g, g * 0.5f, q, sA, true);
      acc += d; g = __builtin_fmaf(d, 0.01f, g);
    }
  }
  t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + sA.h1[0] + sB.c2[4] + sA.c2[1] + sB.h1[2];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int PRE, int MODE>
float run(const char* name, const float* wpack, float* out, long long* cyc) {
  const int iters = 2000;
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL((k_bench<PRE, MODE>), dim3(256), dim3(256), 0, 0, wpack, out, cyc, iters);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL((k_bench<PRE, MODE>), dim3(256), dim3(256), 0, 0, wpack, out, cyc, iters);
  (void)hipEventRecord(b);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  long long cy; (void)hipMemcpy(&cy, cyc, 8, hipMemcpyDeviceToHost);
  std::vector<float> h(256 * 256);
  (void)hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
  double sum = 0; for (float v : h) sum += v;
  printf("%-28s %8.1f ns / tile-step   %6lld s_memtime ticks   checksum %.9g\n", name, ms * 1e6f / iters, cy / iters, sum);
  return ms;
}

int main() {
  const size_t nw = 1 << 16;
  std::vector<float> h(nw);
  for (size_t i = 0; i < nw; ++i) h[i] = 0.05f * (float)((i * 2654435761u >> 8) % 200) / 200.0f - 0.025f;
  float* wpack; float* out; long long* cyc;
  (void)hipMalloc(&wpack, nw * 4); (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 8);
  (void)hipMemcpy(wpack, h.data(), nw * 4, hipMemcpyHostToDevice);
  run<L2O_PRE_IDENTITY, 0>("DM identity  plain", wpack, out, cyc);
  run<L2O_PRE_IDENTITY, 1>("DM identity  pipelined", wpack, out, cyc);
  run<L2O_PRE_LOGSIGN, 0>("DM logsign   plain", wpack, out, cyc);
  run<L2O_PRE_LOGSIGN, 1>("DM logsign   pipelined", wpack, out, cyc);
  run<L2O_PRE_FC_ELU, 0>("RNNProp      plain", wpack, out, cyc);
  run<L2O_PRE_FC_ELU, 1>("RNNProp      pipelined", wpack, out, cyc);
  return 0;
}
