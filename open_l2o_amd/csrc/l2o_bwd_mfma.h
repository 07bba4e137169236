// l2o_bwd_mfma.h -- one back-propagation-through-time step of the optimizer network on the bf16
// matrix cores (the meta-gradient of MetaOptimizer.meta_minimize, DM/meta.py:398-414; optimizee
// gradient held constant, DM/meta.py:328-329).  Same contract, same A / Bm / carry layouts as
// k_cwlstm_bwd_tile (l2o_bwd.h); included by l2o_kernels.hip after l2o_bwd.h.
//
// Why: k_cwlstm_bwd_tile spends 33 us per step on 16 384 coordinates (79 000 cycles per 16-coordinate
// tile) reading fp32 weight rows from LDS for scalar FMAs; the forward step of the same network
// takes 2.7 us on the matrix pipe.  Here a wave owns one state tile in the FORWARD layout
// (l2o_lstm_bx3.h: lane (c, q) = coordinate c, units 4t + q):
//   * the step's forward is recomputed with the forward kernel's own fragments (chunks L1H, L1X,
//     L2A, L2B of wpack, registers) -- but every gate value (i, j, f, o, tanh c') is kept;
//   * the gate gradients dz (four gate types x five units per lane) are ALREADY the B operands of
//     the transposed products d[in | h(t-1)] = W dz: K-chunk r = gate type r, slot 8q + i = unit
//     4i + q, exactly what bx::split5 builds from five lane-local values -- no cross-lane traffic;
//   * the A operands are W (unscaled, Sonnet layout [input row][r * 20 + u]) as bf16x3 fragments,
//     packed by l2o_wpack_host behind the forward section: M-tile rows rho = 4 q' + j hold input
//     row 4j + q' (j = 0..3) so that D row 4q + j lands in lane (c, q) as ITS unit 4j + q; the
//     fifth unit of the two 20-row halves shares one more M-tile (rows j = 0 / 1).  Layer 2: 3
//     M-tiles (h1(t) | h2(t-1)), layer 1: 2 (h1(t-1)) or 3 (fc features | h1(t-1)) M-tiles; each
//     M-tile x 4 K-chunks x 6 products.  The fragments live in LDS (60 / 72 KB per workgroup, read
//     as ds_read_b128 next to the MFMAs that use them), the forward fragments in registers.
//   * all global traffic through per-wave LDS staging exactly like the fp32 tile kernel (the
//     16 x KA block of A, then -- same buffer -- the 16 x KB block of Bm, and the carries).
// MFMAs per tile-step: forward 90 (120 fc) + backward 120 (144 fc).
#pragma once
#include "l2o_lstm_bx3.h"

namespace l2o {
namespace bxb {

using bx::u32x4;

__host__ __device__ constexpr int tiles2() { return 3; }
__host__ __device__ constexpr int tiles1(int pre) { return pre == L2O_PRE_FC_ELU ? 3 : 2; }
__host__ __device__ constexpr int ntiles(int pre) { return tiles2() + tiles1(pre); }
// word offset of the section inside wpack, and of fragment (M-tile, gate chunk r, split level s) inside it;
// M-tiles 0..2 = layer 2, then layer 1
__host__ __device__ constexpr int base(int pre) { return bx::base(pre) + bx::words(pre); }
__host__ __device__ constexpr int frag_rel(int tile, int r, int s) { return ((tile * 4 + r) * 3 + s) * bx::kFragWords; }
__host__ __device__ constexpr int words(int pre) { return ntiles(pre) * 4 * 3 * bx::kFragWords; }
// input row of W served by D row rho of M-tile m of a [first 20 | second 20]-row matrix (-1: none);
// first < 0: the matrix has only the second half (DM layer 1: rows P.. = h1(t-1)) in 2 M-tiles
__host__ __device__ constexpr int src_row(int m, int rho, int first, int second) {
  const int qo = rho >> 2, j = rho & 3;
  if (first >= 0) {
    if (m == 0) return first + 4 * j + qo;
    if (m == 1) return second + 4 * j + qo;
    return j == 0 ? first + 16 + qo : (j == 1 ? second + 16 + qo : -1);
  }
  if (m == 0) return second + 4 * j + qo;
  return j == 0 ? second + 16 + qo : -1;
}

struct Gates {
  float i[kNT], j[kNT], f[kNT], o[kNT], tc[kNT];
};

// acc = the forward kernel's pre-scaled accumulators [-log2e zi, 2 log2e zj, -log2e (zf + 1), -log2e zo]
__device__ __forceinline__ void gates_full(const f32x4 (&acc)[kNT], const float (&cp)[kNT], Gates& g, float (&h)[kNT]) {
  constexpr float k2 = 2.0f * 1.4426950408889634f;
#pragma unroll
  for (int t = 0; t < kNT; ++t) {
    const float e_i = fast_exp2(acc[t][0]), E_j = fast_exp2(-__builtin_fabsf(acc[t][1]));
    const float e_f = fast_exp2(acc[t][2]), e_o = fast_exp2(acc[t][3]);
    g.i[t] = fast_rcp(1.0f + e_i);
    g.j[t] = __builtin_copysignf((1.0f - E_j) * fast_rcp(1.0f + E_j), acc[t][1]);
    g.f[t] = fast_rcp(1.0f + e_f);
    g.o[t] = fast_rcp(1.0f + e_o);
    const float cn = __builtin_fmaf(g.f[t], cp[t], g.i[t] * g.j[t]);
    const float E_c = fast_exp2(-__builtin_fabsf(cn * k2));
    g.tc[t] = __builtin_copysignf((1.0f - E_c) * fast_rcp(1.0f + E_c), cn);
    h[t] = g.tc[t] * g.o[t];
  }
}

// dh, dc_in: gradient w.r.t. this layer's h(t), c(t).  dz*: gate pre-activation gradients, dcp: d c(t-1)
__device__ __forceinline__ void gate_grads(const Gates& g, const float (&cp)[kNT], const float (&dh)[kNT],
                                           const float (&dc_in)[kNT], float (&dz)[4][kNT], float (&dcp)[kNT]) {
#pragma unroll
  for (int t = 0; t < kNT; ++t) {
    const float dc = __builtin_fmaf(dh[t] * g.o[t], 1.0f - g.tc[t] * g.tc[t], dc_in[t]);
    dcp[t] = dc * g.f[t];
    dz[0][t] = dc * g.j[t] * g.i[t] * (1.0f - g.i[t]);
    dz[1][t] = dc * g.i[t] * (1.0f - g.j[t] * g.j[t]);
    dz[2][t] = dc * cp[t] * g.f[t] * (1.0f - g.f[t]);
    dz[3][t] = dh[t] * g.tc[t] * g.o[t] * (1.0f - g.o[t]);
  }
}

// The transposed products keep the one-MFMA-per-product form (six per M-tile, 5 of 8 K-slots used): their fragments
// live in LDS next to the carries, and the packed form of l2o_lstm_bx3.h would need 4/3 of the space.
// acc[m] = sum over the four gate chunks of  W-fragment(m, r) x split(dz[r])
template <int NTL>
__device__ __forceinline__ void tgemm(const unsigned* fr, int lane, const float (&dz)[4][kNT], f32x4 (&acc)[NTL]) {
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int m = 0; m < NTL; ++m) acc[m] = zero;
  static_for<0, 4>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    bx::BOp<false> b;
    bx::split5<false>(dz[r], 0u, b);
    u32x4 a[NTL][3];
#pragma unroll
    for (int m = 0; m < NTL; ++m)
#pragma unroll
      for (int s = 0; s < 3; ++s) a[m][s] = *reinterpret_cast<const u32x4*>(fr + frag_rel(m, r, s) + lane * 4);
    static_for<0, bx::kProducts>([&](auto pc) {
      constexpr int pp = decltype(pc)::value;
#pragma unroll
      for (int m = 0; m < NTL; ++m) acc[m] = bx::mfma_bf(a[m][bx::prod_w(pp)], b.m[bx::prod_x(pp)], acc[m]);
    });
  });
}

}  // namespace bxb

// The staging block and the carries are PER WAVE: inside the step loop a wave only has to order its own LDS accesses
// (in-order LDS pipeline + a compiler fence).  A __syncthreads() there also drains vmcnt -- the write acknowledgements
// of the 15 KB of A / Bm rows the wave stores per step and the prefetch of the next step's history -- three times a
// step.  (Measured: the T = 100 step stays at 1.57 ms -- the kernel is bound by the 2.15 GB of history read and A / Bm rows
// written per launch, 3.1 TB/s; kept because it removes a cross-wave dependency the data flow does not have.)
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// A wave's staging block -> global memory, NF floats (a multiple of 4), both contiguous.  The rolled
// `for (i = lane; i < n / 4; i += 64)` this replaces compiled to ds_read_b128; s_waitcnt lgkmcnt(0); global_store_dwordx4
// per trip -- eleven LDS round trips in a row for the 10 KB Bm block of every step -- and so does the unrolled C++ form, with or
// without sched_group_barriers (the scheduler pairs every read with its store through ONE register quad).  Hence one asm
// statement per FIVE reads, wait included (a load and its wait in separate statements is the hazard l2o_mlp_xcd.h describes).
template <int OFF0, int CNT>
__device__ __forceinline__ void lds_read5_wait(unsigned addr, f32x4 (&v)[5]) {
  static_assert(CNT >= 1 && CNT <= 5 && OFF0 + 4096 < 65536, "ds offset field");
  // (the reads past CNT repeat the last valid one: one asm string for every group size)
  asm volatile("ds_read_b128 %0, %5 offset:%6\n\tds_read_b128 %1, %5 offset:%7\n\tds_read_b128 %2, %5 offset:%8\n\t"
               "ds_read_b128 %3, %5 offset:%9\n\tds_read_b128 %4, %5 offset:%10\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4])
               : "v"(addr), "n"(OFF0), "n"(OFF0 + 1024 * (CNT > 1 ? 1 : CNT - 1)), "n"(OFF0 + 1024 * (CNT > 2 ? 2 : CNT - 1)),
                 "n"(OFF0 + 1024 * (CNT > 3 ? 3 : CNT - 1)), "n"(OFF0 + 1024 * (CNT > 4 ? 4 : CNT - 1))
               : "memory");
}
template <int NF>
__device__ __forceinline__ void stage_block_out(float* __restrict__ dst, const float* stg, int lane) {
  static_assert(NF % 4 == 0, "whole float4s");
  constexpr int N4 = NF / 4, FULL = N4 / 64;                // trips in which every lane has a float4
  asm volatile("" : "+v"(lane));                            // (offsets recomputed here, not hoisted out of the step loop and kept live)
  const unsigned addr = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)stg + 16u * (unsigned)lane;
  f32x4* const d4 = reinterpret_cast<f32x4*>(dst) + lane;
  static_for<0, (FULL + 4) / 5>([&](auto gc) {
    constexpr int k0 = decltype(gc)::value * 5, cnt = FULL - k0 < 5 ? FULL - k0 : 5;
    f32x4 v[5];
    lds_read5_wait<1024 * k0, cnt>(addr, v);
#pragma unroll
    for (int k = 0; k < cnt; ++k) d4[64 * (k0 + k)] = v[k];
  });
  if (N4 % 64 != 0 && lane < N4 % 64) d4[64 * FULL] = reinterpret_cast<const f32x4*>(stg)[lane + 64 * FULL];
}

template <int PRE>
struct BwdMfmaGeom {
  using Geo = BwdTileGeom<PRE>;
  static constexpr int kFragWords = bxb::words(PRE);
  static constexpr int kWaveFloats = 4 * Geo::NC * kH + Geo::NC * Geo::KB;   // carries | the A-then-Bm staging block
  static constexpr int kLdsFloats = kFragWords + 4 * kWaveFloats;
};

template <int PRE>
__global__ __launch_bounds__(256) void k_cwlstm_bwd_mfma(BwdParams p) {
  using Geo = BwdTileGeom<PRE>;
  constexpr int P = Geo::P, K1 = Geo::K1, G = Geo::G, NC = Geo::NC, KA = Geo::KA, KB = Geo::KB;
  constexpr bool FC = PRE == L2O_PRE_FC_ELU;
  constexpr int NT1 = bxb::tiles1(PRE);
#ifdef L2O_BWD_CLOCK
  long long ck[24]; int cki = 0;
#define MCK() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); if (cki < 24) ck[cki++] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define MCK() ((void)0)
#endif
  extern __shared__ float sm[];
  unsigned* fr = reinterpret_cast<unsigned*>(sm);
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, q = lane >> 4;
  MCK();
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* cio = sm + bxb::words(PRE) + wv * BwdMfmaGeom<PRE>::kWaveFloats;   // [4][NC][20] carries in, then out
  float* stg = cio + 4 * NC * kH;                                           // [NC][KA], later [NC][KB]
  {
    // the transposed-product fragments: wpack -> LDS, eight 16-byte loads per thread in flight
    const bx::u32x4* src = reinterpret_cast<const bx::u32x4*>(reinterpret_cast<const unsigned*>(p.wpack) + bxb::base(PRE));
    constexpr int n4 = bxb::words(PRE) / 4;
    for (int b0 = tid; b0 < n4; b0 += 256 * 8) {
      bx::u32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = src[min(b0 + 256 * k, n4 - 1)];   // (unconditional: a predicated v[k] lands in scratch)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int i = b0 + 256 * k;
        if (i < n4) reinterpret_cast<bx::u32x4*>(fr)[i] = v[k];
      }
    }
  }
  constexpr bool PK = bx::packed_default(PRE);       // the forward recomputation uses the net's default gate-GEMM form
  constexpr int kN = bx::chunk_mfmas(PK);
  bx::NetWB<PRE, PK> w;
  bx::load_netw<PRE, true, PK>(w, p.wpack, lane);
  __shared__ __attribute__((aligned(16))) float bias_s[bx::kBiasWords];     // the gate biases = accumulator inits
  bx::stage_bias(bias_s, p.wpack, PRE, threadIdx.x, blockDim.x);          // (the loop's first __syncthreads() orders it)
  bx::set_bias(w, bias_s, q);
#ifndef L2O_BWD_NO_AGPR_PIN
  // the forward fragments (MFMA A operands) pinned to the AGPR half of the register file, like LstmCore::pin():
  // left to itself the allocator parked VALU operands there (212 v_accvgpr_read of the 1 678 instructions of a step)
#pragma unroll
  for (int ch = 0; ch < bx::NetWB<PRE, PK>::NCH; ++ch)
#pragma unroll
    for (int t5 = 0; t5 < kNT; ++t5)
#pragma unroll
      for (int j = 0; j < bx::frags(PK); ++j) asm volatile("" : "+a"(w.a[ch][t5][j]));
#endif
  const unsigned one = bx::bias_one<PK>(q);
  const size_t ntiles = (size_t)p.tile_end[p.nseg - 1];
  const size_t ngrp4 = (ntiles + 3) / 4;
  const size_t RT = (size_t)p.rows_total;
  MCK();                                                    // 1: prologue issued + drained
  const int T = p.T > 0 ? p.T : 1;
  for (size_t g4 = blockIdx.x; g4 < ngrp4; g4 += gridDim.x) {
    __syncthreads();                                        // fragments staged / the previous group's LDS blocks are stored
    MCK();                                                  // 2
    const size_t grp = g4 * 4 + wv;                         // global tile index == row block of A / Bm / carries
    const bool valid = grp < ntiles;
    int sg = 0;
    while (sg + 1 < p.nseg && (int)grp >= p.tile_end[sg]) ++sg;
    const size_t lt = valid ? grp - (sg ? p.tile_end[sg - 1] : 0) : 0;
    const size_t n0 = valid ? grp * NC : 0;
    // tile lt of the panel = tile tp of problem bp (D coordinates, tpp tiles each -- any D: the last tile of a problem
    // may be ragged, exactly as the forward kernels lay out the packed state); its lanes' coordinates in the flat
    // [B * D] history vectors start at bp * D + 16 tp
    const size_t Dp = (size_t)p.seg_d[sg], tpp = (size_t)p.seg_tpp[sg];
    const size_t bp = lt / tpp, tp = lt - bp * tpp;
    const size_t left = Dp - tp * NC;
    const int nv = valid ? (int)(left < (size_t)NC ? left : (size_t)NC) : 0;
    const size_t n = bp * Dp + tp * NC + ((int)c < nv ? c : (nv > 0 ? nv - 1 : 0));   // tail lanes recompute the last coordinate; nothing of theirs is stored
    // ---- the carries of the LAST step through LDS (coalesced); they stay in registers over the steps ----
#pragma unroll
    for (int a4 = 0; a4 < 4; ++a4) {
      float4* cd = reinterpret_cast<float4*>(cio + a4 * NC * kH);
      // (values, not a `cond ? load : zero4` of a const object: that form put the zero vector into scratch)
      float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
      if (p.carry_in) {
        const float4* cs = reinterpret_cast<const float4*>(p.carry_in + ((size_t)a4 * RT + n0) * kH);
        if (lane < nv * 5) v0 = cs[lane];
        if (lane < NC * kH / 4 - 64 && 64 + lane < nv * 5) v1 = cs[64 + lane];
      }
      cd[lane] = v0;
      if (lane < NC * kH / 4 - 64) cd[64 + lane] = v1;
    }
    __syncthreads();                                        // carries are in LDS
    float cdh1[kNT], cdc1[kNT], cdh2[kNT], cdc2[kNT];
#pragma unroll
    for (int t = 0; t < kNT; ++t) {
      cdh1[t] = cio[(0 * NC + c) * kH + 4 * t + q];
      cdc1[t] = cio[(1 * NC + c) * kH + 4 * t + q];
      cdh2[t] = cio[(2 * NC + c) * kH + 4 * t + q];
      cdc2[t] = cio[(3 * NC + c) * kH + 4 * t + q];
    }
    // steps T-1 .. 0 of this tile (l2o_cwlstm_bwd_unroll); one step with the panel table's own pointers otherwise
    double pw1 = p.pw1_last, pw2 = p.pw2_last;              // beta^(step of t) for t = T - 1
    const bool acc_dx = p.table && p.seg_gfinal[sg];        // else the table (or the panel) carries dx_next
    float dxacc = acc_dx ? p.seg_gfinal[sg][n] : 0.0f;   // dL/d(delta_t) = g_final + sum_{tau > t} g_tau
    // The history of step ts - 1 is requested while step ts computes (a single wave per SIMD has nothing
    // else to hide the HBM latency of its 5 KB state tile behind).
    struct Hist { TileState s; float g, m, v, dx; };
    auto fetch = [&](int ts, Hist& h) {
      const float *g_t = p.seg_g[sg], *m_t = p.seg_m[sg], *v_t = p.seg_v[sg], *st_t = p.seg_st[sg], *dx_t = p.seg_dx[sg];
      if (p.table) {
        const float* const* row = p.table + ((size_t)ts * p.nseg + sg) * 5;
        g_t = row[0]; m_t = row[1]; v_t = row[2]; st_t = row[3]; dx_t = row[4];
      }
      load_tile_state(h.s, st_t + lt * kStateFloatsPerTile, lane);
      h.g = g_t[n];
      h.m = FC ? m_t[n] : 0.0f;
      h.v = FC ? v_t[n] : 0.0f;
      h.dx = dx_t ? dx_t[n] : 0.0f;
    };
    Hist cur, nxt;
    fetch(T - 1, cur);
    for (int ts = T - 1; ts >= 0; --ts) {
    if (ts > 0) fetch(ts - 1, nxt);
    float om1 = p.om1, om2 = p.om2;
    if (p.table) {
      om1 = (float)(1.0 - pw1); om2 = (float)(1.0 - pw2);
      pw1 /= (double)p.beta1; pw2 /= (double)p.beta2;
    }
    constexpr int KAC = Geo::KAC;
    const bool compact = p.compact_a != 0;                  // (wave-uniform; T > 0 launches only)
    float* const a_t = compact ? p.act1 + ((size_t)(ts + 1) * RT + n0) * KAC : p.act1 + ((size_t)ts * RT + n0) * KA;
    float* const b_t = p.dz1 + ((size_t)ts * RT + n0) * KB;
    const TileState s = cur.s;                              // h1, c1, h2, c2 BEFORE the step
    const float gv = cur.g;
    float in0 = gv, in1 = 0.0f, f0 = 0.0f, f1 = 0.0f;
    float pre_fc[kNT], fcv[kNT];
    if (FC) {
      const float m_hat = cur.m / om1, v_hat = cur.v / om2;
      const float den = sqrtf(v_hat) + 1e-8f;
      f0 = m_hat / den;
      f1 = gv / den;
#pragma unroll
      for (int t = 0; t < kNT; ++t) {
        pre_fc[t] = __builtin_fmaf(w.fcw1[t], f1, __builtin_fmaf(w.fcw0[t], f0, w.fcb[t]));
        fcv[t] = eluf_(pre_fc[t]);
      }
    } else if (PRE == L2O_PRE_LOGSIGN) {
      in0 = fmaxf(logf(fabsf(gv) + 1.1920928955078125e-07f) * p.k_inv, -1.0f);
      in1 = fminf(fmaxf(gv * p.exp_k, -1.0f), 1.0f);
    }
    const float dxn = acc_dx ? dxacc : cur.dx;
    dxacc += gv;
    MCK();                                                  // 3: loads
    // ---- forward recompute --------------------------------------------------------------------------
    f32x4 acc1[kNT], acc2[kNT];
    // accumulator inits = the biases (l2o_lstm_bx3.h); round 4: unpinned (the pinning asm waited for the ten reads on the
    // spot), the split of h2 hides their latency -- see bx::tile_step
    bx::preload_bias<1, bx::NetWB<PRE, PK>, false>(w, acc2);
    bx::preload_bias<0, bx::NetWB<PRE, PK>, false>(w, acc1);
    {
      bx::BOp<PK> b;
      bx::split5<PK>(s.h2, one, b);
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * kNT, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 32, 0);
      bx::issue<PRE, bx::kChL2B, 0, kN, true>(w, b, acc2);
      bx::split5<PK>(s.h1, one, b);
      bx::issue<PRE, bx::kChL1H, 0, kN, true>(w, b, acc1);
      if constexpr (FC) {
        bx::split5<PK>(fcv, 0u, b);
        bx::issue<PRE, bx::kChL1X, 0, kN, false>(w, b, acc1);
      } else {
#pragma unroll
        for (int t = 0; t < kNT; ++t) {
          acc1[t] += w.win0[t] * in0;
          if (PRE == L2O_PRE_LOGSIGN) acc1[t] += w.win1[t] * in1;
        }
      }
    }
    bxb::Gates g1, g2;
    float h1n[kNT], h2n[kNT];
    bxb::gates_full(acc1, s.c1, g1, h1n);
    {
      bx::BOp<PK> b;
      bx::split5<PK>(h1n, one, b);
      bx::issue<PRE, bx::kChL2A, 0, kN, false>(w, b, acc2);
    }
    bxb::gates_full(acc2, s.c2, g2, h2n);
    float dl = 0.0f;
#pragma unroll
    for (int t = 0; t < kNT; ++t) dl = __builtin_fmaf(h2n[t], w.wl[t], dl);
    const float dlin = quad_q_sum(dl) + w.bl;
    MCK();                                                  // 4: forward
    wave_lds_fence();                                       // the previous step's Bm block has left the staging buffer
    // ---- the A row block: [act1 = in | h1(t-1)] [act2 = h1(t) | h2(t-1)] [h2(t)] [feats] [1] ----------
    // (the rows of a ragged last tile that do not exist are written as zeros: they add nothing to A^T Bm
    //  and the caller need not clear A / Bm)
    const float live = c < nv ? 1.0f : 0.0f;
    if (compact && ts == 0) {
      // block 0 of the compact A: the state BEFORE the unroll's first step in the h1 / h2 columns, zeros elsewhere -- the
      // contraction's h1(t-1) / h2(t-1) operands of step 0 (once per tile: 1 / T of the row traffic)
      float* arow = stg + c * KAC;
#pragma unroll
      for (int t = 0; t < kNT; ++t) {
        if (FC || t == 0) { if (4 * t + q < P) arow[4 * t + q] = 0.0f; }
        arow[P + 4 * t + q] = s.h1[t] * live;
        arow[P + kH + 4 * t + q] = s.h2[t] * live;
      }
      if (q == 0) { for (int e = P + 2 * kH; e < KAC; ++e) arow[e] = 0.0f; }
      wave_lds_fence();
      if (valid) stage_block_out<NC * KAC>(p.act1 + n0 * KAC, stg, lane);
      wave_lds_fence();
    }
    {
      float* arow = stg + c * (compact ? KAC : KA);
      const int o_ft = compact ? P + 2 * kH : K1 + 3 * kH;  // feats (RNNProp), then the ones column
      if (FC) {
#pragma unroll
        for (int t = 0; t < kNT; ++t) arow[4 * t + q] = fcv[t] * live;
        if (q == 0) { arow[o_ft] = f0 * live; arow[o_ft + 1] = f1 * live; }
      } else if (q == 0) {
        arow[0] = in0 * live;
        if (PRE == L2O_PRE_LOGSIGN) arow[1] = in1 * live;
      }
      if (compact) {
#pragma unroll
        for (int t = 0; t < kNT; ++t) {
          arow[P + 4 * t + q] = h1n[t] * live;
          arow[P + kH + 4 * t + q] = h2n[t] * live;
        }
        if (q == 0) arow[KAC - 1] = live;
      } else {
#pragma unroll
        for (int t = 0; t < kNT; ++t) {
          arow[P + 4 * t + q] = s.h1[t] * live;
          arow[K1 + 4 * t + q] = h1n[t] * live;
          arow[K1 + kH + 4 * t + q] = s.h2[t] * live;
          arow[K1 + 2 * kH + 4 * t + q] = h2n[t] * live;
        }
        if (q == 0) arow[KA - 1] = live;
      }
    }
    wave_lds_fence();
    if (valid) {
      if (p.T > 0) {                                        // whole tiles (padding rows as zeros): 16 rows = whole float4s
        if (compact) stage_block_out<NC * KAC>(a_t, stg, lane);
        else stage_block_out<NC * KA>(a_t, stg, lane);
      } else {
        const int na = nv * KA;
        for (int i = lane; i < na / 4; i += 64) reinterpret_cast<float4*>(a_t)[i] = reinterpret_cast<const float4*>(stg)[i];
        for (int e = (na & ~3) + lane; e < na; e += 64) a_t[e] = stg[e];
      }
    }
    wave_lds_fence();                                       // the staging block is free for Bm
    MCK();                                                  // 5: A block out
    // ---- backward -------------------------------------------------------------------------------------
    float* brow = stg + c * KB;
    float ddv = dxn * p.scale;
    if (p.tanh_output) { const float th = tanhf_(dlin); ddv *= 1.0f - th * th; }
    if (q == 0) brow[2 * G] = ddv * live;
    float dz[4][kNT], dh[kNT];
#pragma unroll
    for (int t = 0; t < kNT; ++t) dh[t] = __builtin_fmaf(ddv, w.wl[t], cdh2[t]);
    bxb::gate_grads(g2, s.c2, dh, cdc2, dz, cdc2);                       // cdc2 <- d c2(t-1)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int t = 0; t < kNT; ++t) brow[G + r * kH + 4 * t + q] = dz[r][t] * live;
    {
      f32x4 at[3];
      bxb::tgemm<3>(fr, lane, dz, at);                                   // d[h1(t) | h2(t-1)] = W2 dz2
#pragma unroll
      for (int t = 0; t < 4; ++t) { dh[t] = cdh1[t] + at[0][t]; cdh2[t] = at[1][t]; }
      dh[4] = cdh1[4] + at[2][0];
      cdh2[4] = at[2][1];                                                // cdh2 <- d h2(t-1)
    }
    MCK();                                                  // 6: layer-2 backward
    bxb::gate_grads(g1, s.c1, dh, cdc1, dz, cdc1);                       // cdc1 <- d c1(t-1)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int t = 0; t < kNT; ++t) brow[r * kH + 4 * t + q] = dz[r][t] * live;
    {
      f32x4 at[NT1];
      bxb::tgemm<NT1>(fr + bxb::frag_rel(bxb::tiles2(), 0, 0), lane, dz, at);     // d[in | h1(t-1)] = W1 dz1
      if (FC) {
#pragma unroll
        for (int t = 0; t < kNT; ++t) {
          const float dfc = t < 4 ? at[0][t] : at[NT1 - 1][0];
          brow[2 * G + 1 + 4 * t + q] = live * dfc * (pre_fc[t] > 0.0f ? 1.0f : fast_exp2(pre_fc[t] * 1.4426950408889634f));
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) cdh1[t] = at[1][t];
        cdh1[4] = at[NT1 - 1][1];
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) cdh1[t] = at[0][t];
        cdh1[4] = at[1][0];
      }
    }
    wave_lds_fence();
    MCK();                                                  // 7: layer-1 backward
    // ---- coalesced store of the Bm row block ------------------------------------------------------------
    if (valid) {
      if (p.T > 0) {
        stage_block_out<NC * KB>(b_t, stg, lane);
      } else {
        const int nb = nv * KB;
        for (int i = lane; i < nb / 4; i += 64) reinterpret_cast<float4*>(b_t)[i] = reinterpret_cast<const float4*>(stg)[i];
        for (int e = (nb & ~3) + lane; e < nb; e += 64) b_t[e] = stg[e];
      }
    }
    MCK();                                                  // 8: Bm block out
    cur = nxt;
    }                                                       // steps
    // ---- the carries into the step before the first one ----------------------------------------------------
    if (p.carry_out) {
#pragma unroll
      for (int t = 0; t < kNT; ++t) {
        cio[(0 * NC + c) * kH + 4 * t + q] = cdh1[t];
        cio[(1 * NC + c) * kH + 4 * t + q] = cdc1[t];
        cio[(2 * NC + c) * kH + 4 * t + q] = cdh2[t];
        cio[(3 * NC + c) * kH + 4 * t + q] = cdc2[t];
      }
      __syncthreads();
      if (valid) {
#pragma unroll
        for (int a4 = 0; a4 < 4; ++a4) {
          float4* cd = reinterpret_cast<float4*>(p.carry_out + ((size_t)a4 * RT + n0) * kH);
          const float4* cs = reinterpret_cast<const float4*>(cio + a4 * NC * kH);
          if (lane < nv * 5) cd[lane] = cs[lane];
          if (lane < NC * kH / 4 - 64 && 64 + lane < nv * 5) cd[64 + lane] = cs[64 + lane];
        }
      }
    }
  }
#ifdef L2O_BWD_CLOCK
  if (tid == 0 && blockIdx.x == 100) {
    printf("bwd_mfma block %d ticks:", (int)blockIdx.x);
    for (int k = 1; k < cki && k < 24; ++k) printf(" %lld", ck[k] - ck[k - 1]);
    printf("\n");
  }
#endif
#undef MCK
}

}  // namespace l2o
