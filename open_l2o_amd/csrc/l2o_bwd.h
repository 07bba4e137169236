// l2o_bwd.h -- one step of back-propagation-through-time of the coordinate-wise optimizer
// network (the meta-gradient of MetaOptimizer.meta_minimize, DM/meta.py:398-414, with the
// optimizee gradient treated as a constant: tf.stop_gradient, DM/meta.py:328-329).
// Included by l2o_kernels.hip.
//
// Kernel of the TRAINING path: one thread per coordinate, weights (Sonnet layouts) read as
// scalar loads (wave-uniform addresses -> SGPR operands), the step's forward recomputed from the
// state saved before the step and kept in registers for the backward pass.  It emits, per coordinate,
// the rows the host needs for the weight gradients as plain GEMMs over (steps x coordinates):
//     dW1 = act1^T dz1   db1 = sum dz1        dW2 = act2^T dz2   db2 = sum dz2
//     dw_lin = h2^T dd   db_lin = sum dd      (RNNProp) dW_fc = feats^T du   db_fc = sum du
// and the carries (dh1, dc1, dh2, dc2) that flow to the previous step.
#pragma once

struct BwdParams {
  int B, D, tpp;
  int pre, tanh_output, n_layers;
  float scale, k_inv, exp_k;          // LogAndSign: 1/k, e^k
  float beta1, beta2, om1, om2;       // RNNProp: 1 - beta^k of this step
  const float *wg1, *bg1, *wg2, *bg2, *wl, *bl, *wfc, *bfc;
  const float *g, *m, *v, *st_prev, *dx_next, *carry_in;
  float *carry_out, *act1, *dz1, *act2, *dz2, *h2o, *dd, *feats, *du;
  float* dg;                          // optional: dL/d(network input gradient) per coordinate (second derivatives)
  long s_act1, s_dz1, s_act2, s_dz2, s_h2, s_dd, s_feats, s_du;   // row strides (floats) of the emitted rows
  // tile kernel only: up to 8 panels that share the network in ONE launch (l2o_cwlstm_bwd_multi);
  // panel s covers the tiles [tile_end[s-1], tile_end[s]) and the rows 16 * tile of A / Bm / the carries
  int nseg;
  int tile_end[8];
  long seg_n[8];                      // coordinates of the panel
  // k_cwlstm_bwd_mfma: the panel's tiles are PER PROBLEM (the forward's packed-state layout): tile lt = problem
  // lt / seg_tpp, tile seg_tpp-th of its seg_d coordinates; a flat panel (l2o_cwlstm_bwd_multi) is one problem of seg_n
  long seg_d[8];
  int seg_tpp[8];
  const float *seg_g[8], *seg_m[8], *seg_v[8], *seg_st[8], *seg_dx[8];
  long rows_total;                    // rows of the carry arrays ([4][rows_total][20])
  const float* wpack;                 // l2o_wpack_host output (device) or NULL: selects k_cwlstm_bwd_mfma (l2o_bwd_mfma.h)
  // k_cwlstm_bwd_mfma only: T steps in one launch (l2o_cwlstm_bwd_unroll).  table[(t * nseg + s) * 5 + k] =
  // g, m, v, st_prev, dx_next (may be NULL) of panel s at step t; A / Bm hold T blocks of rows_total rows
  int T;
  const float* const* table;
  const float* seg_gfinal[8];         // dx_next == NULL: dL/d(delta_t) = g_final + sum_{tau > t} g_tau
  double pw1_last, pw2_last;          // beta^(step0 + T - 1)
  // k_cwlstm_bwd_mfma, T > 0 only (l2o_cwlstm_bwd_unroll_compact, round 5): the A rows WITHOUT their duplicated columns.
  // h1(t-1), h2(t-1) of step t are h1(t), h2(t) of step t - 1, so a row keeps [in | h1(t) | h2(t) | feats | 1] (KAc = KA - 40
  // columns) and A holds T + 1 blocks of rows_total rows: block ts + 1 = step ts, block 0 = [0 | h1, h2 before step 0 | 0].
  // The contraction reads [in | h1(t-1)], [h1(t) | h2(t-1)] from blocks ts and ts + 1 (l2o_atb.h: k_atb_bx3's column map).
  int compact_a;
};

// weights are read through the CONSTANT address space: the addresses are wave-uniform, so the
// loads become s_load_dwordx16 and the FMAs take the weight as an SGPR operand (v_fmac v, s, v)
// -- no LDS, no per-lane weight traffic.  (They are never written while the kernel runs.)
// (l2o_cfp, the constant-address-space float pointer, is declared in l2o_common.h)

__device__ __forceinline__ float bw_sig(float x) { return l2o::fast_rcp(1.0f + l2o::fast_exp2(x * -1.4426950408889634f)); }
__device__ __forceinline__ float bw_tanh(float x) {
  const float e = l2o::fast_exp2(__builtin_fabsf(x) * -2.8853900817779268f);
  return __builtin_copysignf((1.0f - e) * l2o::fast_rcp(1.0f + e), x);
}

// packed-state address of (array a, unit u) for coordinate (tile, c)
__device__ __forceinline__ size_t st_addr(size_t tile, int c, int a, int u) {
  const int q = u & 3, t = u >> 2, e = a * 5 + t;
  return tile * kStateFloatsPerTile + ((size_t)(e >> 2) * 64 + (q * 16 + c)) * 4 + (e & 3);
}

// One thread per coordinate; everything a coordinate needs lives in its registers (one wave per
// SIMD: up to 512), except the layer input vector, which is indexed by a run-time k and sits in
// a per-thread LDS column.
template <int PRE>
__global__ __launch_bounds__(64) void k_cwlstm_bwd_step(BwdParams p) {
  constexpr int P = PRE == L2O_PRE_FC_ELU ? kH : (PRE == L2O_PRE_LOGSIGN ? 2 : 1);
  constexpr int K1 = P + kH, G = 4 * kH, NT = 64;
  __shared__ float xin[2 * kH * NT];                     // [40][NT] input vector of the current GEMM (k-major)
  const l2o_cfp W1 = (l2o_cfp)p.wg1, b1 = (l2o_cfp)p.bg1, W2 = (l2o_cfp)p.wg2, b2 = (l2o_cfp)p.bg2;
  const l2o_cfp wl = (l2o_cfp)p.wl, wfc = (l2o_cfp)p.wfc, bfc = (l2o_cfp)p.bfc;
  const int tid = threadIdx.x;
  const size_t N = (size_t)p.B * p.D;
  size_t n = (size_t)blockIdx.x * NT + tid;
  const bool valid = n < N;
  if (!valid) n = N - 1;                                  // keep the wave convergent; stores are masked
  const int b = (int)(n / p.D), j = (int)(n - (size_t)b * p.D);
  const size_t tile = (size_t)b * p.tpp + j / kTile;
  const int c = j % kTile;
  float* xi = xin + tid;                                  // element k at xi[k * NT]

  // z[q] = bias[q] + sum_k in[k] W[k][q]   (weights: scalar loads)
  auto gemm = [&](l2o_cfp W, l2o_cfp bias, int KK, float (&z)[G]) {
#pragma unroll
    for (int q = 0; q < G; ++q) z[q] = bias[q];
    for (int k = 0; k < KK; ++k) {
      const float xv = xi[k * NT];
      const l2o_cfp wr = W + k * G;
#pragma unroll
      for (int q = 0; q < G; ++q) z[q] = __builtin_fmaf(xv, wr[q], z[q]);
    }
  };
  // xi[k] = sum_q dz[q] W[k][q]
  auto gemm_t = [&](l2o_cfp W, int KK, const float (&dz)[G]) {
    for (int k = 0; k < KK; ++k) {
      const l2o_cfp wr = W + k * G;
      float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
#pragma unroll
      for (int q = 0; q < G; q += 4) {
        s0 = __builtin_fmaf(dz[q], wr[q], s0);
        s1 = __builtin_fmaf(dz[q + 1], wr[q + 1], s1);
        s2 = __builtin_fmaf(dz[q + 2], wr[q + 2], s2);
        s3 = __builtin_fmaf(dz[q + 3], wr[q + 3], s3);
      }
      xi[k * NT] = (s0 + s1) + (s2 + s3);
    }
  };
  auto activate = [&](float (&z)[G]) {                    // -> (sig i, tanh j, sig(f+1), sig o)
#pragma unroll
    for (int u = 0; u < kH; ++u) {
      z[u] = bw_sig(z[u]);
      z[kH + u] = bw_tanh(z[kH + u]);
      z[2 * kH + u] = bw_sig(z[2 * kH + u] + 1.0f);
      z[3 * kH + u] = bw_sig(z[3 * kH + u]);
    }
  };

  // ---- features -------------------------------------------------------------
  const float gv = p.g[n];
  float pre_fc[PRE == L2O_PRE_FC_ELU ? kH : 1];
  float f0 = 0.0f, f1 = 0.0f;
  if (PRE == L2O_PRE_FC_ELU) {
    const float m_hat = p.m[n] / p.om1, v_hat = p.v[n] / p.om2;
    const float den = sqrtf(v_hat) + 1e-8f;
    f0 = m_hat / den;
    f1 = gv / den;
#pragma unroll
    for (int u = 0; u < kH; ++u) {
      const float zz = f0 * wfc[u] + f1 * wfc[kH + u] + bfc[u];
      pre_fc[u] = zz;
      xi[u * NT] = zz > 0.0f ? zz : expm1f(zz);
    }
  } else if (PRE == L2O_PRE_LOGSIGN) {
    xi[0] = fmaxf(logf(fabsf(gv) + 1.1920928955078125e-07f) * p.k_inv, -1.0f);
    xi[NT] = fminf(fmaxf(gv * p.exp_k, -1.0f), 1.0f);
  } else {
    xi[0] = gv;
  }
  float c1p[kH], c2p[kH];                                 // c1(t-1), c2(t-1)
#pragma unroll
  for (int u = 0; u < kH; ++u) {
    xi[(P + u) * NT] = p.st_prev[st_addr(tile, c, 0, u)]; // h1(t-1)
    c1p[u] = p.st_prev[st_addr(tile, c, 1, u)];
    c2p[u] = p.st_prev[st_addr(tile, c, 3, u)];
  }
  if (valid) {
    float* row = p.act1 + n * p.s_act1;
    for (int k = 0; k < K1; ++k) row[k] = xi[k * NT];
  }

  // ---- layer 1 forward: gates kept in registers for the backward pass ------------------
  float z1[G];
  gemm(W1, b1, K1, z1);
  activate(z1);
  float tc1[kH];
#pragma unroll
  for (int u = 0; u < kH; ++u) {
    const float c1 = z1[2 * kH + u] * c1p[u] + z1[u] * z1[kH + u];
    tc1[u] = bw_tanh(c1);
    xi[u * NT] = tc1[u] * z1[3 * kH + u];                 // h1(t): layer-2 input 0..19
    xi[(kH + u) * NT] = p.st_prev[st_addr(tile, c, 2, u)];   // h2(t-1)
  }
  if (valid) {
    float* row = p.act2 + n * p.s_act2;
    for (int k = 0; k < 2 * kH; ++k) row[k] = xi[k * NT];
  }
  // ---- layer 2 forward + backward ----------------------------------------------------------
  float z2[G];
  gemm(W2, b2, 2 * kH, z2);
  activate(z2);
  const float* cin = p.carry_in;
  {
    float tc2[kH];
    float dlin = p.bl[0];
#pragma unroll
    for (int u = 0; u < kH; ++u) {
      const float c2 = z2[2 * kH + u] * c2p[u] + z2[u] * z2[kH + u];
      tc2[u] = bw_tanh(c2);
      const float h2 = tc2[u] * z2[3 * kH + u];
      if (valid) p.h2o[n * p.s_h2 + u] = h2;
      dlin = __builtin_fmaf(h2, wl[u], dlin);
    }
    float ddv = p.dx_next[n] * p.scale;
    if (p.tanh_output) { const float th = bw_tanh(dlin); ddv *= 1.0f - th * th; }
    if (valid) p.dd[n * p.s_dd] = ddv;
#pragma unroll
    for (int u = 0; u < kH; ++u) {
      const float gi = z2[u], gj = z2[kH + u], gf = z2[2 * kH + u], go = z2[3 * kH + u];
      const float dh2 = ddv * wl[u] + cin[(2 * N + n) * kH + u];
      const float dc2 = cin[(3 * N + n) * kH + u] + dh2 * go * (1.0f - tc2[u] * tc2[u]);
      if (valid) p.carry_out[(3 * N + n) * kH + u] = dc2 * gf;
      z2[u] = dc2 * gj * gi * (1.0f - gi);
      z2[kH + u] = dc2 * gi * (1.0f - gj * gj);
      z2[2 * kH + u] = dc2 * c2p[u] * gf * (1.0f - gf);
      z2[3 * kH + u] = dh2 * tc2[u] * go * (1.0f - go);
    }
  }
  if (valid) {
    float* row = p.dz2 + n * p.s_dz2;
#pragma unroll
    for (int q = 0; q < G; ++q) row[q] = z2[q];
  }
  gemm_t(W2, 2 * kH, z2);                                 // d[h1; h2(t-1)] = dz2 . W2^T
  // ---- layer 1 backward ------------------------------------------------------------------------
#pragma unroll
  for (int u = 0; u < kH; ++u) {
    const float dh1 = xi[u * NT] + cin[(0 * N + n) * kH + u];
    if (valid) p.carry_out[(2 * N + n) * kH + u] = xi[(kH + u) * NT];
    const float gi = z1[u], gj = z1[kH + u], gf = z1[2 * kH + u], go = z1[3 * kH + u];
    const float dc1 = cin[(1 * N + n) * kH + u] + dh1 * go * (1.0f - tc1[u] * tc1[u]);
    if (valid) p.carry_out[(1 * N + n) * kH + u] = dc1 * gf;
    z1[u] = dc1 * gj * gi * (1.0f - gi);
    z1[kH + u] = dc1 * gi * (1.0f - gj * gj);
    z1[2 * kH + u] = dc1 * c1p[u] * gf * (1.0f - gf);
    z1[3 * kH + u] = dh1 * tc1[u] * go * (1.0f - go);
  }
  if (valid) {
    float* row = p.dz1 + n * p.s_dz1;
#pragma unroll
    for (int q = 0; q < G; ++q) row[q] = z1[q];
  }
  gemm_t(W1, K1, z1);                                     // d[inputs; h1(t-1)] = dz1 . W1^T
  if (valid) {
#pragma unroll
    for (int u = 0; u < kH; ++u) p.carry_out[(0 * N + n) * kH + u] = xi[(P + u) * NT];
    if (PRE != L2O_PRE_FC_ELU && p.dg) {                   // chain through the preprocessing: dL/dg
      float dgv = xi[0];
      if (PRE == L2O_PRE_LOGSIGN) {                        // DM/preprocess.py:63-70, both clamps
        const float gv = p.g[n], ag = fabsf(gv) + 1.1920928955078125e-07f;
        const float d0 = logf(ag) * p.k_inv > -1.0f ? copysignf(p.k_inv / ag, gv) : 0.0f;
        const float d1 = fabsf(gv * p.exp_k) < 1.0f ? p.exp_k : 0.0f;
        dgv = xi[0] * d0 + xi[1 * NT] * d1;
      }
      p.dg[n] = dgv;
    }
    if (PRE == L2O_PRE_FC_ELU) {
      p.feats[n * p.s_feats] = f0;
      p.feats[n * p.s_feats + 1] = f1;
#pragma unroll
      for (int u = 0; u < kH; ++u)
        p.du[n * p.s_du + u] = xi[u * NT] * (pre_fc[u] > 0.0f ? 1.0f : expf(pre_fc[u]));
    }
  }
}

// The fast form for tile-aligned panels (D % 16 == 0) and the interleaved A / Bm row layout:
// FOUR lanes per coordinate (a DPP quad): lane part p = lane & 3 owns gate type p (Sonnet column
// block i | j | f | o) of all 20 units in both layers, so a wave covers the 16 coordinates of ONE
// state tile and a 16 384-coordinate panel fills all 1 024 SIMDs (4 waves per workgroup share the
// LDS copy of the weights).  The gate GEMVs read a quarter of each weight row per lane from LDS
// (5 ds_read_b128 per 20 FMAs), the four gate types of a unit meet through quad-broadcast DPP
// moves, the transposed products are reduced over the quad with two DPP adds.  ALL global traffic
// is coalesced dwordx4 through per-wave LDS tiles: the packed state tile and the carries come in,
// the 16 x KA / 16 x KB row blocks of A / Bm and the carries go out as contiguous 5-10 KB pieces
// (the per-lane scattered form was bound by the texture addresser: ~350 VMEM instructions of
// 16-64 cache lines each per wave).
template <int CTRL>
__device__ __forceinline__ float bw_quad_bcast(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}

// k-loop unrolling of the tile kernel's GEMVs (measured: see DESIGN.md 3)
#ifndef L2O_BWD_UNROLL
#define L2O_BWD_UNROLL 1
#endif
#if L2O_BWD_UNROLL == 1
#define L2O_BWD_UNROLL_PRAGMA _Pragma("nounroll")
#elif L2O_BWD_UNROLL == 0
#define L2O_BWD_UNROLL_PRAGMA _Pragma("unroll")
#else
#define L2O_BWD_UNROLL_PRAGMA _Pragma("unroll 4")
#endif

template <int PRE>
struct BwdTileGeom {
  static constexpr int P = PRE == L2O_PRE_FC_ELU ? kH : (PRE == L2O_PRE_LOGSIGN ? 2 : 1);
  static constexpr int K1 = P + kH, G = 4 * kH, NC = 16;
  static constexpr int KA = K1 + 3 * kH + (PRE == L2O_PRE_FC_ELU ? 2 : 0) + 1;   // act1 | act2 | h2 | feats | 1
  static constexpr int KAC = KA - 2 * kH;                                        // compact: in | h1 | h2 | feats | 1
  static constexpr int KB = 2 * G + 1 + (PRE == L2O_PRE_FC_ELU ? kH : 0);        // dz1 | dz2 | dd | du
  static constexpr int kWaveFloats = 2 * kH * NC + kStateFloatsPerTile + 4 * NC * kH + NC * KA + NC * KB;
  static constexpr int kLdsFloats = K1 * G + 2 * kH * G + 4 * kWaveFloats;
};

template <int PRE>
__global__ __launch_bounds__(256) void k_cwlstm_bwd_tile(BwdParams p) {
#ifdef L2O_BWD_CLOCK
  long long ck[12]; int cki = 0;
#define BCK() do { __syncthreads(); ck[cki++] = __builtin_readcyclecounter(); } while (0)
#else
#define BCK() ((void)0)
#endif
  using Geo = BwdTileGeom<PRE>;
  constexpr int P = Geo::P, K1 = Geo::K1, G = Geo::G, NC = Geo::NC, KA = Geo::KA, KB = Geo::KB;
  extern __shared__ float sm[];
  float* W1 = sm;                 // [K1][80]
  float* W2 = W1 + K1 * G;        // [40][80]
  const int tid = threadIdx.x, lane = tid & 63, part = lane & 3, cl = lane >> 2;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: the panel lookup below stays on the scalar unit
  float* wbase = W2 + 2 * kH * G + wv * Geo::kWaveFloats;
  float* xin = wbase;                         // [40][NC]  input vector of the current GEMV, k-major
  float* stt = xin + 2 * kH * NC;             // the packed state tile (before the step)
  float* cio = stt + kStateFloatsPerTile;     // [4][NC][20] carries in, overwritten by carries out
  float* At = cio + 4 * NC * kH;              // [NC][KA]
  float* Bt = At + NC * KA;                   // [NC][KB]
  {
    // stage both weight matrices with all loads of a thread in flight at once (a one-load-per-
    // iteration loop pays one L2 latency per 16 bytes)
    auto stage = [&](float* dst, const float* src, int n4) {
      for (int base = tid; base < n4; base += 256 * 8) {
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int i = base + 256 * q;
          v[q] = i < n4 ? reinterpret_cast<const float4*>(src)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int i = base + 256 * q;
          if (i < n4) reinterpret_cast<float4*>(dst)[i] = v[q];
        }
      }
    };
    stage(W1, p.wg1, K1 * G / 4);
    stage(W2, p.wg2, 2 * kH * G / 4);
  }
  const l2o_cfp wl = (l2o_cfp)p.wl, wfc = (l2o_cfp)p.wfc, bfc = (l2o_cfp)p.bfc;
  // persistent: a workgroup walks tile groups blockIdx, blockIdx + gridDim, ... (the weights are
  // staged once and the long straight-line body stays in the instruction cache)
  const size_t ntiles = (size_t)p.tile_end[p.nseg - 1];
  const size_t ngrp4 = (ntiles + 3) / 4;                  // groups of 4 tiles
  const size_t RT = (size_t)p.rows_total;
  for (size_t g4 = blockIdx.x; g4 < ngrp4; g4 += gridDim.x) {
  __syncthreads();                                        // the previous group's LDS tiles are fully stored
  const size_t grp = g4 * 4 + wv;                         // global tile index == row block of A / Bm / carries
  const bool valid = grp < ntiles;
  // panel of this tile (wave-uniform)
  int sg = 0;
  while (sg + 1 < p.nseg && (int)grp >= p.tile_end[sg]) ++sg;
  const size_t lt = valid ? grp - (sg ? p.tile_end[sg - 1] : 0) : 0;     // tile inside the panel
  const size_t N = (size_t)p.seg_n[sg];                   // D % 16 == 0, or one flat row (B == 1) with a ragged last tile
  const float* seg_g = p.seg_g[sg];
  const float* seg_m = p.seg_m[sg];
  const float* seg_v = p.seg_v[sg];
  const float* seg_dx = p.seg_dx[sg];
  const size_t n0 = valid ? grp * NC : 0;                 // global row
  const size_t ln0 = lt * NC;                             // first coordinate inside the panel
  const int nv = valid ? (int)(N - ln0 < (size_t)NC ? N - ln0 : (size_t)NC) : 0;   // coordinates of this tile that exist
  const size_t n = ln0 + cl < N ? ln0 + cl : N - 1;       // panel coordinate (tail lanes recompute the last one; nothing of theirs is stored)
  const int c = cl;
  float* xi = xin + cl;                                   // element k at xi[k * NC]
  // ---- coalesced loads: the state tile and the four carry blocks of these 16 coordinates ----
  {
    const float4* src = reinterpret_cast<const float4*>(p.seg_st[sg] + lt * kStateFloatsPerTile);
#pragma unroll
    for (int q = 0; q < 5; ++q) reinterpret_cast<float4*>(stt)[q * 64 + lane] = src[q * 64 + lane];
#pragma unroll
    for (int a4 = 0; a4 < 4; ++a4) {
      const float4* cs = reinterpret_cast<const float4*>(p.carry_in + ((size_t)a4 * RT + n0) * kH);
      float4* cd = reinterpret_cast<float4*>(cio + a4 * NC * kH);
      const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
      cd[lane] = lane < nv * 5 ? cs[lane] : zero4;
      if (lane < NC * kH / 4 - 64) cd[64 + lane] = 64 + lane < nv * 5 ? cs[64 + lane] : zero4;
    }
  }
  auto st_at = [&](int a, int u) {                        // (array a, unit u) of this lane's coordinate
    const int qq = u & 3, tt = u >> 2, e = a * 5 + tt;
    return stt[((e >> 2) * 64 + (qq * 16 + c)) * 4 + (e & 3)];
  };
  float* arow = At + cl * KA;
  float* brow = Bt + cl * KB;
  if (part == 0) arow[KA - 1] = 1.0f;
  // this lane's activation: a * sigmoid(s x + s0) + c0  (tanh x = 2 sigmoid(2x) - 1; forget_bias = 1)
  const float act_s = part == 1 ? 2.0f : 1.0f, act_s0 = part == 2 ? 1.0f : 0.0f;
  const float act_a = part == 1 ? 2.0f : 1.0f, act_c0 = part == 1 ? -1.0f : 0.0f;

  // z[u] = bias[part*20 + u] + sum_k in[k] W[k][part*20 + u]
  auto gemm = [&](const float* W, const float* bias, auto kk_c, float (&z)[kH]) {
    constexpr int KK = decltype(kk_c)::value;
#pragma unroll
    for (int u = 0; u < kH; ++u) z[u] = bias[part * kH + u];
    const float* wp = W + part * kH;
    L2O_BWD_UNROLL_PRAGMA
    for (int k = 0; k < KK; ++k) {
      const float xv = xi[k * NC];
      const float4* wr = reinterpret_cast<const float4*>(wp + k * G);
#pragma unroll
      for (int u4 = 0; u4 < kH / 4; ++u4) {
        const float4 w4 = wr[u4];
        z[4 * u4 + 0] = __builtin_fmaf(xv, w4.x, z[4 * u4 + 0]);
        z[4 * u4 + 1] = __builtin_fmaf(xv, w4.y, z[4 * u4 + 1]);
        z[4 * u4 + 2] = __builtin_fmaf(xv, w4.z, z[4 * u4 + 2]);
        z[4 * u4 + 3] = __builtin_fmaf(xv, w4.w, z[4 * u4 + 3]);
      }
    }
#pragma unroll
    for (int u = 0; u < kH; ++u) z[u] = __builtin_fmaf(act_a, bw_sig(__builtin_fmaf(act_s, z[u], act_s0)), act_c0);
  };
  // xi[k] = sum over the quad of sum_u dz[u] W[k][part*20 + u]   (every lane of the quad gets it)
  auto gemm_t = [&](const float* W, auto kk_c, const float (&dz)[kH]) {
    constexpr int KK = decltype(kk_c)::value;
    const float* wp = W + part * kH;
    L2O_BWD_UNROLL_PRAGMA
    for (int k = 0; k < KK; ++k) {
      const float4* wr = reinterpret_cast<const float4*>(wp + k * G);
      float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
      for (int u4 = 0; u4 < kH / 4; ++u4) {
        const float4 w4 = wr[u4];
        s0 = __builtin_fmaf(dz[4 * u4 + 0], w4.x, s0);
        s1 = __builtin_fmaf(dz[4 * u4 + 1], w4.y, s1);
        s0 = __builtin_fmaf(dz[4 * u4 + 2], w4.z, s0);
        s1 = __builtin_fmaf(dz[4 * u4 + 3], w4.w, s1);
      }
      const float tot = l2o::quad_sum(s0 + s1);
      if (part == (k & 3)) xi[k * NC] = tot;
    }
  };
  // the four gates of every unit, in every lane of the quad
  auto gather = [&](const float (&z)[kH], float (&gi)[kH], float (&gj)[kH], float (&gf)[kH], float (&go)[kH]) {
#pragma unroll
    for (int u = 0; u < kH; ++u) {
      gi[u] = bw_quad_bcast<0x00>(z[u]);   // quad_perm:[0,0,0,0]
      gj[u] = bw_quad_bcast<0x55>(z[u]);   // [1,1,1,1]
      gf[u] = bw_quad_bcast<0xAA>(z[u]);   // [2,2,2,2]
      go[u] = bw_quad_bcast<0xFF>(z[u]);   // [3,3,3,3]
    }
  };
  BCK();
  __syncthreads();                                          // weights staged
  BCK();

  // ---- features (every lane of the quad computes them; lane part == k & 3 owns row k) ------
  const float gv = seg_g[n];
  float pre_fc[PRE == L2O_PRE_FC_ELU ? kH : 1];
  float f0 = 0.0f, f1 = 0.0f;
  if (PRE == L2O_PRE_FC_ELU) {
    const float m_hat = seg_m[n] / p.om1, v_hat = seg_v[n] / p.om2;
    const float den = sqrtf(v_hat) + 1e-8f;
    f0 = m_hat / den;
    f1 = gv / den;
#pragma unroll
    for (int u = 0; u < kH; ++u) {
      const float zz = f0 * wfc[u] + f1 * wfc[kH + u] + bfc[u];
      pre_fc[u] = zz;
      if (part == (u & 3)) xi[u * NC] = zz > 0.0f ? zz : expm1f(zz);
    }
  } else if (PRE == L2O_PRE_LOGSIGN) {
    if (part == 0) xi[0] = fmaxf(logf(fabsf(gv) + 1.1920928955078125e-07f) * p.k_inv, -1.0f);
    if (part == 1) xi[NC] = fminf(fmaxf(gv * p.exp_k, -1.0f), 1.0f);
  } else {
    if (part == 0) xi[0] = gv;
  }
  float c1p[kH], c2p[kH];                                 // c1(t-1), c2(t-1)
#pragma unroll
  for (int u = 0; u < kH; ++u) {
    if (part == ((P + u) & 3)) xi[(P + u) * NC] = st_at(0, u);   // h1(t-1)
    c1p[u] = st_at(1, u);
    c2p[u] = st_at(3, u);
  }
  __syncthreads();
  for (int k = part; k < K1; k += 4) arow[k] = xi[k * NC];

  BCK();
  // ---- layer 1 forward -------------------------------------------------------------------
  float z1[kH];
  gemm(W1, p.bg1, std::integral_constant<int, K1>{}, z1);
  BCK();
  float gi1[kH], gj1[kH], gf1[kH], go1[kH], tc1[kH];
  gather(z1, gi1, gj1, gf1, go1);
  __syncthreads();                                          // everyone is done reading the layer-1 input
#pragma unroll
  for (int u = 0; u < kH; ++u) {
    const float c1 = gf1[u] * c1p[u] + gi1[u] * gj1[u];
    tc1[u] = bw_tanh(c1);
    if (part == (u & 3)) {
      xi[u * NC] = tc1[u] * go1[u];                            // h1(t): layer-2 input 0..19
      xi[(kH + u) * NC] = st_at(2, u);   // h2(t-1)
    }
  }
  __syncthreads();
  for (int k = part; k < 2 * kH; k += 4) arow[K1 + k] = xi[k * NC];
  // ---- layer 2 forward + backward ----------------------------------------------------------
  BCK();
  float z2[kH];
  gemm(W2, p.bg2, std::integral_constant<int, 2 * kH>{}, z2);
  BCK();
  const float* cin0 = cio + (0 * NC + cl) * kH;   // this coordinate's carries dh1, dc1, dh2, dc2
  float* const cin1 = cio + (1 * NC + cl) * kH;
  float* const cin2 = cio + (2 * NC + cl) * kH;
  float* const cin3 = cio + (3 * NC + cl) * kH;
  {
    float gi[kH], gj[kH], gf[kH], go[kH];
    gather(z2, gi, gj, gf, go);
    float tc2[kH];
    float dlin = p.bl[0];
#pragma unroll
    for (int u = 0; u < kH; ++u) {
      const float c2 = gf[u] * c2p[u] + gi[u] * gj[u];
      tc2[u] = bw_tanh(c2);
      const float h2 = tc2[u] * go[u];
      if (part == (u & 3)) arow[K1 + 2 * kH + u] = h2;
      dlin = __builtin_fmaf(h2, wl[u], dlin);
    }
    float ddv = seg_dx[n] * p.scale;
    if (p.tanh_output) { const float th = bw_tanh(dlin); ddv *= 1.0f - th * th; }
    if (part == 0) brow[2 * G] = ddv;
#pragma unroll
    for (int u = 0; u < kH; ++u) {
      const float dh2 = ddv * wl[u] + cin2[u];
      const float dc2 = cin3[u] + dh2 * go[u] * (1.0f - tc2[u] * tc2[u]);
      if (part == (u & 3)) cin3[u] = dc2 * gf[u];             // carry out (all four lanes have read it: one wave)
      const float d_i = dc2 * gj[u] * gi[u] * (1.0f - gi[u]);
      const float d_j = dc2 * gi[u] * (1.0f - gj[u] * gj[u]);
      const float d_f = dc2 * c2p[u] * gf[u] * (1.0f - gf[u]);
      const float d_o = dh2 * tc2[u] * go[u] * (1.0f - go[u]);
      z2[u] = part == 0 ? d_i : (part == 1 ? d_j : (part == 2 ? d_f : d_o));
    }
  }
#pragma unroll
  for (int u = 0; u < kH; ++u) brow[G + part * kH + u] = z2[u];
  __syncthreads();                                          // layer-2 input fully consumed
  BCK();
  gemm_t(W2, std::integral_constant<int, 2 * kH>{}, z2);                                 // d[h1; h2(t-1)] = dz2 . W2^T
  BCK();
  __syncthreads();
  // ---- layer 1 backward ------------------------------------------------------------------------
#pragma unroll
  for (int u = 0; u < kH; ++u) {
    const float dh1 = xi[u * NC] + cin0[u];
    if (part == (u & 3)) cin2[u] = xi[(kH + u) * NC];
    const float dc1 = cin1[u] + dh1 * go1[u] * (1.0f - tc1[u] * tc1[u]);
    if (part == (u & 3)) cin1[u] = dc1 * gf1[u];
    const float d_i = dc1 * gj1[u] * gi1[u] * (1.0f - gi1[u]);
    const float d_j = dc1 * gi1[u] * (1.0f - gj1[u] * gj1[u]);
    const float d_f = dc1 * c1p[u] * gf1[u] * (1.0f - gf1[u]);
    const float d_o = dh1 * tc1[u] * go1[u] * (1.0f - go1[u]);
    z1[u] = part == 0 ? d_i : (part == 1 ? d_j : (part == 2 ? d_f : d_o));
  }
#pragma unroll
  for (int u = 0; u < kH; ++u) brow[part * kH + u] = z1[u];
  __syncthreads();
  BCK();
  gemm_t(W1, std::integral_constant<int, K1>{}, z1);                                     // d[inputs; h1(t-1)] = dz1 . W1^T
  BCK();
  __syncthreads();
  {
    float* c0w = cio + (0 * NC + cl) * kH;
#pragma unroll
    for (int u = 0; u < kH; ++u)
      if (part == (u & 3)) c0w[u] = xi[(P + u) * NC];
    if (PRE == L2O_PRE_FC_ELU) {
      if (part == 0) {
        arow[K1 + 3 * kH] = f0;
        arow[K1 + 3 * kH + 1] = f1;
      }
#pragma unroll
      for (int u = 0; u < kH; ++u)
        if (part == (u & 3)) brow[2 * G + 1 + u] = xi[u * NC] * (pre_fc[u] > 0.0f ? 1.0f : expf(pre_fc[u]));
    }
  }
  __syncthreads();
  BCK();
  // ---- coalesced stores: the A / Bm row blocks and the carries of these 16 coordinates ---------
  if (valid) {
    const int na = nv * KA, nb = nv * KB;                 // floats of the row blocks that exist
    float* ag = p.act1 + n0 * KA;
    for (int i = lane; i < na / 4; i += 64) reinterpret_cast<float4*>(ag)[i] = reinterpret_cast<const float4*>(At)[i];
    for (int e = (na & ~3) + lane; e < na; e += 64) ag[e] = At[e];
    float* bg = p.dz1 + n0 * KB;
    for (int i = lane; i < nb / 4; i += 64) reinterpret_cast<float4*>(bg)[i] = reinterpret_cast<const float4*>(Bt)[i];
    for (int e = (nb & ~3) + lane; e < nb; e += 64) bg[e] = Bt[e];
#pragma unroll
    for (int a4 = 0; a4 < 4; ++a4) {
      float4* cd = reinterpret_cast<float4*>(p.carry_out + ((size_t)a4 * RT + n0) * kH);
      const float4* cs = reinterpret_cast<const float4*>(cio + a4 * NC * kH);
      if (lane < nv * 5) cd[lane] = cs[lane];
      if (lane < NC * kH / 4 - 64 && 64 + lane < nv * 5) cd[64 + lane] = cs[64 + lane];
    }
  }
  }                                                       // tile groups
#ifdef L2O_BWD_CLOCK
  BCK();
  if (tid == 0 && blockIdx.x == 0) {
    printf("bwd_tile ticks:");
    for (int q = 1; q < cki; ++q) printf(" %lld", ck[q] - ck[q - 1]);
    printf("\n");
  }
#endif
#undef BCK
}

// layers == (): delta = scale * (tanh)(w . feats + b): emit feats rows and dd
template <int PRE>
__global__ void k_linear_bwd_step(BwdParams p) {
  const size_t n = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t N = (size_t)p.B * p.D;
  if (n >= N) return;
  const float gv = p.g[n];
  float a0 = gv, a1 = 0.0f;
  if (PRE == L2O_PRE_LOGSIGN) {
    a0 = fmaxf(logf(fabsf(gv) + 1.1920928955078125e-07f) * p.k_inv, -1.0f);
    a1 = fminf(fmaxf(gv * p.exp_k, -1.0f), 1.0f);
  }
  p.act1[n * p.s_act1] = a0;
  p.act1[n * p.s_act1 + 1] = a1;
  float ddv = p.dx_next[n] * p.scale;
  if (p.tanh_output) {
    const float th = tanhf(a0 * p.wl[0] + a1 * (PRE == L2O_PRE_LOGSIGN ? p.wl[1] : 0.0f) + p.bl[0]);
    ddv *= 1.0f - th * th;
  }
  p.dd[n * p.s_dd] = ddv;
  if (p.dg) {                                             // dL/dg through Linear and the preprocessing
    float dgv = ddv * p.wl[0];
    if (PRE == L2O_PRE_LOGSIGN) {
      const float ag = fabsf(gv) + 1.1920928955078125e-07f;
      const float d0 = logf(ag) * p.k_inv > -1.0f ? copysignf(p.k_inv / ag, gv) : 0.0f;
      const float d1 = fabsf(gv * p.exp_k) < 1.0f ? p.exp_k : 0.0f;
      dgv = ddv * (p.wl[0] * d0 + p.wl[1] * d1);
    }
    p.dg[n] = dgv;
  }
}
