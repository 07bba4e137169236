// l2o_unroll_cu.h -- the fused persistent unroll for optimizees that do NOT fit a CU's LDS
// (D <= 512 with D % 4 == 0, any M: BASELINE config 3 = Lasso 256 x 512 per problem).  Included by
// l2o_kernels.hip after k_unroll; written for gfx950 only.
//
// Replaces, for these sizes, the step-granular pair {l2o_problem_fg, l2o_cwlstm_step} x T
// (DM/meta.py:338-359 time_step; RNNProp fork DM/meta_rnnprop_train.py:397-423) with ONE
// launch: one workgroup (4 waves, one per SIMD, 512 registers per lane) per problem, T steps.
//
//   LDS   : the LSTM state of the problem's coordinates (5 KB per 16-coordinate tile, packed
//           layout of l2o_common.h) for up to 7 tiles per wave -- 140 KB of the 160 KB for
//           D = 512 -- plus x, x*s, s, m, v and the 4 partial gradients.  The state of a
//           wave's EIGHTH tile (D > 448) lives in its registers (a second inlined copy of the
//           tile body): 32 tiles x 5 KB is exactly the whole LDS.
//   VGPR  : the bf16x3 optimizer weights (loop-invariant), the row groups of the matrix in flight
//   HBM   : the matrix streamed ONCE per step (k_problem_fg1's scheme: a wave keeps the rows it
//           visits in registers between r_i = <row, x> - y_i and g += r_i row), nothing else:
//           x / state / m / v never leave the CU between steps (the step-granular path moves
//           664 B per coordinate-step of them through HBM, and pays 2 launches per step).
//           A wave keeps 2-3 row groups (16-24 KB) in flight while it reduces another; the first
//           groups of step t+1 are requested at the end of the optimizer phase of step t, ahead of
//           the barrier (which waits for LDS traffic only), so the stream's start-up latency
//           overlaps the barrier and the GEMV prologue.
// Per step:  GEMV phase (4 waves x M/4 rows) -> partial g, partial f -> barrier ->
//            optimizer phase (wave w: tiles w, w+4, ...) -> x, x*s in LDS -> barrier.
#pragma once

namespace l2o {

constexpr int kCuWaves = 4;
constexpr int kCuThreads = 64 * kCuWaves;
constexpr int kCuMaxLdsSlots = 7;            // tile slots per wave whose state is LDS-resident
constexpr int kCuSlotF4 = 5 * 64;            // float4 per tile slot (packed tile state)
// Row groups (4 rows each) per wave in the GEMV ring.  RNNProp with two chunks per lane (D > 256) has 240
// fragment registers + a 32-register group: a fourth group spills (measured: 4.29 vs 4.70 G on config 3).
#ifndef L2O_CU_RING_RNNPROP2
#define L2O_CU_RING_RNNPROP2 3
#endif
constexpr int cu_ring(int pre, int nv) { return (pre == L2O_PRE_FC_ELU && nv == 2) ? L2O_CU_RING_RNNPROP2 : 4; }

struct UnrollCuLayout { int tpp, nslots, nlds, DP; size_t lds; };
static inline UnrollCuLayout unroll_cu_layout(int D) {
  UnrollCuLayout L;
  L.tpp = (D + kTile - 1) / kTile;
  L.nslots = (L.tpp + kCuWaves - 1) / kCuWaves;
  L.nlds = L.nslots < kCuMaxLdsSlots ? L.nslots : kCuMaxLdsSlots;
  L.DP = L.tpp * kTile;
  L.lds = (size_t)kCuWaves * L.nlds * kCuSlotF4 * 16 + sizeof(float) * ((size_t)kCuWaves * D + 5 * (size_t)L.DP + 8);
  return L;
}

// HIST: also record what back-propagation through time needs (l2o_unroll_record): the packed state BEFORE each
// step, the gradient fed to the network, RNNProp's moments after the step, and the gradient at x_T.
template <int PRE, int NV, bool HIST>
__global__ __launch_bounds__(kCuThreads) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_unroll_cu(UnrollArgs a) {
  constexpr int kCuRing = cu_ring(PRE, NV);
  extern __shared__ float4 cu_smem[];
  const ProbParams& pp = a.pp;
  const int D = pp.D, M = pp.M;
  const int tpp = (D + kTile - 1) / kTile;
  const int nslots = (tpp + kCuWaves - 1) / kCuWaves;
  const int nlds = nslots < kCuMaxLdsSlots ? nslots : kCuMaxLdsSlots;
  const int DP = tpp * kTile;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const int b = blockIdx.x;
  float4* stL = cu_smem + (size_t)wv * nlds * kCuSlotF4;                 // this wave's tile slots
  float* part = reinterpret_cast<float*>(cu_smem + (size_t)kCuWaves * nlds * kCuSlotF4);   // [4][D]
  float* xL = part + kCuWaves * D;       // [DP] x
  float* xsL = xL + DP;                  // [DP] x * s (what the optimizee sees)
  float* scL = xsL + DP;                 // [DP] s
  float* mL = scL + DP;                  // [DP] RNNProp moments
  float* vL = mL + DP;
  float* red = vL + DP;                  // [4]

  const int kind = pp.kind;
  const bool kCos = kind == L2O_PROB_RASTRIGIN || kind == L2O_PROB_SQUARE_COS;
  const bool kSq = kind == L2O_PROB_QUADRATIC || kind == L2O_PROB_SQUARE_COS;
  const float coef = kSq ? 1.0f : 0.5f;
  const float cg = kSq ? 2.0f : 1.0f;
  const float* Wb = pp.W + (pp.w_shared ? (size_t)0 : (size_t)b * M * D);
  const l2o_cfp yb = (l2o_cfp)(pp.y + (size_t)b * M);
  const float* Cb = kCos ? pp.C + (size_t)b * D : nullptr;

  // ---- the matrix stream: groups of four rows, wave w owns groups w, w + 4, ... ------------
  constexpr int kRowStep = 4 * kCuWaves;
  // lanes beyond the last column (D < 256 NV) read column 0 instead of branching around the load:
  // their x chunk is zero and their partial gradient is never stored
  int jcol[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) jcol[v] = 4 * (64 * v + lane) < D ? 4 * (64 * v + lane) : 0;
  auto load4 = [&](int i0, float4 (&w4)[4][NV]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = i0 + k < M ? i0 + k : M - 1;             // clamped rows contribute r = 0 below
      // the row offset is wave-uniform; passing it through an opaque scalar keeps the compiler from
      // hoisting one 64-bit per-lane address per (row group, chunk) out of the step loop and spilling them
      unsigned long long ro = (unsigned long long)i * (unsigned)D;
      asm volatile("" : "+s"(ro));
      const float* rowp = Wb + ro;
#pragma unroll
      for (int v = 0; v < NV; ++v) w4[k][v] = *reinterpret_cast<const float4*>(rowp + jcol[v]);
    }
  };
  // the ring is live in the GEMV phase only (and across barrier B2), where the registers of the network's
  // temporaries are free
  float4 wa[4][NV], wb[4][NV], wc[4][NV], wd[4][NV];
  const int i_first = 4 * wv;
  // (round 4: every conditional refill has a zeroing else-branch.  Without it a ring array is a phi with its OLD values on
  //  the not-taken path and stays live through the whole optimizer phase -- what made k_unroll_cu8 spill 100 registers,
  //  DESIGN.md 3.1c)
  auto zero4 = [&](float4 (&w4)[4][NV]) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int v = 0; v < NV; ++v) w4[k][v] = float4{0.f, 0.f, 0.f, 0.f};
  };
  auto load4z = [&](int i0, float4 (&w4)[4][NV]) {
    if (i0 < M) load4(i0, w4); else zero4(w4);
  };
  auto load_head = [&]() {                 // the ring's groups of a step's first trip
    load4z(i_first, wa);
    load4z(i_first + kRowStep, wb);
    if (kCuRing >= 3) load4z(i_first + 2 * kRowStep, wc);
    if (kCuRing >= 4) load4z(i_first + 3 * kRowStep, wd);
  };
  load_head();

  // ---- problem vectors and LSTM state into LDS ----------------------------------------------
  for (int j = tid; j < DP; j += kCuThreads) {
    const bool live = j < D;
    const size_t idx = (size_t)b * D + (live ? j : D - 1);
    const float xv = live ? (a.x_in ? a.x_in : a.x)[idx] : 0.0f;
    const float sc = (live && pp.x_scale) ? pp.x_scale[idx] : 1.0f;
    xL[j] = xv; scL[j] = sc; xsL[j] = xv * sc;
    mL[j] = (PRE == L2O_PRE_FC_ELU && live && !a.zero_state) ? a.m[idx] : 0.0f;
    vL[j] = (PRE == L2O_PRE_FC_ELU && live && !a.zero_state) ? a.v[idx] : 0.0f;
  }
  float* st_b = a.st + (size_t)b * tpp * kStateFloatsPerTile;
  for (int k = 0; k < nlds; ++k) {
    const int tile = wv + kCuWaves * k;
    if (tile < tpp) {
      const float4* src = reinterpret_cast<const float4*>(st_b + (size_t)tile * kStateFloatsPerTile);
#pragma unroll
      for (int jj = 0; jj < 5; ++jj)
        stL[k * kCuSlotF4 + jj * 64 + lane] = a.zero_state ? float4{0.f, 0.f, 0.f, 0.f} : src[jj * 64 + lane];
    }
  }
  const int tile7 = wv + kCuWaves * kCuMaxLdsSlots;            // the register-resident eighth tile
  const bool has7 = nslots > kCuMaxLdsSlots && tile7 < tpp;    // wave-uniform
  TileState s7;
#pragma unroll
  for (int t5 = 0; t5 < kNT; ++t5) { s7.h1[t5] = 0.f; s7.c1[t5] = 0.f; s7.h2[t5] = 0.f; s7.c2[t5] = 0.f; }
  if (has7 && !a.zero_state) load_tile_state(s7, st_b + (size_t)tile7 * kStateFloatsPerTile, lane);

  // packed gate GEMM (80 fragment registers per chunk) only beside ONE column block of the problem: with two, or with
  // RNNProp's four chunks, the fragments spill (config 3: 4.8 -> 2.8 G coordinate-steps/s)
  constexpr bool PK = NV == 1 && bx::packed_default(PRE);
  bx::NetWB<PRE, PK> w;
  bx::load_netw<PRE, true, PK>(w, a.np.wpack, lane);
  __shared__ __attribute__((aligned(16))) float bias_s[bx::kBiasWords];     // the gate biases = accumulator inits
  bx::stage_bias(bias_s, a.np.wpack, PRE, tid, blockDim.x);               // (ordered by the __syncthreads() below)
  bx::set_bias(w, bias_s, q);
  // pin the 180-240 fragment registers to the accumulation half of the register file (MFMA reads its A
  // operand from there directly): the architectural VGPRs stay free for the row ring and the gate math.
  // Left to itself the allocator spreads the fragments over both halves and spills the ring.
#pragma unroll
  for (int ch = 0; ch < bx::NetWB<PRE, PK>::NCH; ++ch)
#pragma unroll
    for (int t5 = 0; t5 < kNT; ++t5)
#pragma unroll
      for (int s3 = 0; s3 < bx::frags(PK); ++s3) asm volatile("" : "+a"(w.a[ch][t5][s3]));
  float p1h = a.p1_hi, p1l = a.p1_lo, p2h = a.p2_hi, p2l = a.p2_lo;
  float om1 = 1.0f, om2 = 1.0f;
  __syncthreads();

  int tcur = 0;                                             // the step being computed (history slots)
  const size_t hist_n = (size_t)pp.B_local * D;
  // the gradient of coordinate j = tile * 16 + c from the four waves' partial sums
  auto grad_of = [&](int tile, float& xsv_out) {
    const int j = tile * kTile + c;
    const bool live = j < D;
    const int jc = live ? j : D - 1;
    float sum = part[jc];
#pragma unroll
    for (int p = 1; p < kCuWaves; ++p) sum += part[p * D + jc];
    const float xsv = xsL[j], sc = scL[j];
    float gj = cg * sum;
    if (kind == L2O_PROB_LASSO) gj += pp.l1 * (xsv > 0.f ? 1.f : (xsv < 0.f ? -1.f : 0.f));
    if (kCos) gj += pp.twopi * pp.alpha * Cb[jc] * l2o::sin_f(pp.twopi * xsv);
    xsv_out = xsv;
    return live ? gj * pp.inv_bg * sc : 0.0f;
  };
  // one optimizer step for a 16-coordinate tile whose state is in `s`
  auto do_tile = [&](int tile, TileState& s) {
    const int j = tile * kTile + c;
    const bool live = j < D;
    float xsv;
    const float gv = grad_of(tile, xsv);
    const float sc = scL[j], xj = xL[j];
    if (HIST) {
      store_tile_state(s, a.hist_st + (((size_t)tcur * pp.B_local + b) * tpp + tile) * kStateFloatsPerTile, lane);
      if (live && q == 0) a.hist_g[(size_t)tcur * hist_n + (size_t)b * D + j] = gv;
    }
    float in0, in1;
    if (PRE == L2O_PRE_FC_ELU) {
      float m = mL[j], v = vL[j];
      rnnprop_inputs(gv, m, v, a.np.beta1, a.np.beta2, a.np.omb1, a.np.omb2, om1, om2, in0, in1);
      if (!live) { in0 = 0.0f; in1 = 0.0f; }
      if (q == 0) { mL[j] = m; vL[j] = v; }
      if (HIST && live && q == 0) {
        a.hist_m[(size_t)tcur * hist_n + (size_t)b * D + j] = m;
        a.hist_v[(size_t)tcur * hist_n + (size_t)b * D + j] = v;
      }
    } else {
      preprocess_grad<PRE>(gv, a.np.k_inv_ln2, a.np.exp_k, in0, in1);
    }
    float d = bx::tile_step<PRE>(w, s, in0, in1, q);
    if (a.np.tanh_output) d = tanhf_(d);
    d *= a.np.scale;
    const float xn = xj + d;
    if (live && q == 0) { xL[j] = xn; xsL[j] = xn * sc; }
  };

  // Stagger (round 4): every workgroup alternates a matrix stream (HBM-bound when all stream at once) with an optimizer
  // phase that moves no bytes; launched together, the 256 workgroups stay roughly in phase and HBM idles while they compute.
  // Odd workgroups start L2O_CU_STAGGER x 3.4 us late, so that one half streams while the other half computes.
#ifndef L2O_CU_STAGGER
#define L2O_CU_STAGGER 0
#endif
  if (L2O_CU_STAGGER > 0 && (b & 1)) {
#pragma unroll 1
    for (int i = 0; i < L2O_CU_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
  }
  for (int t = 0;; ++t) {
    const bool want_g = t < a.T || HIST;                      // (history mode: the gradient at x_T is recorded too)
    tcur = t;
    // ---- optimizee: f_b(x s) and the partial gradients of this wave's rows -------------------
    float4 xv[NV], ga[NV];
    float facc = 0.0f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int j = 4 * (64 * v + lane);
      const float4 zero = {0.f, 0.f, 0.f, 0.f};
      xv[v] = zero; ga[v] = zero;
      if (j < D) {
        xv[v] = *reinterpret_cast<const float4*>(xsL + j);
        if (wv == 0) {                                        // the separable terms of f: once per problem
          const float xe[4] = {xv[v].x, xv[v].y, xv[v].z, xv[v].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (kind == L2O_PROB_LASSO) facc += pp.l1 * __builtin_fabsf(xe[e]);
            if (kCos) facc += pp.alpha - pp.alpha * Cb[j + e] * l2o::cos_f(pp.twopi * xe[e]);
          }
        }
      }
    }
    auto use4 = [&](int i0, const float4 (&w4)[4][NV]) {
      float acc[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float s0 = 0.0f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          s0 = __builtin_fmaf(w4[k][v].x, xv[v].x, s0);
          s0 = __builtin_fmaf(w4[k][v].y, xv[v].y, s0);
          s0 = __builtin_fmaf(w4[k][v].z, xv[v].z, s0);
          s0 = __builtin_fmaf(w4[k][v].w, xv[v].w, s0);
        }
        acc[k] = s0;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool ok = i0 + k < M;
        const float r = ok ? wave_sum64(acc[k]) - yb[ok ? i0 + k : 0] : 0.0f;
        if (lane == 0) facc += coef * r * r;
        if (want_g) {
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            ga[v].x = __builtin_fmaf(r, w4[k][v].x, ga[v].x);
            ga[v].y = __builtin_fmaf(r, w4[k][v].y, ga[v].y);
            ga[v].z = __builtin_fmaf(r, w4[k][v].z, ga[v].z);
            ga[v].w = __builtin_fmaf(r, w4[k][v].w, ga[v].w);
          }
        }
      }
    };
    // kCuRing row groups in the ring, all but one in flight while that one is reduced; the first trip's
    // were requested at the end of the previous optimizer phase, ahead of barrier B2
    for (int i0 = i_first; i0 < M; i0 += kCuRing * kRowStep) {
      use4(i0, wa);
      load4z(i0 + kCuRing * kRowStep, wa);
      if (i0 + kRowStep < M) use4(i0 + kRowStep, wb);
      load4z(i0 + (kCuRing + 1) * kRowStep, wb);
      if (kCuRing >= 3) {
        if (i0 + 2 * kRowStep < M) use4(i0 + 2 * kRowStep, wc);
        load4z(i0 + (kCuRing + 2) * kRowStep, wc);
      }
      if (kCuRing >= 4) {
        if (i0 + 3 * kRowStep < M) use4(i0 + 3 * kRowStep, wd);
        load4z(i0 + (kCuRing + 3) * kRowStep, wd);
      }
    }
    if (want_g) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int j = 4 * (64 * v + lane);
        if (j < D) *reinterpret_cast<float4*>(part + wv * D + j) = ga[v];
      }
    }
    const float fw = wave_sum64(facc);
    if (lane == 0) red[wv] = fw;
    lds_barrier();                                            // B1: partial gradients and partial f complete
    if (tid == 0) a.fx_part[(size_t)t * pp.B_local + b] = ((red[0] + red[1]) + red[2]) + red[3];
    if (!want_g) break;
    if (HIST && t == a.T) {                                   // the gradient at x_T, then done
      for (int tile = wv; tile < tpp; tile += kCuWaves) {
        float xsv;
        const float gv = grad_of(tile, xsv);
        const int j = tile * kTile + c;
        if (j < D && q == 0) a.hist_gfinal[(size_t)b * D + j] = gv;
      }
      break;
    }

    // ---- optimizer network on this wave's tiles ------------------------------------------------
    if (PRE == L2O_PRE_FC_ELU) { om1 = 1.0f - p1h; om2 = 1.0f - p2h; }
#pragma unroll 1
    for (int k = 0; k < nlds; ++k) {
      const int tile = wv + kCuWaves * k;
      if (tile >= tpp) break;
      float4* slot = stL + k * kCuSlotF4 + lane;
      TileState s;
      {
        const float4 v0 = slot[0], v1 = slot[64], v2 = slot[128], v3 = slot[192], v4 = slot[256];
        const float e[20] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y,
                             v2.z, v2.w, v3.x, v3.y, v3.z, v3.w, v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int t5 = 0; t5 < kNT; ++t5) { s.h1[t5] = e[t5]; s.c1[t5] = e[5 + t5]; s.h2[t5] = e[10 + t5]; s.c2[t5] = e[15 + t5]; }
      }
      do_tile(tile, s);
      {
        float e[20];
#pragma unroll
        for (int t5 = 0; t5 < kNT; ++t5) { e[t5] = s.h1[t5]; e[5 + t5] = s.c1[t5]; e[10 + t5] = s.h2[t5]; e[15 + t5] = s.c2[t5]; }
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) {
          float4 v4; v4.x = e[4 * jj]; v4.y = e[4 * jj + 1]; v4.z = e[4 * jj + 2]; v4.w = e[4 * jj + 3];
          slot[jj * 64] = v4;
        }
      }
    }
    if (has7) do_tile(tile7, s7);
    if (PRE == L2O_PRE_FC_ELU) {                              // beta^k as a float-float running product
      float hi = p1h * a.np.beta1, er = __builtin_fmaf(p1h, a.np.beta1, -hi);
      float lo = __builtin_fmaf(p1l, a.np.beta1, er), sum = hi + lo;
      p1l = lo - (sum - hi); p1h = sum;
      hi = p2h * a.np.beta2; er = __builtin_fmaf(p2h, a.np.beta2, -hi);
      lo = __builtin_fmaf(p2l, a.np.beta2, er); sum = hi + lo;
      p2l = lo - (sum - hi); p2h = sum;
    }
    load_head();                                              // (the network's temporaries are dead: registers for the ring)
    lds_barrier();                                            // B2: x s of the next step complete, `part` free
  }

  // ---- write back: x, moments, LSTM state ------------------------------------------------------
  for (int j = tid; j < D; j += kCuThreads) {
    const size_t idx = (size_t)b * D + j;
    a.x[idx] = xL[j];
    if (PRE == L2O_PRE_FC_ELU) { a.m[idx] = mL[j]; a.v[idx] = vL[j]; }
  }
  for (int k = 0; k < nlds; ++k) {
    const int tile = wv + kCuWaves * k;
    if (tile < tpp) {
      float4* dst = reinterpret_cast<float4*>(st_b + (size_t)tile * kStateFloatsPerTile);
#pragma unroll
      for (int jj = 0; jj < 5; ++jj) dst[jj * 64 + lane] = stL[k * kCuSlotF4 + jj * 64 + lane];
    }
  }
  if (has7) store_tile_state(s7, st_b + (size_t)tile7 * kStateFloatsPerTile, lane);
}

}  // namespace l2o
