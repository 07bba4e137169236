// l2o_unroll_lds.h -- the fused persistent unroll for LARGE shards: one problem per workgroup / CU, EIGHT waves
// (two per SIMD), the bf16x3 gate GEMM with its weight fragments in LDS.  Included by l2o_kernels.hip after k_unroll.
//
// Why (round 4): a shard of more than #CU / 2 problems runs the two-CU kernel as consecutive chunk launches -- every
// chunk pays the cross-CU exchange and leaves each SIMD with one latency-bound wave (BASELINE config 4 on one GPU: 8 chunks,
// 2 x 4 470 = 8 940 CU-cycles per problem-step).  With one problem per CU there is no exchange at all, but 5..8 tiles
// need two waves per SIMD, i.e. <= 256 registers per wave: the register-resident bf16x3 fragments (240 AGPRs) do not
// fit, and k_unroll falls back to the fp32 MFMA there (11 200 CU-cycles per problem-step, slower than the chunks).
// Here the 60 KB of packed fragments live in LDS (one ds_read_b128 per MFMA, bx::NetWBL) and the problem's matrix in
// REGISTERS (the wave's 16 rows and 16 columns of W: 64 registers, as in k_unroll_pair), so LDS carries the fragments,
// x, r and nothing else.
//   per step:  B1 | r = W xs - y for the wave's 16 rows (8 ds_read_b128 of xs, 32 FMAs; 10 L2B MFMAs interleaved)
//              -> rs, per-wave loss partial | B2 | g = W^T r for the wave's 16 coordinates (8 reads of rs, 32 FMAs; 10 more)
//              -> preprocess -> input FMAs -> layer-1 gates -> split h1 -> 20 MFMAs (L2A) + the next step's 20 (L1H)
//              under the layer-2 gate block -> Linear -> x += delta -> xs -> LDS
// 65 <= padded size <= 128 (5..8 tiles); DM nets (identity / LogAndSign preprocessing: three packed chunks, 60 KB) and
// RNNProp (fc + ELU: four chunks, 80 KB).
#pragma once

// (timing ablations, scripts/ab.sh ablate_lds: L2O_LDS_ABL_NOBAR drops the two barriers, _NOGEMV the xs / rs reads,
//  _NOFRAG (l2o_lstm_bx3.h) the fragment reads -- each gives wrong numerics and is never shipped)
#ifdef L2O_LDS_ABL_NOBAR
#define L2O_LDS_BARRIER() __builtin_amdgcn_sched_barrier(0)
#else
#define L2O_LDS_BARRIER() do { if (HIST) lds_barrier(); else __syncthreads(); } while (0)
#endif
#ifdef L2O_LDS_ABL_NOGEMV
#define L2O_LDS_VEC(p, m, w) (w)
#else
#define L2O_LDS_VEC(p, m, w) \
  (*reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((const __attribute__((address_space(3))) float*)(p) + 16 * (m)))
#endif
// experiment switches (scripts/build_variants.sh):
//   L2O_LDS_PRIO        the first wave of every SIMD (waves 0..3) runs at s_setprio 3, its partner (waves 4..7) at 0
//   L2O_LDS_MFMA_ORDER  chunk L1H (h1 of the previous step) rides in the r pass, chunk L2B in the g pass, and finish()
//                       issues chunk L2A only (NEXT = false): no next-step MFMAs queued in front of the partner wave's L2A
#ifndef L2O_LDS_MFMA_ORDER
#define L2O_LDS_MFMA_ORDER 1
#endif

template <int PRE, int KIND, bool HIST>
__global__ __launch_bounds__(512) void k_unroll_lds(UnrollArgs a) {
  const long long kernel_t0 = __builtin_readcyclecounter();
  constexpr int CH = 8, SQ = 16 * CH;
  using Core = LstmCoreLds<PRE>;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* frs = sm;                                   // [Core::kFragWords]  packed fragments (16-byte aligned)
  float* xs = frs + Core::kFragWords;                // [SQ]
  float* rs = xs + SQ;                               // [SQ]
  float* fpart = rs + SQ;                            // [8]
  const ProbParams& pp = a.pp;
  const int D = pp.D, M = pp.M;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nw = blockDim.x >> 6;                    // waves = tiles of the problem (5..8)
  const int c = lane & 15, q = lane >> 4;
  const int b = blockIdx.x;

  // ---- the matrix in registers: row 16 wv + c (r pass) and column 16 wv + c (g pass), 16-byte chunk q of every tile
  const float* Wb = pp.W + (pp.w_shared ? (size_t)0 : (size_t)b * M * D);
  f32x4 wr[CH], wt[CH];
  {
    const int row = wv * kTile + c, col = wv * kTile + c;
#pragma unroll
    for (int m = 0; m < CH; ++m) {
      float e[4], f[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int cc = 16 * m + 4 * q + k;           // column (r pass) / row (g pass) index of this element
        e[k] = (row < M && cc < D) ? Wb[(size_t)row * D + cc] : 0.0f;
        f[k] = (cc < M && col < D) ? Wb[(size_t)cc * D + col] : 0.0f;
      }
      wr[m] = f32x4{e[0], e[1], e[2], e[3]};
      wt[m] = f32x4{f[0], f[1], f[2], f[3]};
    }
  }
  const int myrow = wv * kTile + c;
  const float myy = myrow < M ? pp.y[(size_t)b * M + myrow] : 0.0f;

  Core core;
  core.load(a.np.wpack, lane);
  core.stage_frags(frs, a.np.wpack, tid, blockDim.x, lane);
  __shared__ __attribute__((aligned(16))) float bias_s[Core::kBiasFloats];
  core.stage_bias(bias_s, a.np.wpack, tid, blockDim.x, q);
  const int j = wv * kTile + c;
  const bool live = j < D;
  const size_t idx = (size_t)b * D + j;
  const int tile = b * nw + wv;
  TileState s;
  float* st_tile = a.st + (size_t)tile * kStateFloatsPerTile;
  if (a.zero_state) {
#pragma unroll
    for (int t5 = 0; t5 < kNT; ++t5) s.h1[t5] = s.c1[t5] = s.h2[t5] = s.c2[t5] = 0.0f;
  } else {
    load_tile_state(s, st_tile, lane);
  }
  float xv = live ? (a.x_in ? a.x_in : a.x)[idx] : 0.0f;
  const float sc = (live && pp.x_scale) ? pp.x_scale[idx] : 1.0f;
  float cj = 0.0f;
  constexpr bool kCos = KIND == L2O_PROB_RASTRIGIN || KIND == L2O_PROB_SQUARE_COS;
  if (kCos) cj = live ? pp.C[idx] : 0.0f;
  // RNNProp (fc + ELU preprocessing, four packed chunks = 80 KB of fragments): Adam moments in registers, the bias
  // corrections beta^(step0 + t) as double-float products, exactly as k_unroll / k_unroll_pair carry them
  float mv = 0.0f, vv = 0.0f;
  if (PRE == L2O_PRE_FC_ELU && !a.zero_state) { mv = live ? a.m[idx] : 0.0f; vv = live ? a.v[idx] : 0.0f; }
  float p1h = a.p1_hi, p1l = a.p1_lo, p2h = a.p2_hi, p2l = a.p2_lo;
  constexpr bool kSq = KIND == L2O_PROB_QUADRATIC || KIND == L2O_PROB_SQUARE_COS;
  const float coef = kSq ? 1.0f : 0.5f;
  const float cg = (KIND == L2O_PROB_QUADRATIC ? 2.0f : 1.0f) * pp.inv_bg;
  const float kTwoPi = pp.twopi;
  const float* xsq = xs + 4 * q;
  const float* rsq = rs + 4 * q;

#ifdef L2O_LDS_PRIO
  if (wv < 4) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
#endif
  f32x4 acc1[kNT], acc2[kNT];
  core.init(s, q);
  if (tid < SQ) { xs[tid] = 0.0f; rs[tid] = 0.0f; }
  __syncthreads();                                   // fragments, bias table, zeroed xs staged
  core.preload(acc1, acc2);
  if (!L2O_LDS_MFMA_ORDER) core.template issue_l1_prev<0, Core::kTotal>(s, acc1);
  if (q == 0) xs[j] = live ? xv * sc : 0.0f;

  const size_t hist_n = (size_t)pp.B_local * D;
  PhaseClock pc;
  const long long loop_t0 = __builtin_readcyclecounter();
  for (int t = 0;; ++t) {
    const float xsv = xv * sc;
    if (HIST && t < a.T)
      store_tile_state(s, a.hist_st + ((size_t)t * pp.B_local * nw + tile) * kStateFloatsPerTile, lane);
    L2O_LDS_BARRIER();                                 // B1: xs complete
    // ---- r = W xs - y for the wave's 16 rows  ||  the first 10 layer-2 MFMAs of the previous h2
    float4 racc = {0.f, 0.f, 0.f, 0.f};                    // (scalar FMAs: two waves per SIMD -- see dot4pk in l2o_unroll_pair.h)
    static_for<0, CH>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      const f32x4 xv4 = L2O_LDS_VEC(xsq, m, wt[m]);
      if (L2O_LDS_MFMA_ORDER) core.template issue_l1_prev<(Core::kTotal * m) / CH, (Core::kTotal * (m + 1)) / CH>(s, acc1);
      else core.template issue_l2_prev<(Core::kHalf * m) / CH, (Core::kHalf * (m + 1)) / CH>(s, acc2);
      dot4q(wr[m], xv4, racc);
    });
    const float r = quad_q_sum(hsum4(racc)) - myy;  // (rows >= M: W row and y are zero -> r == 0)
    float contrib = 0.0f;
    if (q == 0) {
      rs[myrow] = r;
      contrib = coef * r * r;
      if (live) {
        if (KIND == L2O_PROB_LASSO) contrib += pp.l1 * __builtin_fabsf(xsv);
        if (kCos) contrib += pp.alpha - pp.alpha * cj * l2o::cos_f(kTwoPi * xsv);
      }
    }
    contrib = wave_sum64(contrib);
    if (lane == 0) fpart[wv] = contrib;
    L2O_LDS_BARRIER();                                 // B2: rs, fpart complete
    if (tid == 0) {
      float f = fpart[0];
      for (int k = 1; k < nw; ++k) f += fpart[k];
      a.fx_part[(size_t)t * pp.B_local + b] = f;
    }
    if (t == a.T && !HIST) break;

    // ---- g = W^T r for this wave's 16 coordinates  ||  the other 10 of those MFMAs
    float4 gacc4 = {0.f, 0.f, 0.f, 0.f};
    static_for<0, CH>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      const f32x4 rv4 = L2O_LDS_VEC(rsq, m, wr[m]);
      if (L2O_LDS_MFMA_ORDER) core.template issue_l2_prev<(Core::kTotal * m) / CH, (Core::kTotal * (m + 1)) / CH>(s, acc2);
      else core.template issue_l2_prev<Core::kHalf + ((Core::kTotal - Core::kHalf) * m) / CH,
                                       Core::kHalf + ((Core::kTotal - Core::kHalf) * (m + 1)) / CH>(s, acc2);
      dot4q(wt[m], rv4, gacc4);
    });
    float gv = quad_q_sum(hsum4(gacc4));
    if (KIND == L2O_PROB_SQUARE_COS) gv *= 2.0f;
    if (KIND == L2O_PROB_LASSO) gv += pp.l1 * (xsv > 0.f ? 1.f : (xsv < 0.f ? -1.f : 0.f));
    if (kCos) gv += kTwoPi * pp.alpha * cj * l2o::sin_f(kTwoPi * xsv);
    gv = live ? gv * cg * sc : 0.0f;
    if (HIST && live && q == 0) {
      if (t < a.T) a.hist_g[(size_t)t * hist_n + idx] = gv;
      else a.hist_gfinal[idx] = gv;
    }
    if (HIST && t == a.T) break;

    float in0, in1;
    if (PRE == L2O_PRE_FC_ELU) {
      rnnprop_inputs(gv, mv, vv, a.np.beta1, a.np.beta2, a.np.omb1, a.np.omb2, 1.0f - p1h, 1.0f - p2h, in0, in1);
      if (HIST && live && q == 0) { a.hist_m[(size_t)t * hist_n + idx] = mv; a.hist_v[(size_t)t * hist_n + idx] = vv; }
      if (!live) { in0 = 0.0f; in1 = 0.0f; }
      {
        float hi = p1h * a.np.beta1, er = __builtin_fmaf(p1h, a.np.beta1, -hi);
        float lo = __builtin_fmaf(p1l, a.np.beta1, er), sum = hi + lo;
        p1l = lo - (sum - hi); p1h = sum;
        hi = p2h * a.np.beta2; er = __builtin_fmaf(p2h, a.np.beta2, -hi);
        lo = __builtin_fmaf(p2l, a.np.beta2, er); sum = hi + lo;
        p2l = lo - (sum - hi); p2h = sum;
      }
    } else {
      preprocess_grad<PRE>(gv, a.np.k_inv_ln2, a.np.exp_k, in0, in1);
    }
    float d = core.template finish<!L2O_LDS_MFMA_ORDER>(s, acc1, acc2, in0, in1, q, pc);
    if (L2O_LDS_MFMA_ORDER) core.refresh(s);             // split h2 -> the L2B operand of the next step
    if (a.np.tanh_output) {
      asm volatile("");
      d = tanhf_(d);
    }
    xv = __builtin_fmaf(d, a.np.scale, xv);
    // the next step's scaled iterate -> LDS now (this step's readers of xs all passed barrier B2)
    if (q == 0) xs[j] = live ? xv * sc : 0.0f;
  }
  // (the step loop of problem 0 in shader-clock cycles -> the workspace header, see PairWs::ticks; NULL without a workspace)
  if (a.ticks && blockIdx.x == 0 && threadIdx.x == 0) *a.ticks = __builtin_readcyclecounter() - loop_t0;

  if (live && q == 0) {
    a.x[idx] = xv;
    if (PRE == L2O_PRE_FC_ELU) { a.m[idx] = mv; a.v[idx] = vv; }
  }
  store_tile_state(s, st_tile, lane);
  if (a.ticks && blockIdx.x == 0 && threadIdx.x == 0) a.ticks[1] = __builtin_readcyclecounter() - kernel_t0;   // (PairWs::ticks_total)
}
