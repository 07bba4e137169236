// l2o_unroll_cu8.h -- the streaming fused unroll (l2o_unroll_cu.h: D <= 512, any M: BASELINE config 3) with EIGHT waves per
// workgroup -- two per SIMD.  Included by l2o_kernels.hip after l2o_unroll_cu.h; written for gfx950 only.
//
// Why (round 4, after k_unroll_lds): k_unroll_cu's step is an 11 us matrix stream followed by a 17 us optimizer phase in
// which every SIMD runs ONE wave through eight tile steps, one after the other -- and a single wave is issue-bound
// (one VALU instruction per ~5.3 cycles where the pipe takes one per ~2.7).  Unlike the waves of
// k_unroll_lds, the tile steps of the optimizer phase are independent of each other (no barrier until all tiles are
// done), so two free-running waves per SIMD fill each other's stalls.  Two waves per SIMD mean <= 256 registers per
// wave, so -- as in k_unroll_lds -- the bf16x3 fragments live in LDS (the PACKED form for every net: 60 KB DM, 80 KB
// RNNProp, i.e. 80 MFMAs per RNNProp tile step instead of the register form's 120), and the LSTM state that used to fill
// the LDS moves into registers: a wave owns at most four tiles, KR of them register-resident (20 registers each), the
// others in an LDS slot.
//   LDS  : fragments | state slots of the tiles beyond KR per wave | part[8][D] | x, x*s, s, m, v | red[8]
//   per step:  GEMV phase (8 waves x M/8 rows, a ring of 2 row groups per wave) -> partial g, partial f -> barrier B1 ->
//              optimizer phase (wave w: tiles w, w + 8, ...) -> x, x*s in LDS -> barrier B2
#pragma once

namespace l2o {

constexpr int kCu8Waves = 8;
constexpr int kCu8Threads = 64 * kCu8Waves;

struct UnrollCu8Layout { int tpp, nslots, nlds, DP; size_t frag_floats, lds; };
// KR: register-resident tiles per wave
static inline UnrollCu8Layout unroll_cu8_layout(int D, int pre, int KR) {
  UnrollCu8Layout L;
  L.tpp = (D + kTile - 1) / kTile;
  L.nslots = (L.tpp + kCu8Waves - 1) / kCu8Waves;
  L.nlds = L.nslots > KR ? L.nslots - KR : 0;
  L.DP = L.tpp * kTile;
  L.frag_floats = (size_t)bx::packed_words(pre);
  // (+ the DM nets' input-weight rows, read from LDS like the biases: round 5)
  const size_t win_floats = pre == L2O_PRE_FC_ELU ? 0 : (size_t)(pre == L2O_PRE_LOGSIGN ? 2 : 1) * kNT * 256;
  L.lds = sizeof(float) * (L.frag_floats + (size_t)kCu8Waves * L.nlds * kCuSlotF4 * 4 + (size_t)kCu8Waves * D +
                           5 * (size_t)L.DP + 8 + win_floats);
  return L;
}

#ifndef L2O_CU8_RING
#define L2O_CU8_RING 2    // row groups (4 rows each) per wave in flight: 8 waves x 2 x 8 KB = 128 KB per CU at D = 512
                          // (config 3, RNNProp, four register tiles: ring 2 -> 3 spilled registers, 4.60 ms; 3 -> 60, 5.39; 4 -> 126, 6.02)
#endif
template <int PRE, int NV, int KR, bool HIST>
__global__ __launch_bounds__(kCu8Threads) void k_unroll_cu8(UnrollArgs a) {
  constexpr int kRing = L2O_CU8_RING;
  extern __shared__ __attribute__((aligned(16))) float cu8_smem[];
  const ProbParams& pp = a.pp;
  const int D = pp.D, M = pp.M;
  const int tpp = (D + kTile - 1) / kTile;
  const int nslots = (tpp + kCu8Waves - 1) / kCu8Waves;
  const int nlds = nslots > KR ? nslots - KR : 0;
  const int DP = tpp * kTile;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const int b = blockIdx.x;
  using Core = LstmCoreLds<PRE, bx::NetWBLF<PRE>>;      // (fenced fragment reads: see NetWBLF)
  float* frs = cu8_smem;                                                     // [Core::kFragWords] packed fragments
  float4* stL = reinterpret_cast<float4*>(frs + Core::kFragWords) + (size_t)wv * nlds * kCuSlotF4;   // this wave's slots
  float* part = frs + Core::kFragWords + (size_t)kCu8Waves * nlds * kCuSlotF4 * 4;                   // [8][D]
  float* xL = part + kCu8Waves * D;      // [DP] x
  float* xsL = xL + DP;                  // [DP] x * s (what the optimizee sees)
  float* scL = xsL + DP;                 // [DP] s
  float* mL = scL + DP;                  // [DP] RNNProp moments
  float* vL = mL + DP;
  float* red = vL + DP;                  // [8]
  float* winL = red + 8;                 // [Core WT::kWinFloats] the DM nets' input-weight rows (16-byte aligned: DP % 16 == 0)

  const int kind = pp.kind;
  const bool kCos = kind == L2O_PROB_RASTRIGIN || kind == L2O_PROB_SQUARE_COS;
  const bool kSq = kind == L2O_PROB_QUADRATIC || kind == L2O_PROB_SQUARE_COS;
  const float coef = kSq ? 1.0f : 0.5f;
  const float cg = kSq ? 2.0f : 1.0f;
  const float* Wb = pp.W + (pp.w_shared ? (size_t)0 : (size_t)b * M * D);
  const l2o_cfp yb = (l2o_cfp)(pp.y + (size_t)b * M);
  const float* Cb = kCos ? pp.C + (size_t)b * D : nullptr;

  // ---- the matrix stream: groups of four rows, wave w owns groups w, w + 8, ... ------------
  constexpr int kRowStep = 4 * kCu8Waves;
  int jcol[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) jcol[v] = 4 * (64 * v + lane) < D ? 4 * (64 * v + lane) : 0;
  auto load4 = [&](int i0, float4 (&w4)[4][NV]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = i0 + k < M ? i0 + k : M - 1;             // clamped rows contribute r = 0 below
      // (opaque scalar row offset: see k_unroll_cu; the row index is wave-uniform by construction)
      const unsigned long long ro0 = (unsigned long long)i * (unsigned)D;
      unsigned rlo = __builtin_amdgcn_readfirstlane((unsigned)ro0), rhi = __builtin_amdgcn_readfirstlane((unsigned)(ro0 >> 32));
      asm volatile("" : "+s"(rlo), "+s"(rhi));
      const unsigned long long ro = ((unsigned long long)rhi << 32) | rlo;
      const float* rowp = Wb + ro;
#pragma unroll
      for (int v = 0; v < NV; ++v) w4[k][v] = *reinterpret_cast<const float4*>(rowp + jcol[v]);
    }
  };
  // The ring lives INSIDE the GEMV phase (declared there, every conditional load paired with a zeroing else-branch): as
  // loop-carried variables with conditional loads -- k_unroll_cu's form, where 512 registers make it harmless -- the 64
  // ring registers stay live through the whole optimizer phase (a phi with the old values), which at 256 registers is
  // the difference between 0 and 100 spilled registers.
  auto zero4 = [&](float4 (&w4)[4][NV]) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int v = 0; v < NV; ++v) w4[k][v] = float4{0.f, 0.f, 0.f, 0.f};
  };
  auto load4z = [&](int i0, float4 (&w4)[4][NV]) {
    if (i0 < M) load4(i0, w4); else zero4(w4);
  };
  const int i_first = 4 * wv;

  // ---- problem vectors into LDS, LSTM state into registers / slots ---------------------------
  for (int j = tid; j < DP; j += kCu8Threads) {
    const bool live = j < D;
    const size_t idx = (size_t)b * D + (live ? j : D - 1);
    const float xv = live ? (a.x_in ? a.x_in : a.x)[idx] : 0.0f;
    const float sc = (live && pp.x_scale) ? pp.x_scale[idx] : 1.0f;
    xL[j] = xv; scL[j] = sc; xsL[j] = xv * sc;
    mL[j] = (PRE == L2O_PRE_FC_ELU && live && !a.zero_state) ? a.m[idx] : 0.0f;
    vL[j] = (PRE == L2O_PRE_FC_ELU && live && !a.zero_state) ? a.v[idx] : 0.0f;
  }
  float* st_b = a.st + (size_t)b * tpp * kStateFloatsPerTile;
  TileState sr[KR];                                           // tiles wv + 8 k, k < KR
#pragma unroll
  for (int k = 0; k < KR; ++k) {
#pragma unroll
    for (int t5 = 0; t5 < kNT; ++t5) { sr[k].h1[t5] = 0.f; sr[k].c1[t5] = 0.f; sr[k].h2[t5] = 0.f; sr[k].c2[t5] = 0.f; }
    const int tile = wv + kCu8Waves * k;
    if (tile < tpp && !a.zero_state) load_tile_state(sr[k], st_b + (size_t)tile * kStateFloatsPerTile, lane);
  }
  for (int k = 0; k < nlds; ++k) {
    const int tile = wv + kCu8Waves * (KR + k);
    if (tile < tpp) {
      const float4* src = reinterpret_cast<const float4*>(st_b + (size_t)tile * kStateFloatsPerTile);
#pragma unroll
      for (int jj = 0; jj < 5; ++jj)
        stL[k * kCuSlotF4 + jj * 64 + lane] = a.zero_state ? float4{0.f, 0.f, 0.f, 0.f} : src[jj * 64 + lane];
    }
  }

  Core core;
  core.load(a.np.wpack, lane);
  core.stage_frags(frs, a.np.wpack, tid, kCu8Threads, lane);
  __shared__ __attribute__((aligned(16))) float bias_s[Core::kBiasFloats];
  core.stage_bias(bias_s, a.np.wpack, tid, kCu8Threads, q);
  bx::stage_win(core.w, winL, a.np.wpack, tid, kCu8Threads, lane);
  float p1h = a.p1_hi, p1l = a.p1_lo, p2h = a.p2_hi, p2l = a.p2_lo;
  float om1 = 1.0f, om2 = 1.0f;
  __syncthreads();

  int tcur = 0;
  const size_t hist_n = (size_t)pp.B_local * D;
  // the lane's coordinate inside a tile, re-made opaque every step: with up to five inlined copies of the tile body LICM
  // otherwise hoists a dozen per-lane LDS / global addresses PER COPY out of the step loop (~50 registers held for the
  // whole unroll in a kernel that has 256)
  int cc = c;
  auto grad_of = [&](int tile, float& xsv_out) __attribute__((always_inline)) {
    const int j = tile * kTile + cc;
    const bool live = j < D;
    const int jc = live ? j : D - 1;
    float s01 = part[jc] + part[D + jc], s23 = part[2 * D + jc] + part[3 * D + jc];
    float s45 = part[4 * D + jc] + part[5 * D + jc], s67 = part[6 * D + jc] + part[7 * D + jc];
    const float sum = (s01 + s23) + (s45 + s67);
    const float xsv = xsL[j], sc = scL[j];
    float gj = cg * sum;
    if (kind == L2O_PROB_LASSO) gj += pp.l1 * (xsv > 0.f ? 1.f : (xsv < 0.f ? -1.f : 0.f));
    if (kCos) gj += pp.twopi * pp.alpha * Cb[jc] * l2o::sin_f(pp.twopi * xsv);
    xsv_out = xsv;
    return live ? gj * pp.inv_bg * sc : 0.0f;
  };
  // (always_inline: with up to five call sites the inliner otherwise turns the tile body into a real function -- a stack
  //  frame in scratch and the state passed through memory)
  auto do_tile = [&](int tile, TileState& s) __attribute__((always_inline)) {
    const int j = tile * kTile + cc;
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
    float d = bx::tile_step_w<PRE, true, bx::NetWBLF<PRE>>(core.w, s, in0, in1, q);
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
    const bool want_g = t < a.T || HIST;
    tcur = t;
    cc = c;
    asm volatile("" : "+v"(cc));
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
    float4 ring[kRing][4][NV];
    static_for<0, kRing>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
#ifndef L2O_CU8_ABL_NOGEMV
      load4z(i_first + r * kRowStep, ring[r]);
#else
      zero4(ring[r]);
#endif
    });
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
#ifdef L2O_CU8_ABL_NOGEMV
    for (int i0 = i_first; i0 < 0; i0 += kRing * kRowStep) {
#else
    for (int i0 = i_first; i0 < M; i0 += kRing * kRowStep) {
#endif
      static_for<0, kRing>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        if (i0 + r * kRowStep < M) use4(i0 + r * kRowStep, ring[r]);
        load4z(i0 + (kRing + r) * kRowStep, ring[r]);
      });
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
    if (tid == 0)
      a.fx_part[(size_t)t * pp.B_local + b] = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
    if (!want_g) break;
    if (HIST && t == a.T) {                                   // the gradient at x_T, then done
      for (int tile = wv; tile < tpp; tile += kCu8Waves) {
        float xsv;
        const float gv = grad_of(tile, xsv);
        const int j = tile * kTile + cc;
        if (j < D && q == 0) a.hist_gfinal[(size_t)b * D + j] = gv;
      }
      break;
    }

    // ---- optimizer network on this wave's tiles ------------------------------------------------
    if (PRE == L2O_PRE_FC_ELU) { om1 = 1.0f - p1h; om2 = 1.0f - p2h; }
#ifndef L2O_CU8_ABL_NOOPT     // (timing ablations: -DL2O_CU8_ABL_NOOPT no optimizer phase, -DL2O_CU8_ABL_NOGEMV no matrix stream)
    static_for<0, KR>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      const int tile = wv + kCu8Waves * k;
      if (tile < tpp) do_tile(tile, sr[k]);
    });
#endif
#pragma unroll 1
    for (int k = 0; k < nlds; ++k) {
      const int tile = wv + kCu8Waves * (KR + k);
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
    if (PRE == L2O_PRE_FC_ELU) {                              // beta^k as a float-float running product
      float hi = p1h * a.np.beta1, er = __builtin_fmaf(p1h, a.np.beta1, -hi);
      float lo = __builtin_fmaf(p1l, a.np.beta1, er), sum = hi + lo;
      p1l = lo - (sum - hi); p1h = sum;
      hi = p2h * a.np.beta2; er = __builtin_fmaf(p2h, a.np.beta2, -hi);
      lo = __builtin_fmaf(p2l, a.np.beta2, er); sum = hi + lo;
      p2l = lo - (sum - hi); p2h = sum;
    }
    lds_barrier();                                            // B2: x s of the next step complete, `part` free
  }

  // ---- write back: x, moments, LSTM state ------------------------------------------------------
  for (int j = tid; j < D; j += kCu8Threads) {
    const size_t idx = (size_t)b * D + j;
    a.x[idx] = xL[j];
    if (PRE == L2O_PRE_FC_ELU) { a.m[idx] = mL[j]; a.v[idx] = vL[j]; }
  }
#pragma unroll
  for (int k = 0; k < KR; ++k) {
    const int tile = wv + kCu8Waves * k;
    if (tile < tpp) store_tile_state(sr[k], st_b + (size_t)tile * kStateFloatsPerTile, lane);
  }
  for (int k = 0; k < nlds; ++k) {
    const int tile = wv + kCu8Waves * (KR + k);
    if (tile < tpp) {
      float4* dst = reinterpret_cast<float4*>(st_b + (size_t)tile * kStateFloatsPerTile);
#pragma unroll
      for (int jj = 0; jj < 5; ++jj) dst[jj * 64 + lane] = stL[k * kCuSlotF4 + jj * 64 + lane];
    }
  }
}

}  // namespace l2o
