// l2o_unroll_pair.h -- the fused persistent unroll with every problem split over TWO
// workgroups (two CUs).  Included by l2o_kernels.hip after k_unroll.
//
// Why: a batch of B <= #CU/2 problems (BASELINE config 2: B = 128 on 256 CUs) leaves half
// the chip idle with one problem per CU, and the per-step critical path of a problem is
// bound by its own SIMDs' fp32 issue cycles.  Splitting the coordinates (tiles) of a problem
// over two CUs halves that path; the price is ONE exchange of the scaled iterate x*s
// (64 floats each way for d = 128) per step between the two partner workgroups.
//
// Exchange protocol (placement independent, MI355X_MICROARCH.md "valid forms"): data-tagged
// 8-byte granules {float x, uint tag = step + 1} written with ONE agent-scope relaxed atomic
// 64-bit store (global_store_dwordx2 sc1: write-through) and polled with agent-scope relaxed
// 64-bit loads (sc1: L1 bypass).  A granule is self-validating, so no flag, no fence.  Two
// parity buffers: a workgroup can publish step t+2 only after it has consumed the partner's
// step t+1, which the partner publishes only after it consumed ours of step t+1 -- the
// buffer of parity (t+2)&1 = t&1 is therefore free.  The whole buffer is zeroed by a
// hipMemsetAsync ahead of every launch (tag 0 is never valid).  Every spin is bounded: on
// timeout the kernel raises ws->status and stops waiting (results are then garbage and the
// host reports L2O_ERR_HIP) -- a non-resident partner can never hang the GPU.
//
// Each half computes the FULL residual r = W xs - y redundantly (W is resident in both CUs'
// LDS; 2 x 16 rows per wave) and the gradient / LSTM only for its own tiles.
#pragma once

struct PairWs {               // header of the caller-owned workspace
  unsigned status;            // 0 ok, 1 = partner timeout
  unsigned pad[15];
  long long phases[16];       // phase clock dump of the -DL2O_PROFILE_PHASES build (else unused)
};

struct UnrollPairArgs {
  UnrollArgs u;
  PairWs* ws;
  unsigned long long* xbuf;   // [B][2 halves][2 parities][NWH*16] granules
  float* fx_half;             // [(T+1)][2*B]
};

__device__ __forceinline__ unsigned long long pack_granule(float v, unsigned tag) {
  return ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
}

template <int PRE, int KIND, int CH>
__global__ __launch_bounds__(256) void k_unroll_pair(UnrollPairArgs pa) {
  constexpr int SQ = 16 * CH;
  constexpr int S = SQ + 16;
  extern __shared__ float sm[];
  const UnrollArgs& a = pa.u;
  const ProbParams& pp = a.pp;
  const int D = pp.D, M = pp.M;
  constexpr int NWH = CH / 2;            // waves (tiles) per half; tiles beyond the real count idle
  float* Ws = sm;                        // [SQ][S]       W  (all rows, all columns)
  float* WTs = Ws + SQ * S;              // [NWH*16][S]   W^T rows of this half's coordinates
  float* xs = WTs + NWH * 16 * S;        // [SQ]
  float* rs = xs + SQ;                   // [SQ]
  float* ys = rs + SQ;                   // [SQ]
  float* fpart = ys + SQ;                // [8]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  // partner workgroups are blockIdx b and b + 8 inside a group of 16 (same XCD under the
  // observed round-robin placement -- a speed choice only)
  const int bid = blockIdx.x;
  const int half = (bid >> 3) & 1;
  const int b = ((bid >> 4) << 3) | (bid & 7);          // problem index
  if (b >= pp.B_local) return;                          // padding blocks of the last group of 16 (both halves)
  const int tile_in_prob = half * NWH + wv;             // this wave's coordinate tile
  const int gq = lane & 3, gr = lane >> 2;              // GEMV role

  // ---- stage the problem into LDS -----------------------------------------
  const float* Wb = pp.W + (size_t)b * M * D;
  const int lds_floats = SQ * S + NWH * 16 * S + 3 * SQ;
  for (int i = tid; i < lds_floats; i += blockDim.x) sm[i] = 0.0f;
  __syncthreads();
  const int jlo = half * NWH * 16, jhi = jlo + NWH * 16;
  for (int e = tid; e < M * D; e += blockDim.x) {
    const int i = e / D, j = e - i * D;
    const float v = Wb[e];
    Ws[i * S + j] = v;
    if (j >= jlo && j < jhi) WTs[(j - jlo) * S + i] = v;
  }
  for (int i = tid; i < M; i += blockDim.x) ys[i] = pp.y[(size_t)b * M + i];

  // ---- per-lane persistent registers -------------------------------------
#ifdef L2O_PAIR_FP32
  using Core = LstmCore<PRE, false>;
#else
  using Core = LstmCore<PRE, true>;      // <= 4 waves per workgroup: bf16x3 gate GEMM, weights in VGPR + AGPR
#endif
  Core core;
  core.load(a.np.wpack, lane);
  const int j = tile_in_prob * kTile + c;
  const bool live = j < D;
  const size_t idx = (size_t)b * D + j;
  const int tpp = (D + kTile - 1) / kTile;
  const bool tile_real = tile_in_prob < tpp;            // the padded tile of an odd tile count is idle
  TileState s;
  float* st_tile = a.st + ((size_t)b * tpp + (tile_real ? tile_in_prob : 0)) * kStateFloatsPerTile;
  if (tile_real) load_tile_state(s, st_tile, lane);
  else {
#pragma unroll
    for (int t = 0; t < kNT; ++t) s.h1[t] = s.c1[t] = s.h2[t] = s.c2[t] = 0.0f;
  }
  float xv = live ? a.x[idx] : 0.0f;
  const float sc = (live && pp.x_scale) ? pp.x_scale[idx] : 1.0f;
  float cj = 0.0f;
  constexpr bool kCos = KIND == L2O_PROB_RASTRIGIN || KIND == L2O_PROB_SQUARE_COS;
  if (kCos) cj = live ? pp.C[idx] : 0.0f;
  float mv = 0.0f, vv = 0.0f;
  if (PRE == L2O_PRE_FC_ELU) { mv = live ? a.m[idx] : 0.0f; vv = live ? a.v[idx] : 0.0f; }
  float p1h = a.p1_hi, p1l = a.p1_lo, p2h = a.p2_hi, p2l = a.p2_lo;
  constexpr bool kSq = KIND == L2O_PROB_QUADRATIC || KIND == L2O_PROB_SQUARE_COS;
  const float coef = kSq ? 1.0f : 0.5f;
  const float cg = (KIND == L2O_PROB_QUADRATIC ? 2.0f : 1.0f) * pp.inv_bg;   // x2 folded in (exact)
  const float kTwoPi = pp.twopi;
  const float* wtrow = WTs + (wv * kTile + gr) * S + 4 * gq;
  const float* xsq = xs + 4 * gq;
  const float* rsq = rs + 4 * gq;
  const int perm_src = (4 * c) << 2;
  const int npg = NWH * 16;                                      // granules per (half, parity)
  unsigned long long* mine = pa.xbuf + ((size_t)b * 2 + half) * 2 * npg;
  const unsigned long long* theirs = pa.xbuf + ((size_t)b * 2 + (half ^ 1)) * 2 * npg;
  const int pj = (half ^ 1) * npg + wv * kTile + c;              // partner coordinate this lane fetches
  bool dead = false;                                             // partner timed out

  f32x4 acc1[kNT], acc2[kNT];
  core.init(s, q);
  core.template issue_l1_prev<0, Core::kTotal>(s, acc1);
  PhaseClock pc;
  pc.start();

  for (int t = 0;; ++t) {
    const float xsv = xv * sc;
    const unsigned tag = (unsigned)t + 1u;
    const int par = t & 1;
    // ---- publish this half's x*s (one granule per coordinate), then fetch the partner's
    if (q == 0) {
      xs[j] = live ? xsv : 0.0f;
      __hip_atomic_store(mine + par * npg + wv * kTile + c, pack_granule(live ? xsv : 0.0f, tag),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    pc.mark(0);                                                   // publish
    core.template issue_l2_prev<0, Core::kHalf>(s, acc2);                   // matrix work that covers the latency
    if (q == 0) {
      const unsigned long long* src = theirs + par * npg + wv * kTile + c;
      unsigned long long g = 0;
      int spins = 0;
#ifdef L2O_ABLATE_EXCHANGE
      dead = true;
#endif
      if (!dead) {
#pragma nounroll
        for (;;) {
          g = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((unsigned)(g >> 32) == tag) break;
          if (++spins > (1 << 20)) { dead = true; atomicExch(&pa.ws->status, 1u); break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      xs[pj] = __uint_as_float((unsigned)g);
    }
    pc.mark(1);                                             // 12 MFMAs + partner poll
    __syncthreads();                                        // B1: xs (both halves) complete
    pc.mark(2);
    // ---- r = W xs - y : every half computes all rows, 2 x 16 per wave  ||  13 more MFMAs
    float contrib = 0.0f;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = (2 * wv + p) * kTile + gr;            // NWH waves x 32 rows = all SQ rows
      float4 racc = {0.f, 0.f, 0.f, 0.f};
      const float* wrow = Ws + row * S + 4 * gq;
      static_for<0, CH>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        const float4 wv4 = *reinterpret_cast<const float4*>(wrow + 16 * m);
        const float4 xv4 = *reinterpret_cast<const float4*>(xsq + 16 * m);
        dot4(wv4, xv4, racc);
      });
      if (p == 0) core.template issue_l2_prev<Core::kHalf, Core::kTotal>(s, acc2);
      const float r = quad_sum(hsum4(racc)) - ys[row];
      if (gq == 0) {
        rs[row] = r;                      // rows >= M: W row and y are zero -> r == 0
        const bool mine_row = half == 0 ? (row < npg) : (row >= npg);   // count every row once
        if (mine_row) contrib = __builtin_fmaf(coef * r, r, contrib);
      }
    }
    if (live && q == 0) {
      if (KIND == L2O_PROB_LASSO) contrib += pp.l1 * __builtin_fabsf(xsv);
      if (kCos) contrib += pp.alpha - pp.alpha * cj * cosf(kTwoPi * xsv);
    }
    pc.mark(3);                                             // r pass + 13 MFMAs
    contrib = wave_sum64(contrib);
    if (lane == 0) fpart[wv] = contrib;
    __syncthreads();                                        // B2: rs, fpart complete
    pc.mark(4);
    if (tid == 0) {
      float f = fpart[0];
      for (int k = 1; k < NWH; ++k) f += fpart[k];
      pa.fx_half[(size_t)t * 2 * pp.B_local + 2 * b + half] = f;
    }
    if (t == a.T) break;

    // ---- g = W^T r for this wave's 16 coordinates ------------------------------
    float4 gacc4 = {0.f, 0.f, 0.f, 0.f};
    static_for<0, CH>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      const float4 wt4 = *reinterpret_cast<const float4*>(wtrow + 16 * m);
      const float4 rv4 = *reinterpret_cast<const float4*>(rsq + 16 * m);
      dot4(wt4, rv4, gacc4);
    });
    const float gacc = quad_sum(hsum4(gacc4));
    float gv = __int_as_float(__builtin_amdgcn_ds_bpermute(perm_src, __float_as_int(gacc)));
    if (KIND == L2O_PROB_SQUARE_COS) gv *= 2.0f;            // only the ||wx-y||^2 part carries the 2
    if (KIND == L2O_PROB_LASSO) gv += pp.l1 * (xsv > 0.f ? 1.f : (xsv < 0.f ? -1.f : 0.f));
    if (kCos) gv += kTwoPi * pp.alpha * cj * sinf(kTwoPi * xsv);
    gv = live ? gv * cg * sc : 0.0f;

    float in0, in1;
    if (PRE == L2O_PRE_FC_ELU) {
      rnnprop_inputs(gv, mv, vv, a.np.beta1, a.np.beta2, a.np.omb1, a.np.omb2, 1.0f - p1h, 1.0f - p2h, in0, in1);
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
    float d = core.template finish<true>(s, acc1, acc2, in0, in1, q, pc);   // marks 5 (g pass .. input MFMAs), 6, 7, 8
    if (a.np.tanh_output) d = tanhf_(d);
    xv = __builtin_fmaf(d, a.np.scale, xv);
    pc.mark(9);
  }
#ifdef L2O_PROFILE_PHASES
  if (blockIdx.x == 0 && tid == 0) pc.dump(pa.ws->phases);
#endif

  if (live && q == 0) {
    a.x[idx] = xv;
    if (PRE == L2O_PRE_FC_ELU) { a.m[idx] = mv; a.v[idx] = vv; }
  }
  if (tile_real) store_tile_state(s, st_tile, lane);
}

// fx_part[t][b] = fx_half[t][2b] + fx_half[t][2b+1]
__global__ void k_combine_halves(const float* __restrict__ fx_half, float* __restrict__ fx_part, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) fx_part[i] = fx_half[2 * i] + fx_half[2 * i + 1];
}
