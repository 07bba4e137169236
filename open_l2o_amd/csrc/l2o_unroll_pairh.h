// l2o_unroll_pairh.h -- the two-CU fused unroll (l2o_unroll_pair.h) with the gradient of the
// linear-residual term taken from the PREPARED normal matrix.  Included by l2o_kernels.hip.
//
// Every optimizee of the fused forms is  f(x) = coef |W xs - y|^2 + separable terms,  xs = x * x_scale
// (problems.quadratic / lasso / rastrigin / square_cos, DM/problems.py:73-213, 959-994).  The reference evaluates
// r = W xs - y and lets autodiff produce  W^T r.  Per step that is TWO dependent GEMV passes over W and, with a
// problem split over two CUs, an exchange of the partial residuals BETWEEN them: publish -> partner -> poll sat in
// the middle of the step's critical path (profiles/r02p_phase_clock_k_unroll_pair.txt: partial r 778 + window 875
// + g pass 678 of 4 751 ticks).  Here
//      g_lin = W^T (W xs - y) = H xs - q,      H = W^T W (D x D),   q = W^T y
// with H, q computed ONCE per problem by k_pair_prepare (float64 accumulation, rounded once: the rounding of H is
// below that of the reference's own fp32 GEMV).  The step's critical path then holds ONE GEMV, and what the halves
// exchange is the iterate itself, published the moment it exists:
//      x(t) -> publish own 64 | own-half of H xs under the wait | partner's 64 -> LDS | other half | g
// The loss keeps the reference's form -- r = W xs - y from W, f = coef |r|^2 -- so f is NOT formed from H (x^T H x
// - 2 q^T x + y^T y cancels catastrophically near a minimum); r is off the critical path: its FMAs sit next to
// the H ones, its reductions and the store ride behind chunk L2A's MFMAs (LstmCore::finish, shadow hook).
// The gradient differs from the reference's W^T r in ROUNDING only (both carry errors of order eps |W|^2 |x|);
// parity with the oracle is asserted at the same tolerance as before (tests/test_hip_kernels.py).
//
// Register budget per lane = the round-1 kernel's: H[j][32 columns] + W[row j][32 columns] = 64 registers (was:
// W twice, row- and column-major).  Exchange protocol, tags, salt, same-XCD plain stores, bounded spins, sticky
// status: exactly l2o_unroll_pair.h's; 64 granules each way per step instead of 128.
#pragma once

struct UnrollPairHArgs {
  UnrollPairArgs p;
  const float* H;     // [nW][SQ][SQ], zero beyond D (nW = 1 for a shared W)
  const float* qv;    // [B][SQ] = W^T y
  int b0, nb;         // this launch steps problems [b0, b0 + nb) of the shard: a batch of more than #CU / 2 problems
                      // runs as consecutive launches of <= #CU / 2 (exchange granules and loss partials are per launch)
};

// ---- prepare: H = W^T W and q = W^T y, float64 accumulation, rounded once ---------------------
// grid (nW, 2): block (matrix, upper / lower half of the rows of H).  W (M x D, columns zero-padded to SQ) is staged
// in LDS once; the 16 x 16 threads own (SQ / 32) x (SQ / 16) tiles of the half: per row of W a thread reads its
// SQ / 32 + SQ / 16 values and issues their outer product as float64 FMAs (the DFMA rate bounds the block:
// 4 096 per thread at SQ = 128 plus 12 v_cvt_f64_f32 per 32 of them: 30 us; all blocks run concurrently).  With a per-problem W the block also forms its
// half of q (a shared W: k_pair_prepare_q, one block per problem).
template <int CH>
__global__ __launch_bounds__(256) void k_pair_prepare_h(const float* __restrict__ W, const float* __restrict__ y, int M,
                                                        int D, float* __restrict__ H, float* __restrict__ qv) {
  constexpr int SQ = 16 * CH, HR = SQ / 2, RT = HR / 16, CT = SQ / 16;
  extern __shared__ float wl[];                            // [M][SQ] (as doubles, converted once: 77 us instead of 30 --
                                                           // the 96 B per thread and row of W make the loop LDS-bound)
  __shared__ float ys[SQ];
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15, half = blockIdx.y;
  if (qv && tid < SQ) ys[tid] = tid < M ? y[(size_t)blockIdx.x * M + tid] : 0.0f;
  const float* Wb = W + (size_t)blockIdx.x * M * D;
  float* Hb = H + (size_t)blockIdx.x * SQ * SQ;
  {
    // all of the thread's loads in flight before the first LDS write (a plain loop issues them one latency at a time)
    constexpr int NQ = (SQ * SQ / 4 + 255) / 256;          // float4 groups per thread for M == SQ
    float4 v[NQ];
    const bool vec = D == SQ && (((size_t)Wb) & 15) == 0;
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      const int e4 = tid + 256 * k, r = e4 / (SQ / 4), c = 4 * (e4 - r * (SQ / 4));
      v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < M) {
        if (vec) v[k] = *reinterpret_cast<const float4*>(Wb + (size_t)r * D + c);
        else {
          const float* p = Wb + (size_t)r * D + c;
          v[k].x = c < D ? p[0] : 0.0f; v[k].y = c + 1 < D ? p[1] : 0.0f;
          v[k].z = c + 2 < D ? p[2] : 0.0f; v[k].w = c + 3 < D ? p[3] : 0.0f;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      const int e4 = tid + 256 * k;
      if (e4 < M * (SQ / 4)) *reinterpret_cast<float4*>(wl + 4 * e4) = v[k];
    }
  }
  __syncthreads();
  double acc[RT][CT];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j) acc[i][j] = 0.0;
  const float* pa = wl + half * HR + ty * RT;
  const float* pb = wl + tx * CT;
#pragma unroll 4
  for (int r = 0; r < M; ++r) {
    double a[RT], bb[CT];
#pragma unroll
    for (int i = 0; i < RT; ++i) a[i] = (double)pa[r * SQ + i];
#pragma unroll
    for (int j = 0; j < CT; ++j) bb[j] = (double)pb[r * SQ + j];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < CT; ++j) acc[i][j] = __builtin_fma(a[i], bb[j], acc[i][j]);
  }
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j) Hb[(size_t)(half * HR + ty * RT + i) * SQ + tx * CT + j] = (float)acc[i][j];
  if (qv && tid < HR) {                                    // per-problem W: q_i for the rows of this half
    const int i = half * HR + tid;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int r = 0;
#pragma unroll 2
    for (; r + 3 < M; r += 4) {
      s0 = __builtin_fma((double)wl[r * SQ + i], (double)ys[r], s0);
      s1 = __builtin_fma((double)wl[(r + 1) * SQ + i], (double)ys[r + 1], s1);
      s2 = __builtin_fma((double)wl[(r + 2) * SQ + i], (double)ys[r + 2], s2);
      s3 = __builtin_fma((double)wl[(r + 3) * SQ + i], (double)ys[r + 3], s3);
    }
    for (; r < M; ++r) s0 = __builtin_fma((double)wl[r * SQ + i], (double)ys[r], s0);
    qv[(size_t)blockIdx.x * SQ + i] = (float)((s0 + s1) + (s2 + s3));
  }
}
// shared W: grid B, block SQ threads: q[b][i] = sum_r W[r][i] y[b][r]
__global__ void k_pair_prepare_q(const float* __restrict__ W, const float* __restrict__ y, int M, int D, int SQ,
                                 float* __restrict__ qv) {
  const int b = blockIdx.x, i = threadIdx.x;
  double s0 = 0.0, s1 = 0.0;
  if (i < D) {
    int r = 0;
    for (; r + 1 < M; r += 2) {
      s0 = __builtin_fma((double)W[(size_t)r * D + i], (double)y[(size_t)b * M + r], s0);
      s1 = __builtin_fma((double)W[(size_t)(r + 1) * D + i], (double)y[(size_t)b * M + r + 1], s1);
    }
    if (r < M) s0 = __builtin_fma((double)W[(size_t)r * D + i], (double)y[(size_t)b * M + r], s0);
  }
  qv[(size_t)b * SQ + i] = (float)(s0 + s1);
}

template <int PRE, int KIND, int CH, bool HIST>
__global__ __launch_bounds__(256) void k_unroll_pairh(UnrollPairHArgs ha) {
  constexpr int SQ = 16 * CH;            // padded rows (and columns) of the problem
  constexpr int NWH = CH / 2;            // waves (tiles) per half; tiles beyond the real count idle
  constexpr int NC = 16 * NWH;           // columns (coordinates) and residual rows owned by a half = SQ / 2
  __shared__ float xs[NC];               // this half's scaled iterate
  __shared__ float xo[NC];               // the partner's
  const UnrollPairArgs& pa = ha.p;
  const UnrollArgs& a = pa.u;
  const ProbParams& pp = a.pp;
  const int D = pp.D, M = pp.M;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  // partner workgroups are blockIdx b and b + 8 inside a group of 16 (same XCD under the
  // observed round-robin placement -- a speed choice only)
  const int bid = blockIdx.x;
  const unsigned salt = pa.use_salt ? ((pa.ws->seq + 1u) & 0x7fffu) << 16 : 0u;
  const int half = (bid >> 3) & 1;
  const int bl = ((bid >> 4) << 3) | (bid & 7);         // problem index inside this launch's chunk
  if (bl >= ha.nb) return;                              // padding blocks of the last group of 16 (both halves)
  const int b = ha.b0 + bl;                             // problem index inside the batch shard
  const int tile_in_prob = half * NWH + wv;             // this wave's coordinate tile
  const int j = tile_in_prob * kTile + c;               // the lane's coordinate AND its residual row
  const bool live = j < D;

  // ---- H row j and W row j in registers, as 16-byte chunks q of every 16-column group: own columns, partner's ---
  const int col0 = half * NC, ocol0 = (half ^ 1) * NC;
  const float* Hrow = ha.H + (pp.w_shared ? (size_t)0 : (size_t)b * SQ * SQ) + (size_t)j * SQ;
  const float* Wrow = pp.W + (pp.w_shared ? (size_t)0 : (size_t)b * M * D) + (size_t)j * D;
  float4 hown[NWH], hoth[NWH], wown[NWH], woth[NWH];
#pragma unroll
  for (int m = 0; m < NWH; ++m) {
    hown[m] = *reinterpret_cast<const float4*>(Hrow + col0 + 16 * m + 4 * q);      // (zero rows / columns beyond D)
    hoth[m] = *reinterpret_cast<const float4*>(Hrow + ocol0 + 16 * m + 4 * q);
    float e[4], f[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int co = col0 + 16 * m + 4 * q + k, cp = ocol0 + 16 * m + 4 * q + k;
      e[k] = (j < M && co < D) ? Wrow[co] : 0.0f;
      f[k] = (j < M && cp < D) ? Wrow[cp] : 0.0f;
    }
    wown[m] = make_float4(e[0], e[1], e[2], e[3]);
    woth[m] = make_float4(f[0], f[1], f[2], f[3]);
  }
  const float myy = j < M ? pp.y[(size_t)b * M + j] : 0.0f;
  const float myq = ha.qv[(size_t)b * SQ + j];

  // ---- per-lane persistent registers -------------------------------------
  using Core = LstmCore<PRE, true>;      // <= 4 waves per workgroup: bf16x3 gate GEMM, weights in VGPR + AGPR
  Core core;
  core.load(a.np.wpack, lane);
  core.pin();   // fragments -> AGPRs (MFMA reads them there): the VGPRs hold H, W, the state and the gate math
  __shared__ __attribute__((aligned(16))) float bias_s[Core::kBiasFloats];   // the gate biases = accumulator inits
  core.stage_bias(bias_s, a.np.wpack, tid, blockDim.x, q);   // (the handshake's __syncthreads() below orders it)
  const size_t idx = (size_t)b * D + j;
  const int tpp = (D + kTile - 1) / kTile;
  const bool tile_real = tile_in_prob < tpp;            // the padded tile of an odd tile count is idle
  TileState s;
  float* st_tile = a.st + ((size_t)b * tpp + (tile_real ? tile_in_prob : 0)) * kStateFloatsPerTile;
  if (tile_real && !a.zero_state) load_tile_state(s, st_tile, lane);
  else {
#pragma unroll
    for (int t = 0; t < kNT; ++t) s.h1[t] = s.c1[t] = s.h2[t] = s.c2[t] = 0.0f;
  }
  float xv = live ? (a.x_in ? a.x_in : a.x)[idx] : 0.0f;
  const float sc = (live && pp.x_scale) ? pp.x_scale[idx] : 1.0f;
  float cj = 0.0f;
  constexpr bool kCos = KIND == L2O_PROB_RASTRIGIN || KIND == L2O_PROB_SQUARE_COS;
  if (kCos) cj = live ? pp.C[idx] : 0.0f;
  float mv = 0.0f, vv = 0.0f;
  if (PRE == L2O_PRE_FC_ELU && !a.zero_state) { mv = live ? a.m[idx] : 0.0f; vv = live ? a.v[idx] : 0.0f; }
  float p1h = a.p1_hi, p1l = a.p1_lo, p2h = a.p2_hi, p2l = a.p2_lo;
  constexpr bool kSq = KIND == L2O_PROB_QUADRATIC || KIND == L2O_PROB_SQUARE_COS;
  const float coef = kSq ? 1.0f : 0.5f;
  const float cg = (KIND == L2O_PROB_QUADRATIC ? 2.0f : 1.0f) * pp.inv_bg;   // x2 folded in (exact)
  const float kTwoPi = pp.twopi;
  const float* xsq = xs + 4 * q;
  const float* xoq = xo + 4 * q;
  unsigned long long* mine = pa.xbuf + ((size_t)bl * 2 + half) * 2 * SQ;
  const unsigned long long* theirs = pa.xbuf + ((size_t)bl * 2 + (half ^ 1)) * 2 * SQ;
  const int slot = wv * kTile + c;                                // this lane's granule inside a (half, parity) block
  bool dead = false;                                             // partner timed out
  // ---- handshake (l2o_unroll_pair.h): same XCD?  Through the coherent path, in the slot of parity 1 that the
  // step loop first touches at step 1.
  __shared__ int same_xcd_s;
  if (tid == 0) {
    unsigned my_xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc));
    my_xcc &= 0xfu;
    const unsigned kHsTag = 0x80000000u | salt | 0xffffu;
    __hip_atomic_store(mine + SQ, ((unsigned long long)kHsTag << 32) | my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long g = 0;
    int spins = 0;
    bool ok = true;
#pragma nounroll
    for (;;) {
      g = __hip_atomic_load(theirs + SQ, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((unsigned)(g >> 32) == kHsTag) break;
      if (++spins > (1 << 20)) { ok = false; break; }       // (the step loop reports a missing partner)
      __builtin_amdgcn_s_sleep(1);
    }
    same_xcd_s = ok && pa.plain_stores && ((unsigned)g & 0xfu) == my_xcc;
  }
  __syncthreads();
  const bool same_xcd = same_xcd_s != 0;

  f32x4 acc1[kNT], acc2[kNT];
  core.init(s, q);
  core.preload(acc1, acc2);                                 // accumulator inits of the first step (the biases)
  PhaseClock pc;
  pc.start();

  const size_t hist_n = (size_t)pp.B_local * D;
  // ---- one evaluation of the optimizee at x(t): publish, both halves of H xs (-> hacc) and of W xs (-> racc; its
  // partner half is left to the caller: xo4).  LAST = the evaluation of x(T): no network step follows, no MFMAs.
  // (The unroll is a `for t < T` loop plus this evaluation once more: with the exit test in the middle of ONE loop
  //  body hipcc carried ~75 extra register copies per step through the latch.)
  auto evaluate = [&](auto last_c, int t, float xsv, float4& hacc, float4& racc, float4 (&xo4)[NWH]) {
    constexpr bool LAST = decltype(last_c)::value;
    const unsigned tag = salt | ((unsigned)t + 1u);     // (T + 1 < 65 535 when salt != 0; the handshake tag ends in 0xffff)
    const int par = t & 1;
    if (q == 0) {
      xs[slot] = xsv;
#ifndef L2O_ABLATE_PUBLISH
      // the iterate goes to the partner the moment it exists.  Same XCD (handshake): a PLAIN 8-byte store keeps the
      // granule in the XCD's L2, where the partner's sc1 (L1-bypassing) poll finds it
      if (same_xcd)
        __hip_atomic_store(mine + par * SQ + slot, pack_granule(xsv, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else
        __hip_atomic_store(mine + par * SQ + slot, pack_granule(xsv, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    }
    if (HIST && !LAST && tile_real)
      store_tile_state(s, a.hist_st + ((size_t)t * pp.B_local * tpp + (size_t)b * tpp + tile_in_prob) *
                                          kStateFloatsPerTile, lane);
    pc.mark(0);                                             // xs -> LDS + publish
    lds_barrier();                                          // B1: this half's xs complete
    pc.mark(2);
    // the partner published its iterate when we did: a first poll load goes out NOW (its round trip runs under both
    // recurrent chunks), a second one after chunk L2B in case the first was too early
    const unsigned long long* src = theirs + par * SQ + slot;
    unsigned long long g = 0, g2 = 0;
#ifdef L2O_ABLATE_EXCHANGE
    dead = true;
#endif
    if (q == 1 && !dead) g = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ---- own half of H xs and of W xs, IN the shadow of chunk L2B's MFMAs: a single wave issues in order, a
    // back-to-back MFMA holds the issue port for 16 cycles of which it needs 4 -- two FMAs ride in each gap
    // (explicit: left alone, hipcc sinks the FMAs below the next barrier and issues the 40 MFMAs back to back)
    {
      float4 x4[NWH];
      lds_read_f4<NWH>(x4, xsq);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < NWH; ++m) dot4(hown[m], x4[m], hacc);
#pragma unroll
      for (int m = 0; m < NWH; ++m) dot4(wown[m], x4[m], racc);
      if (!LAST) {
        core.template issue_l2_prev<0, Core::kTotal>(s, acc2);
#ifndef L2O_PAIRH_OWN_VPM
#define L2O_PAIRH_OWN_VPM 2     // VALU per MFMA gap: 1 .. 4 all 8.5-8.6 G (profiles/r02w_variants_own_vpm.txt)
#endif
#pragma unroll
        for (int i = 0; i < (8 * NWH + L2O_PAIRH_OWN_VPM - 1) / L2O_PAIRH_OWN_VPM; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                     // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, L2O_PAIRH_OWN_VPM, 0);     // VALU
        }
      }
      // (the sums are pinned HERE: the IR-level sinking pass otherwise moves the FMAs to their use behind B2)
      asm volatile("" : "+v"(hacc.x), "+v"(hacc.y), "+v"(hacc.z), "+v"(hacc.w), "+v"(racc.x), "+v"(racc.y), "+v"(racc.z),
                        "+v"(racc.w));
      __builtin_amdgcn_sched_barrier(0);
    }
    pc.mark(3);                                             // own-half GEMV under chunk L2B
    if (!LAST) {
      if (q == 1 && !dead) g2 = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_sched_barrier(0);
      core.template issue_l1_prev<0, Core::kTotal>(s, acc1);
    }
    if (q == 1) {
      int spins = 0;
      if (!LAST && (unsigned)(g >> 32) != tag) g = g2;
      if (!dead && (unsigned)(g >> 32) != tag) {
#pragma nounroll
        for (;;) {
          g = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((unsigned)(g >> 32) == tag) break;
          if (++spins > (1 << 20)) { dead = true; atomicExch(&pa.ws->status, 1u); break; }
#ifndef L2O_POLL_NOSLEEP
          __builtin_amdgcn_s_sleep(1);
#endif
        }
      }
      xo[slot] = __uint_as_float((unsigned)g);
    }
    pc.mark(1);                                             // chunk L1H + partner poll
    lds_barrier();                                          // B2: xo complete
    pc.mark(4);
    // ---- the partner's half: H on the critical path; W (the loss) is the caller's
    lds_read_f4<NWH>(xo4, xoq);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < NWH; ++m) dot4(hoth[m], xo4[m], hacc);
  };
  // this wave's share of f_b(x_t): rows j of this half (every row of the problem belongs to exactly one lane quad),
  // the separable terms of its coordinates
  auto loss_of = [&](float4 racc, const float4 (&xo4)[NWH], float xsv) {
#pragma unroll
    for (int m = 0; m < NWH; ++m) dot4(woth[m], xo4[m], racc);
    const float r = quad_q_sum(hsum4(racc)) - myy;           // rows >= M: W row and y are zero -> r == 0
    float contrib = q == 0 ? coef * r * r : 0.0f;
    if (live && q == 0) {
      if (KIND == L2O_PROB_LASSO) contrib += pp.l1 * __builtin_fabsf(xsv);
      if (kCos) contrib += pp.alpha - pp.alpha * cj * l2o::cos_f(kTwoPi * xsv);
    }
    return wave_sum64(contrib);
  };
  auto grad_of = [&](const float4& hacc, float xsv) {
    float gv = quad_q_sum(hsum4(hacc)) - myq;                // = (W^T (W xs - y))_j
    if (KIND == L2O_PROB_SQUARE_COS) gv *= 2.0f;            // only the ||wx-y||^2 part carries the 2
    if (KIND == L2O_PROB_LASSO) gv += pp.l1 * (xsv > 0.f ? 1.f : (xsv < 0.f ? -1.f : 0.f));
    if (kCos) gv += kTwoPi * pp.alpha * cj * l2o::sin_f(kTwoPi * xsv);
    return live ? gv * cg * sc : 0.0f;
  };
  float* const fx_wave = pa.fx_half + (size_t)bl * (2 * NWH) + half * NWH + wv;  // + t * nb * 2 NWH
  const size_t fx_stride = (size_t)ha.nb * (2 * NWH);

  for (int t = 0; t < a.T; ++t) {
    const float xsv = live ? xv * sc : 0.0f;
    float4 hacc = {0.f, 0.f, 0.f, 0.f}, racc = {0.f, 0.f, 0.f, 0.f};
    float4 xo4[NWH];
    evaluate(std::false_type(), t, xsv, hacc, racc, xo4);
    const float gv = grad_of(hacc, xsv);
    if (HIST && live && q == 0) a.hist_g[(size_t)t * hist_n + idx] = gv;
    // the loss is off the critical path: LstmCore::finish runs it between the MFMAs of chunk L2A
    float fw = 0.0f;
    auto loss = [&]() {
      fw = loss_of(racc, xo4, xsv);
      asm volatile("" : "+v"(fw));                           // (computed where it is called)
    };

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
    float d = core.template finish<false>(s, acc1, acc2, in0, in1, q, pc, loss);
    if (lane == 0) fx_wave[(size_t)t * fx_stride] = fw;
    core.refresh(s);   // marks 5 (other-half GEMV .. inputs), 6, 7, 10, 8
    if (a.np.tanh_output) {                                 // a real (uniform) branch: as a select hipcc computes the
      asm volatile("");                                      // exp + rcp of tanh on every step of the nets without it
      d = tanhf_(d);
    }
    xv = __builtin_fmaf(d, a.np.scale, xv);
    pc.mark(9);
  }
  {
    // f(x_T) (and, recording, the gradient there)
    const float xsv = live ? xv * sc : 0.0f;
    float4 hacc = {0.f, 0.f, 0.f, 0.f}, racc = {0.f, 0.f, 0.f, 0.f};
    float4 xo4[NWH];
    evaluate(std::true_type(), a.T, xsv, hacc, racc, xo4);
    const float fw = loss_of(racc, xo4, xsv);
    if (lane == 0) fx_wave[(size_t)a.T * fx_stride] = fw;
    if (HIST) {
      const float gT = grad_of(hacc, xsv);                  // (cross-lane sums: every lane takes part)
      if (live && q == 0) a.hist_gfinal[idx] = gT;
    }
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
