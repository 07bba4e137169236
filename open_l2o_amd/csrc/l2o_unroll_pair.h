// l2o_unroll_pair.h -- the fused persistent unroll with every problem split over TWO
// workgroups (two CUs).  Included by l2o_kernels.hip after k_unroll.
//
// Why: a batch of B <= #CU/2 problems (BASELINE config 2: B = 128 on 256 CUs) leaves half
// the chip idle with one problem per CU, and the per-step critical path of a problem is
// bound by its own SIMDs' fp32 issue cycles.  Splitting the coordinates (tiles) of a problem
// over two CUs halves that path; the price is ONE exchange per step between the two partner
// workgroups: each half multiplies ITS columns of W with its part of the scaled iterate and
// the halves swap the partial residuals (one granule per row, 128 each way for d = 128).
//
// Exchange protocol (placement independent, MI355X_MICROARCH.md "valid forms"): data-tagged
// 8-byte granules {float x, uint tag = launch salt | step + 1} written with ONE agent-scope relaxed atomic
// 64-bit store (global_store_dwordx2 sc1: write-through) and polled with agent-scope relaxed
// 64-bit loads (sc1: L1 bypass).  A granule is self-validating, so no flag, no fence.  Two
// parity buffers: a workgroup can publish step t+2 only after it has consumed the partner's
// step t+1, which the partner publishes only after it consumed ours of step t+1 -- the
// buffer of parity (t+2)&1 = t&1 is therefore free.  The granule area is zeroed by a
// hipMemsetAsync ahead of every launch (tag 0 is never valid) and the tags carry the launch
// sequence number of the workspace header on top.  Every spin is bounded: on timeout the kernel
// raises the STICKY ws->status and stops waiting (results are then garbage and the host reports
// L2O_ERR_HIP) -- a non-resident partner can never hang the GPU.  Confirmed same-XCD partners
// publish with plain stores (see the handshake below; L2O_OPT_PAIR_PLAIN_STORES).
//
// The matrix never touches LDS: a half keeps its SQ x SQ/2 column block of W twice in
// registers (row-major for the partial residual, column-major for the gradient: 2 x SQ/4
// floats per lane), because with W in LDS the two GEMV passes were LDS-bandwidth bound
// (1 KB per wave ds_read_b128 x 32 per wave per pass = 1024 cycles per CU).
#pragma once

struct PairWs {               // header of the caller-owned workspace (never cleared by the library)
  unsigned status;            // 0 ok, 1 = partner timeout.  STICKY: raised by the kernel, cleared by the caller
  unsigned seq;               // launch sequence number: read by every workgroup of a launch (tag salt), advanced
                              // by k_combine_halves after it -- the library keeps no host-side launch state
  unsigned fault;             // TEST HOOK (ABI v12; 0 in production): non-zero = every workgroup behaves as if its partner
                              // never showed up -- raises `status` at once and stops polling.  Lets a test force the
                              // timeout path (and the host's recovery from it) deterministically; the caller clears it.
  unsigned pad0;
  long long ticks;            // ABI v12 (bytes 16..23): shader-clock cycles (s_memtime) wave 0 of workgroup 0 spent in the step
                              // loop of the LAST launch on this workspace -- T steps + the final loss evaluation.  What
                              // bench.py's roofline block divides its work model by: measured cycles, no clock assumption
  long long ticks_total;      // (bytes 24..31) the same wave from kernel entry to its last store: prologue and epilogue included
  unsigned pad[8];
  long long phases[16];       // phase clock dump of the -DL2O_PROFILE_PHASES build (else unused)
};
static_assert(sizeof(PairWs) == 64 + 128, "workspace header layout (include/l2o_abi.h)");

struct UnrollPairArgs {
  UnrollArgs u;
  PairWs* ws;
  unsigned long long* xbuf;   // [B][2 halves][2 parities][SQ] granules (partial residuals)
  float* fx_half;             // [(T+1)][B][2 * NWH]: one partial per (step, problem, wave)
  unsigned use_salt;          // tags carry the launch sequence in their upper bits (T + 1 < 65 535): a granule left by an
                              // earlier launch never matches, on top of the memset of the granule area ahead of every launch
  unsigned plain_stores;      // L2O_OPT_PAIR_PLAIN_STORES: a confirmed same-XCD pair publishes with plain stores
  int b0, nb;                 // this launch steps problems [b0, b0 + nb) of the shard: a batch of more than #CU / 2 problems
                              // runs as consecutive launches of <= #CU / 2 (exchange granules and loss partials are per launch)
};

__device__ __forceinline__ unsigned long long pack_granule(float v, unsigned tag) {
  return ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
}

// N ds_read_b128 at p, p + 64 B, ... issued together, then one s_waitcnt lgkmcnt(0).
// NOT directly behind MFMAs: the hazard recognizer does not see loads inside inline asm, so nothing would keep
// their data from landing in a register an in-flight MFMA still reads as SrcC (the allocator hands dead chain
// registers out at once).  Every use below sits behind an LDS barrier (>= the 18 wait states of the longest
// MFMA WAR hazard); an experiment in k_mlp_unroll that issued such reads right after 60 MFMAs produced NaNs
// (profiles/archive_r01_r03/r02u_mlp_variants.txt).
template <int N>
__device__ __forceinline__ void lds_read_f4(float4 (&v)[N], const float* p) {
  const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
  static_assert(N == 1 || N == 2 || N == 4 || N == 8, "");
  if constexpr (N == 1)
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v[0]) : "v"(addr) : "memory");
  if constexpr (N == 2)
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:64\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]) : "v"(addr) : "memory");
  if constexpr (N == 4)
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:64\n\tds_read_b128 %2, %4 offset:128\n\t"
                 "ds_read_b128 %3, %4 offset:192\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(addr) : "memory");
  if constexpr (N == 8)
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:64\n\tds_read_b128 %2, %8 offset:128\n\t"
                 "ds_read_b128 %3, %8 offset:192\n\tds_read_b128 %4, %8 offset:256\n\tds_read_b128 %5, %8 offset:320\n\t"
                 "ds_read_b128 %6, %8 offset:384\n\tds_read_b128 %7, %8 offset:448\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(addr) : "memory");
}

// Round 4: compiler-visible 16-byte LDS reads (the compiler places the s_waitcnt itself) in front of independent VALU
// work of the caller, ordered with sched_group_barrier: N ds_read_b128 back to back, then the VALU block that hides their
// latency.  (A first attempt with two asm statements -- issue / wait with the destinations as tied operands -- was WRONG:
// the register allocator split the live ranges and copied the destinations between the two statements, i.e. before the
// data had landed.)
template <int N>
__device__ __forceinline__ void lds_load_f4(l2o::f32x4 (&v)[N], const float* p) {
  const auto* q = reinterpret_cast<const __attribute__((address_space(3))) l2o::f32x4*>(
      (const __attribute__((address_space(3))) float*)p);
#pragma unroll
  for (int m = 0; m < N; ++m) v[m] = q[4 * m];             // (+64 bytes per read: the row / column chunk of tile m)
}
// (dot4 of l2o_kernels.hip with the LDS operand as an ext-vector register quad: no float4 <-> f32x4 copies)
__device__ __forceinline__ void dot4v(const float4 a, const l2o::f32x4 b, float4& acc) {
  acc.x = __builtin_fmaf(a.x, b[0], acc.x);
  acc.y = __builtin_fmaf(a.y, b[1], acc.y);
  acc.z = __builtin_fmaf(a.z, b[2], acc.z);
  acc.w = __builtin_fmaf(a.w, b[3], acc.w);
}

// The same four chains as two v_pk_fma_f32 (lanes x,y | z,w): half the GEMV's instructions; the operands are register
// quads (W packed once in the prologue, x / r straight from ds_read_b128), so the halves are sub-registers, no copies.
// hsum4pk adds in hsum4's order: bit-identical to dot4v + hsum4.  (L2O_GEMV_PK=0 restores the scalar FMAs.)
// Measured (profiles/archive_r04/r04aa_*): config 2, ONE wave per SIMD -- the wave is issue-bound and 32 fewer instructions per step are
// worth 4 % (8.78 -> 9.15 G); with TWO waves per SIMD (k_unroll_lds) the VALU pipe is the limit, a packed FMA
// occupies it twice as long, and the packed form is 1.5-3 % SLOWER: those keep the scalar FMAs.
#ifndef L2O_GEMV_PK
#define L2O_GEMV_PK 1
#endif
struct Acc4pk { l2o::bx::f32x2 lo, hi; };
__device__ __forceinline__ void dot4pk(const l2o::f32x4 a, const l2o::f32x4 b, Acc4pk& acc) {
  acc.lo = __builtin_elementwise_fma(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1), acc.lo);
  acc.hi = __builtin_elementwise_fma(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3), acc.hi);
}
__device__ __forceinline__ float hsum4pk(const Acc4pk& a) { return (a.lo.x + a.lo.y) + (a.hi.x + a.hi.y); }
__device__ __forceinline__ l2o::f32x4 as_quad(const float4 v) { return l2o::f32x4{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ void dot4q(const l2o::f32x4 a, const l2o::f32x4 b, float4& acc) {
  acc.x = __builtin_fmaf(a[0], b[0], acc.x);
  acc.y = __builtin_fmaf(a[1], b[1], acc.y);
  acc.z = __builtin_fmaf(a[2], b[2], acc.z);
  acc.w = __builtin_fmaf(a[3], b[3], acc.w);
}

// HIST: also record the per-step history for the meta-gradient (l2o_unroll_record); a template
// parameter so that the plain unroll carries none of it
// EXACT (L2O_OPT_EXACT_GATES): the fp32 MFMA gate GEMM (bit-equal to an fmaf chain) instead of the bf16x3 split
#ifndef L2O_PAIR_LDS_BARRIERS
#define L2O_PAIR_LDS_BARRIERS 0
#endif
// (round 4 also ran this body with the fragments in LDS and two workgroups per CU -- k_unroll_pair2; it measured like
//  k_unroll_lds and was removed in round 5)
template <int PRE, int KIND, int CH, bool HIST, bool EXACT>
__device__ __forceinline__ void unroll_pair_body(const UnrollPairArgs& pa) {
  const long long kernel_t0 = __builtin_readcyclecounter();
  constexpr int SQ = 16 * CH;            // padded rows (and columns) of the problem
  constexpr int NWH = CH / 2;            // waves (tiles) per half; tiles beyond the real count idle
  constexpr int NC = 16 * NWH;           // columns (coordinates) owned by a half = SQ / 2
  __shared__ __attribute__((aligned(16))) float xs[NC];   // this half's scaled iterate
  __shared__ __attribute__((aligned(16))) float rs[SQ];   // the full residual
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
  if (bl >= pa.nb) return;                              // padding blocks of the last group of 16 (both halves)
  const int b = pa.b0 + bl;                             // problem index inside the batch shard
  const int tile_in_prob = half * NWH + wv;             // this wave's coordinate tile
  // GEMV role of a lane = its LSTM role: row / column gr = c, 16-byte chunk gq = q.  The four chunk partial sums
  // of a row / column then sit on the lanes (c, 0..3) and two permlane swaps add them INTO the lanes that feed
  // the gradient to the network -- no ds_bpermute (an LDS round trip) between the g pass and the gate math.
  const int gq = q, gr = c;

  // ---- the matrix lives in registers: no LDS bandwidth in the two GEMV passes -------------
  //  wr[p][m] : row (2 wv + p) 16 + gr, own columns 16 m + 4 gq + {0..3}     (partial r = W xs)
  //  wt[m]    : own column wv 16 + gr,   rows       16 m + 4 gq + {0..3}     (g = W^T r)
  const float* Wb = pp.W + (pp.w_shared ? (size_t)0 : (size_t)b * M * D);
  const int col0 = half * NC;
  float4 wr[2][NWH], wt[CH];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int row = (2 * wv + p) * kTile + gr;
#pragma unroll
    for (int m = 0; m < NWH; ++m) {
      float e[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int col = col0 + 16 * m + 4 * gq + k;
        e[k] = (row < M && col < D) ? Wb[(size_t)row * D + col] : 0.0f;
      }
      wr[p][m] = make_float4(e[0], e[1], e[2], e[3]);
    }
  }
  {
    const int col = col0 + wv * kTile + gr;
#pragma unroll
    for (int m = 0; m < CH; ++m) {
      float e[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = 16 * m + 4 * gq + k;
        e[k] = (row < M && col < D) ? Wb[(size_t)row * D + col] : 0.0f;
      }
      wt[m] = make_float4(e[0], e[1], e[2], e[3]);
    }
  }
  constexpr bool kPk = L2O_GEMV_PK != 0;                    // packed GEMV FMAs: one wave per SIMD (see dot4pk)
  l2o::f32x4 wrq[2][NWH], wtq[CH];                          // (the same values as register quads for the packed FMAs)
#pragma unroll
  for (int m = 0; m < NWH; ++m) { wrq[0][m] = as_quad(wr[0][m]); wrq[1][m] = as_quad(wr[1][m]); }
#pragma unroll
  for (int m = 0; m < CH; ++m) wtq[m] = as_quad(wt[m]);
  // the residual rows this lane finishes: gq == 0 -> p = 0, gq == 1 -> p = 1 (gq 2, 3 idle)
  const int myrow = (2 * wv + (gq & 1)) * kTile + gr;
  const float myy = (gq < 2 && myrow < M) ? pp.y[(size_t)b * M + myrow] : 0.0f;
  const bool row_counted = gq < 2 && (half == 0 ? (myrow < NC) : (myrow >= NC));   // every row once per pair

  // ---- per-lane persistent registers -------------------------------------
  // <= 4 waves per workgroup: bf16x3 gate GEMM, weights in VGPR + AGPR
  using Core = LstmCore<PRE, !EXACT>;
  Core core;
  core.load(a.np.wpack, lane);
  core.pin();   // fragments -> AGPRs (MFMA reads them there): the VGPRs hold W, the state and the gate math
  __shared__ __attribute__((aligned(16))) float bias_s[Core::kBiasFloats];   // the gate biases = accumulator inits
  core.stage_bias(bias_s, a.np.wpack, tid, blockDim.x, q);   // (the handshake's __syncthreads() below orders it)
  const int j = tile_in_prob * kTile + c;
  const bool live = j < D;
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
  const float* xsq = xs + 4 * gq;
  const float* rsq = rs + 4 * gq;
  unsigned long long* mine = pa.xbuf + ((size_t)bl * 2 + half) * 2 * SQ;
  const unsigned long long* theirs = pa.xbuf + ((size_t)bl * 2 + (half ^ 1)) * 2 * SQ;
  bool dead = false;                                             // partner timed out
  if (pa.ws->fault != 0) {                                       // (test hook: the injected timeout, see PairWs)
    dead = true;
    if (tid == 0) atomicExch(&pa.ws->status, 1u);
  }
  // ---- handshake: do the two halves of this problem run on the same XCD?  HIP promises nothing about
  // placement (observed: block b on XCD b % 8, hence the b / b + 8 pairing above), so the halves tell each
  // other their XCC_ID once, through the coherent (agent-scope) path, in a granule slot that the step loop
  // does not touch before step 1.  Only a confirmed same-XCD pair uses the L2-resident plain stores.
  __shared__ __attribute__((aligned(16))) int same_xcd_s4[4];
  int& same_xcd_s = same_xcd_s4[0];
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
      if (dead || ++spins > (1 << 20)) { ok = false; break; }       // (the step loop reports a missing partner)
      __builtin_amdgcn_s_sleep(1);
    }
    same_xcd_s = ok && pa.plain_stores && ((unsigned)g & 0xfu) == my_xcc;
  }
  __syncthreads();
  const bool same_xcd = same_xcd_s != 0;

  f32x4 acc1[kNT], acc2[kNT];
  core.init(s, q);
  core.preload(acc1, acc2);                                 // accumulator inits of the first step (the biases)
  // (both recurrent chunks -- L1H: h1(t-1) -> layer 1, L2B: h2(t-1) -> layer 2 -- are issued inside the step loop,
  //  in the window where the wave waits for its partner's partial residuals)
  PhaseClock pc;
  pc.start();

  // Step order (round 4; round 3's order and the variants measured against it: docs/DESIGN_history_r04.md 3.1b): the scaled
  // iterate goes to LDS the moment the update exists -- at the END of a step, ahead of the split of h2 (27 VALU + the
  // register copies of the loop-carried B operands sat between the update and its LDS write: ~200 cycles of the step's
  // critical path) -- and the split runs at the top of the next step UNDER the xs reads; the loss reduction runs under the
  // residual reads of the g pass; the two row partials share one swap butterfly.
  if (q == 0) xs[wv * kTile + c] = live ? xv * sc : 0.0f;
  const size_t hist_n = (size_t)pp.B_local * D;
  const long long loop_t0 = __builtin_readcyclecounter();
  for (int t = 0;; ++t) {
    const float xsv = xv * sc;
    const unsigned tag = salt | ((unsigned)t + 1u);     // (T + 1 < 65 535 when salt != 0; the handshake tag ends in 0xffff)
    const int par = t & 1;
    pc.mark(0);
    // (recording: barriers that wait for LDS traffic only -- a __syncthreads() also waits for the write acknowledgement
    //  of the 5 KB of history the wave has just stored)
    if (HIST || L2O_PAIR_LDS_BARRIERS) lds_barrier(); else __syncthreads();          // B1: this half's xs complete
    pc.mark(2);
    // ---- partial residual over this half's columns: rows 2 x 16 per wave, all SQ rows per half
    float part;
    {
      float4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = {0.f, 0.f, 0.f, 0.f};
      Acc4pk r0p = {{0.f, 0.f}, {0.f, 0.f}}, r1p = {{0.f, 0.f}, {0.f, 0.f}};
      l2o::f32x4 x4v[NWH];
      lds_load_f4<NWH>(x4v, xsq);
      core.refresh(s);                     // split h2(t-1) -> chunk L2B operand, under the LDS latency (t = 0: repeats core.init)
      __builtin_amdgcn_sched_group_barrier(0x100, NWH, 0);     // the DS reads first ...
      __builtin_amdgcn_sched_group_barrier(0x002, 48, 0);      // ... then the split's VALU block, then the FMAs
#pragma unroll
      for (int m = 0; m < NWH; ++m) {
        if (kPk) { dot4pk(wrq[0][m], x4v[m], r0p); dot4pk(wrq[1][m], x4v[m], r1p); }
        else { dot4v(wr[0][m], x4v[m], r0); dot4v(wr[1][m], x4v[m], r1); }
      }
      // both row partials through ONE butterfly: the 16-lane swap pairs row groups (0,1) and (2,3) of p0 AND p1 at once,
      // the 32-lane swap finishes both; odd lane groups end with the p1 sum, even ones with the p0 sum -- the lanes
      // that publish them.  Same additions in the same order as two quad_q_sum calls (bit-identical), 5 instead of 13
      // instructions and one dependent swap chain instead of two.
      const float h0 = kPk ? hsum4pk(r0p) : hsum4(r0), h1 = kPk ? hsum4pk(r1p) : hsum4(r1);
      const u32x2 sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(h0), __float_as_uint(h1), false, false);
      part = xor32_add(__uint_as_float(sw[0]) + __uint_as_float(sw[1]));
    }
    // ---- exchange the partial sums (one granule per row), the previous-h2 matrix work covers the latency
    if (gq < 2) {
      // partner on the same XCD (handshake below): a PLAIN 8-byte store keeps the granule in the XCD's L2, where
      // the partner's sc1 (L1-bypassing) poll finds it; an agent-scope (sc1) store drops the line from L2
      // and the poll pays the fabric round trip (profiles: 5.55 -> 5.89 G coordinate-steps/s on config 2)
      if (same_xcd)
        __hip_atomic_store(mine + par * SQ + myrow, pack_granule(part, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else
        __hip_atomic_store(mine + par * SQ + myrow, pack_granule(part, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // (round 5) cos / sin of 2 pi x s for the rastrigin / square_cos terms: computed HERE, in front of the recurrent MFMAs
    // whose issue they interleave with, instead of behind the partner poll (the loss term) and inside the g pass (the
    // gradient term) -- both on the step's critical path.  Same function, same argument: bit-identical results.
    l2o::SinCos trig = {0.0f, 1.0f};
    if (kCos) trig = l2o::sincos_f(kTwoPi * xsv);
    pc.mark(3);                                             // partial r + publish
    // 30 MFMAs (L2B) give the partner time to publish; the first poll load goes out THEN and its L2 round trip
    // is covered by the other 30 MFMAs (L1H) -- in program order, a single wave issues in order
#ifndef L2O_PAIR_POLL_AT
#define L2O_PAIR_POLL_AT 20   // = after chunk L2B.  Packed chunks (20 MFMAs): 5 -> 7.72, 10 -> 7.86, 15 -> 7.95, 20 -> 8.02, 30 -> 7.98 G (config 2)
#endif
    constexpr int kPollAt = L2O_PAIR_POLL_AT < Core::kTotal ? L2O_PAIR_POLL_AT : Core::kTotal;   // MFMAs before the first poll load
    constexpr int kPollAt1 = L2O_PAIR_POLL_AT > Core::kTotal ? L2O_PAIR_POLL_AT - Core::kTotal : 0;
    core.template issue_l2_prev<0, kPollAt>(s, acc2);
    if (kPollAt1 > 0) core.template issue_l1_prev<0, kPollAt1>(s, acc1);
    const unsigned long long* src = theirs + par * SQ + (gq < 2 ? myrow : 0);
    unsigned long long g = 0;
#ifdef L2O_ABLATE_EXCHANGE
    dead = true;
#endif
    __builtin_amdgcn_sched_barrier(0);
    if (gq < 2 && !dead) g = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_sched_barrier(0);
    if (kPollAt < Core::kTotal) core.template issue_l2_prev<kPollAt, Core::kTotal>(s, acc2);
    core.template issue_l1_prev<kPollAt1, Core::kTotal>(s, acc1);
    float contrib = 0.0f;
    if (gq < 2) {
      int spins = 0;
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
      const float r = (part + __uint_as_float((unsigned)g)) - myy;   // rows >= M: W row and y are zero -> r == 0
      rs[myrow] = r;
      if (row_counted) contrib = coef * r * r;
    }
    if (live && q == 0) {
      if (KIND == L2O_PROB_LASSO) contrib += pp.l1 * __builtin_fabsf(xsv);
      if (kCos) contrib += pp.alpha - pp.alpha * cj * trig.c;
    }
    // the state BEFORE this step's update, for the meta-gradient.  Stored HERE: the poll above is this step's last wait
    // on vmcnt (loads and stores retire in order), the barriers of the recording kernel wait for LDS traffic only, so
    // the 5 KB per wave drain under the gate blocks instead of sitting in front of a wait (recording kernel / plain
    // kernel time at config-2 size: 1.26 -> 1.22, profiles/archive_r01_r03/r03t_*)
    // (non-temporal stores for these records: 240 -> 338 us per recording unroll -- they stall the store path)
    if (HIST && t < a.T && tile_real)
      store_tile_state(s, a.hist_st + ((size_t)t * pp.B_local * tpp + (size_t)b * tpp + tile_in_prob) *
                                          kStateFloatsPerTile, lane);
    pc.mark(1);                                             // previous-h2 MFMAs + partner poll
    if (HIST || L2O_PAIR_LDS_BARRIERS) lds_barrier(); else __syncthreads();          // B2: rs complete
    pc.mark(4);
    // this wave's share of f_b(x_t): reduced AFTER the barrier (the DPP chain fills the LDS latency of the g
    // pass instead of sitting in front of the barrier) and written straight to HBM -- no LDS round, no
    // thread-0 sum on the step's critical path; k_combine_halves adds the 2 x NWH partials per (step, problem)
    // (round 4: the residual reads of the g pass go out FIRST; the reduction's DPP chain and the store fill their latency --
    //  in round 3's ISA the chain sat in front of reads that carried their own wait)
    l2o::f32x4 rv4v[CH];
    lds_load_f4<CH>(rv4v, rsq);
    {
      const float fw = wave_sum64(contrib);
      __builtin_amdgcn_sched_group_barrier(0x100, CH, 0);      // the DS reads, then the reduction's DPP chain
      __builtin_amdgcn_sched_group_barrier(0x002, 24, 0);
      if (lane == 0) pa.fx_half[((size_t)t * pa.nb + bl) * (2 * NWH) + half * NWH + wv] = fw;
    }
    if (t == a.T && !HIST) break;

    // ---- g = W^T r for this wave's 16 coordinates ------------------------------
    // all CH residual reads are issued back to back (hipcc serialises them on one register
    // quad otherwise: CH x LDS latency on the critical path), one wait, then the FMAs
    float4 gacc4 = {0.f, 0.f, 0.f, 0.f};
    Acc4pk gaccp = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
    for (int m = 0; m < CH; ++m) {
      if (kPk) dot4pk(wtq[m], rv4v[m], gaccp);
      else dot4v(wt[m], rv4v[m], gacc4);
    }
    float gv = quad_q_sum(kPk ? hsum4pk(gaccp) : hsum4(gacc4));
    if (KIND == L2O_PROB_SQUARE_COS) gv *= 2.0f;            // only the ||wx-y||^2 part carries the 2
    if (KIND == L2O_PROB_LASSO) gv += pp.l1 * (xsv > 0.f ? 1.f : (xsv < 0.f ? -1.f : 0.f));
    if (kCos) gv += kTwoPi * pp.alpha * cj * trig.s;
    gv = live ? gv * cg * sc : 0.0f;
    if (HIST && live && q == 0) {
      if (t < a.T) a.hist_g[(size_t)t * hist_n + idx] = gv;
      else a.hist_gfinal[idx] = gv;
    }
    if (HIST && t == a.T) break;                            // (history mode: the gradient at x_T was still needed)

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
    float d = core.template finish<false, bx::NoShadow, false>(s, acc1, acc2, in0, in1, q, pc);   // (re-armed below)
    if (a.np.tanh_output) {                                 // a real (uniform) branch: as a select hipcc computes the
      asm volatile("");                                      // exp + rcp of tanh on every step of the nets without it
      d = tanhf_(d);
    }
    xv = __builtin_fmaf(d, a.np.scale, xv);
    // the next step's scaled iterate -> LDS NOW (its readers sit behind barrier B1; this step's readers of xs all
    // passed barrier B2 before any wave gets here)
    __builtin_amdgcn_sched_barrier(0);
    if (q == 0) xs[wv * kTile + c] = live ? xv * sc : 0.0f;
    __builtin_amdgcn_sched_barrier(0);
    // the next step's accumulator inits (the gate biases: 10 ds_read_b128) go out HERE: their latency overlaps the wait
    // for barrier B1, which drains this wave's LDS queue anyway
    core.preload_unpinned(acc1, acc2);
    __builtin_amdgcn_sched_barrier(0);
    pc.mark(9);
  }
#ifdef L2O_PROFILE_PHASES
  if (blockIdx.x == 0 && tid == 0) pc.dump(pa.ws->phases);
#endif
  if (bid == 0 && tid == 0) pa.ws->ticks = __builtin_readcyclecounter() - loop_t0;

  if (live && q == 0) {
    a.x[idx] = xv;
    if (PRE == L2O_PRE_FC_ELU) { a.m[idx] = mv; a.v[idx] = vv; }
  }
  if (tile_real) store_tile_state(s, st_tile, lane);
  if (bid == 0 && tid == 0) pa.ws->ticks_total = __builtin_readcyclecounter() - kernel_t0;
}

template <int PRE, int KIND, int CH, bool HIST, bool EXACT = false>
__global__ __launch_bounds__(256) void k_unroll_pair(UnrollPairArgs pa) {
  unroll_pair_body<PRE, KIND, CH, HIST, EXACT>(pa);
}

// The epilogue of a two-CU unroll, one workgroup (64 threads) per step t; runs after every workgroup of the unroll
// has finished:
//   fx_part[t][b] = sum of the 2 x NWH per-wave partials of (step t, problem b), fixed order
//   fx[t]         = (sum_b fx_part[t][b]) / B_global, the summation order of k_reduce_fx (optional)
//   the exchange granules are zeroed for the NEXT launch (tag 0 is never valid; no memset launch per unroll)
//   the launch sequence word the next launch salts its tags with advances
#ifndef L2O_TU_ILP       // (not a template: defined in the main translation unit only, see l2o_ilp_kernels.h)
__global__ __launch_bounds__(256) void k_combine_halves(const float* __restrict__ fx_half, float* __restrict__ fx_part,
                                                        int nb, int nparts, float inv_bg, float* __restrict__ fx,
                                                        unsigned long long* __restrict__ xbuf, long xwords, PairWs* ws,
                                                        int b0, int B_local) {
  // nb problems of this launch = problems [b0, b0 + nb) of the shard (fx_part rows have B_local entries).  256 threads:
  // wave 0 forms the sums (one problem per lane, the partials of a problem as 16-byte loads, all in flight), every
  // wave zeroes a share of the granules (4.8 -> 3.x us: the kernel sits between two unrolls of a bench step)
  const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  if (tid < 64) {
    float acc = 0.0f;
    for (int b = lane; b < nb; b += 64) {
      const float* p = fx_half + ((size_t)t * nb + b) * nparts;
      float f;
      if ((nparts & 3) == 0) {                           // (2 * NWH = 4 or 8 partials: whole float4s, 16-byte aligned)
        const float4 p0 = *reinterpret_cast<const float4*>(p);
        f = ((p0.x + p0.y) + p0.z) + p0.w;               // (the order of the scalar loop below)
        for (int k = 4; k < nparts; k += 4) {
          const float4 pk = *reinterpret_cast<const float4*>(p + k);
          f = (((f + pk.x) + pk.y) + pk.z) + pk.w;
        }
      } else {
        f = p[0];
        for (int k = 1; k < nparts; ++k) f += p[k];
      }
      fx_part[(size_t)t * B_local + b0 + b] = f;
      acc += f;
    }
    if (fx) {                                             // (single-launch batches only: nb == B_local)
      acc = l2o::wave_sum64(acc);
      if (lane == 0) fx[t] = acc * inv_bg;
    }
  }
  const long per = (xwords + gridDim.x - 1) / gridDim.x;
  const long lo = (long)t * per, hi = lo + per < xwords ? lo + per : xwords;
  for (long e = lo + tid; e < hi; e += 256) xbuf[e] = 0ull;
  if (t == 0 && tid == 0) ws->seq = ws->seq + 1u;
}
#endif
