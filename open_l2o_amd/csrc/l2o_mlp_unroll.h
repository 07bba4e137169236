// l2o_mlp_unroll.h -- the fused persistent unroll for the neural optimizee problems.mnist
// (DM/problems.py:246-288: a [n_in -> H -> O] MLP, mean sparse-softmax cross-entropy on a fresh
// minibatch per evaluation, DM/problems.py:282-286) stepped by ONE coordinate-wise LSTM optimizer shared by
// its four variables (DM/meta.py:338-359; RNNProp DM/meta_rnnprop_train.py:371-423): T steps in ONE launch.
// Included by l2o_kernels.hip.
//
// Why: the step-granular path spends 22 us per step in three latency-bound launches (forward, backward,
// LSTM step) and re-reads the 5 MB of LSTM state and the weight fragments every step.  Here every
// 16-coordinate tile of the 15 910 coordinates (996 tiles) lives on its own SIMD for the whole unroll --
// LSTM state, Adam moments and the iterate in registers, bf16x3 fragments in AGPRs -- and the optimizee is
// evaluated by the same workgroups:
//
//   workgroup i (4 waves) owns the flat w1 coordinates [64 i, 64 i + 64) (the last workgroups own b1, w2, b2)
//   per step
//     publish   b1 / w2 / b2 owners broadcast their coordinates as {value, tag} granules        (230 values)
//     partial   P_i[s][h] = sum over the workgroup's OWN w1 coordinates (k, h) of img[s][k] w1[k][h]
//               -- the hidden pre-activation split over K; published as granules                (batch x H)
//     reduce    workgroup r sums outputs [r R, r R + R) over all partials in a FIXED order and publishes them
//     gather    every workgroup polls the batch x H sums and the small parameters, then computes the rest of
//               the forward (sigmoid / relu, layer 2, softmax, loss) and dZ, dH redundantly in LDS   (30 kMAC)
//     gradient  every lane forms the gradient of ITS coordinate from LDS (64 MACs), RNNProp / LogAndSign
//               inputs, LSTM tile step, x += delta
//
// i.e. one all-reduce of batch x H floats per step through self-validating granules (no flag, no fence, no
// grid barrier object; MI355X_MICROARCH.md "valid forms": 8-byte agent-scope atomics on both sides).  Buffer
// reuse: P is single-buffered (a workgroup can only overwrite P_i(t) after it holds all sums of step t, each of
// which was published after its reducer had consumed every P(t)); the sums and the small parameters are
// double-buffered by step parity (their publishers run at most one evaluation ahead of the slowest reader).
// Every spin is bounded; a timeout raises the sticky status word of the workspace header.
#pragma once

#ifndef L2O_MU_SLEEP2
#define L2O_MU_SLEEP2 1    // back-off between two polls of a granule pair (x 64 clocks)
#endif
#ifndef L2O_MU_SLEEP1
#define L2O_MU_SLEEP1 1
#endif
constexpr int kMuMaxBatch = 128;
constexpr int kMuMaxH = 32;
constexpr int kMuMaxO = 16;
constexpr int kMuMinH = 8;                 // (bounds the k-rows a workgroup's 64 w1 coordinates touch)
constexpr int kMuMaxKR = 64 / kMuMinH + 2; // k-rows of the image a workgroup needs per sample
constexpr int kMuMaxR = 16;                // outputs per reducing workgroup: batch * H <= 16 * workgroups
constexpr int kMuMaxG = (kMuMaxBatch * kMuMaxH + 255) / 256;   // sums gathered per thread
// ---- XCD-hierarchical all-reduce of the FAST form (round 4).  The flat protocol crosses the fabric twice per step
// (partials -> reducers, sums -> everybody: 5.0 k + 6.2 k of the step's ticks).  Workgroups are placed round-robin over
// the 8 XCDs (verified per launch through an XCC_ID handshake; any mismatch -> the flat protocol), so group g = wg % 8
// shares one L2: (A) the partials go to a reducer of the SAME XCD with plain stores (the line stays in that L2, the
// reader's L1-bypassing load finds it there -- the trick of k_unroll_pair), (B) the 8 per-XCD partial sums of an output
// cross the fabric ONCE and every XCD adds them in the same order (bit-identical sums on all XCDs), (C) the sums are
// gathered from the XCD's own copy, again through L2.  Local hop + fabric hop + local hop instead of two fabric hops.
constexpr int kMuHierG = 8;                // groups = XCDs
constexpr int kMuHierM = 32;               // members (workgroups) per group, max
constexpr int kMuHierR1Max = 48;           // outputs per local reducer, max (even)
constexpr int kMuHierS = kMuHierM * kMuHierR1Max;

struct MlpWs {                 // header of the caller-owned workspace (never cleared by the library)
  unsigned status;             // STICKY: 1 = a publisher never showed up
  unsigned seq;                // launch sequence number (tag salt), advanced by workgroup 0 at the end of a launch
  unsigned fault;              // TEST HOOK (ABI v12; 0 in production): non-zero = no workgroup waits for anybody -- the status
                               // is raised at once (the injected timeout; see PairWs in l2o_unroll_pair.h)
  unsigned pad0;
  long long ticks;             // ABI v12 (bytes 16..23): shader-clock cycles wave 0 of workgroup 0 spent in the step loop of
                               // the last launch (see PairWs)
  long long ticks_total;       // (bytes 24..31) ... from kernel entry to its last store
  unsigned pad[8];
  long long phases[16];        // phase clock dump of the -DL2O_PROFILE_PHASES build (else unused)
};

struct MlpUnrollArgs {
  NetParams np;
  int n_in, H, O, batch, act;
  const float* images;
  const int* labels;
  const int* idx;              // [T + 1][batch]
  float* x[4];                 // w1 [n_in, H], b1 [H], w2 [H, O], b2 [O]   in-out
  float* st[4];                // packed LSTM state per variable            in-out
  float* m[4];                 // RNNProp moments per variable              in-out
  float* v[4];
  const float* xscale[4];      // per-coordinate scale or NULL
  int n[4];                    // coordinates per variable
  int tile_begin[5];           // running tile count
  int T;
  float p1_hi, p1_lo, p2_hi, p2_lo;
  float* fx;                   // [T + 1]
  MlpWs* ws;
  unsigned long long* P;       // [nw1][batch * H]        partial pre-activations
  unsigned long long* S;       // [2][batch * H]          their sums, by step parity
  unsigned long long* Sm;      // [2][H + H * O + O]      b1, w2, b2 (scaled), by step parity
  int nwg, nw1, R;             // workgroups; workgroups that own w1 coordinates; outputs per reducer
  unsigned use_salt;
  // XCD-hierarchical all-reduce (round 4, FAST only; hier_R1 == 0: off): outputs per LOCAL reducer, and its buffers
  int hier_R1;
  unsigned long long* HS;      // [nwg]                         XCC_ID handshake granules
  unsigned long long* X;       // [2][8][kMuHierM][hier_R1]     per-XCD partial sums, by step parity (the ONE fabric hop)
  unsigned long long* S1;      // [2][8][kMuHierS]              every XCD's own copy of the sums, by step parity
  // HIST instantiation (l2o_mlp_unroll_record): per variable, the history the meta-gradient needs
  float* hist_st[4];           // [T][packed state]   the LSTM state BEFORE step t
  float* hist_g[4];            // [T + 1][n]          the (scaled) gradient at x_t; slot T = the gradient at x_T
  float* hist_m[4];            // [T + 1][n]          RNNProp moments AFTER step t in slot t + 1 (slot 0 untouched)
  float* hist_v[4];
};

__device__ __forceinline__ unsigned long long mu_granule(float v, unsigned tag) {
  return ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
}
// the bounded spins stay ROLLED (round 5): left alone the compiler unrolls each poll loop eight times -- 5 242 instead of
// 3 361 instructions in the step loop, 120 instead of 90 spilled SGPRs (a v_readlane per use)
#ifndef L2O_MU_POLL_UNROLLED
#define L2O_MU_POLL_PRAGMA _Pragma("nounroll")
#else
#define L2O_MU_POLL_PRAGMA
#endif
// bounded poll of one granule; returns the value, raises *dead on timeout
__device__ __forceinline__ float mu_poll(const unsigned long long* p, unsigned tag, bool& dead, unsigned* status) {
  unsigned long long g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int spins = 0;
#ifdef L2O_MU_ABL_NOWAIT   // (timing ablation: every granule of the STEP LOOP counts as arrived -- wrong numerics; the step
  if ((tag & 0xffffu) != 0xfffeu) return __uint_as_float((unsigned)g);   //  without waiting for partners; the handshake still waits)
#endif
L2O_MU_POLL_PRAGMA
  while ((unsigned)(g >> 32) != tag && !dead) {
    if (++spins > (1 << 20)) { dead = true; atomicExch(status, 2u); break; }
    __builtin_amdgcn_s_sleep(L2O_MU_SLEEP1);
    g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return __uint_as_float((unsigned)g);
}

// 16-byte (two-granule) write-through store / L1-bypassing load: MI355X_MICROARCH.md "valid forms" (16-B sc1 on both sides)
typedef unsigned mu_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mu_store2(unsigned long long* p, float v0, float v1, unsigned tag) {
  mu_u32x4 d = {__float_as_uint(v0), tag, __float_as_uint(v1), tag};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(d) : "memory");
}
// the same two granules with a PLAIN store: the line stays (dirty) in the writer's L2 -- for readers on the same XCD only
__device__ __forceinline__ void mu_store2_local(unsigned long long* p, float v0, float v1, unsigned tag) {
  mu_u32x4 d = {__float_as_uint(v0), tag, __float_as_uint(v1), tag};
  asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(d) : "memory");
}
// The same three accesses as SCALAR base + 32-bit per-lane byte offset (round 5): no 64-bit per-lane pointer exists, so there is
// nothing for LICM to hoist out of the step loop and for the allocator to spill (see `poff` in k_mlp_unroll).  `base` must be
// wave-uniform (mu_uniform makes a pointer that the compiler cannot prove uniform scalar).
// The `s_nop 4` is NOT optional: gfx9 needs five wait states between a VALU write of an SGPR and a vector-memory read of it,
// the hazard recogniser does not look into inline asm, and a base reloaded from its spill lane (v_readlane, i.e. a VALU
// SGPR write) directly in front of the asm makes the access use the STALE pair -- the -DL2O_PROFILE_PHASES build (more
// SGPR spills) died with a memory access fault exactly there.
__device__ __forceinline__ const unsigned long long* mu_uniform(const unsigned long long* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const unsigned long long*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void mu_store2_at(const unsigned long long* base, unsigned byte_off, float v0, float v1, unsigned tag) {
  mu_u32x4 d = {__float_as_uint(v0), tag, __float_as_uint(v1), tag};
  asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc1" ::"v"(byte_off), "v"(d), "s"(base) : "memory");
}
__device__ __forceinline__ void mu_store2_local_at(const unsigned long long* base, unsigned byte_off, float v0, float v1, unsigned tag) {
  mu_u32x4 d = {__float_as_uint(v0), tag, __float_as_uint(v1), tag};
  asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2" ::"v"(byte_off), "v"(d), "s"(base) : "memory");
}
__device__ __forceinline__ mu_u32x4 mu_load2_at(const unsigned long long* base, unsigned byte_off) {
  mu_u32x4 d;
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 sc1" : "=v"(d) : "v"(byte_off), "s"(base) : "memory");
  return d;
}
__device__ __forceinline__ mu_u32x4 mu_load2(const unsigned long long* p) {
  mu_u32x4 d;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(d) : "v"(p) : "memory");
  return d;
}
__device__ __forceinline__ void mu_wait_loads() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// (Measured and removed in round 5, docs/DESIGN_history_r04.md 3.3 / 8.2: holding the FIRST load of a hop back by n x 64
//  clocks so that it arrives behind the granules -- +-0.3 .. +3.5 %, the waiting is for partners that really are late;
//  LDS-only workgroup barriers in the step loop -- 1 % slower; the next minibatch's indices requested a step ahead and passed
//  through LDS -- 2 % slower; the forward tail on the VALU -- the matrix-core tail is 14 % faster.  Both memory-side ideas
//  were measured AGAIN on the spill-free kernel (profiles/archive_r05/r05j_c5_prefetch_barriers_ab.txt): indices staged a step ahead in
//  registers, all gathers issued back to back -- the phase clock's 1.8 k-cycle wait at the head of the tail goes, the kernel
//  is 1 % SLOWER (1.830 vs 1.813 ms); LDS-only barriers on top -- 2.6 % slower (1.877 ms).  Vector memory returns in order
//  and every poll ends in vmcnt(0): a minibatch gather (HBM / MALL latency) that is in flight when a hop starts sits in
//  front of the hop's granule loads.  The dependent index -> column chain and the draining barriers keep it out.)
// The step loop's workgroup barrier
__device__ __forceinline__ void mu_barrier() {
  __syncthreads();
}

// two granules at p until both carry `tag` (bounded, with back-off: a failed poll is a fabric request that competes
// with the stores it waits for)
__device__ __forceinline__ mu_u32x4 mu_poll2_at(const unsigned long long* base, unsigned byte_off, mu_u32x4 d, unsigned tag, bool& dead,
                                                unsigned* status) {
  int spins = 0;
#ifdef L2O_MU_ABL_NOWAIT
  return d;
#endif
L2O_MU_POLL_PRAGMA
  while ((d[1] != tag || d[3] != tag) && !dead) {
    if (++spins > (1 << 17)) { dead = true; atomicExch(status, 2u); break; }
    __builtin_amdgcn_s_sleep(L2O_MU_SLEEP2);
    d = mu_load2_at(base, byte_off);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  return d;
}
__device__ __forceinline__ mu_u32x4 mu_poll2(const unsigned long long* p, mu_u32x4 d, unsigned tag, bool& dead, unsigned* status) {
  int spins = 0;
#ifdef L2O_MU_ABL_NOWAIT
  return d;
#endif
L2O_MU_POLL_PRAGMA
  while ((d[1] != tag || d[3] != tag) && !dead) {
    if (++spins > (1 << 17)) { dead = true; atomicExch(status, 2u); break; }
    __builtin_amdgcn_s_sleep(L2O_MU_SLEEP2);
    d = mu_load2(p);
    mu_wait_loads();
  }
  return d;
}

// FAST: the reference's shape (hidden 20, 10 classes, minibatch 64 = one sample per lane) with static loop bounds,
// pairwise 16-byte granule traffic and a barrier-free forward tail; else the generic loops.
// HIST: also record the per-step history for the meta-gradient (l2o_mlp_unroll_record): T + 1 gradient evaluations
// (the one at x_T included), T optimizer steps.
template <int PRE, bool FAST, bool HIST = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_mlp_unroll(MlpUnrollArgs a) {
  const long long kernel_t0 = __builtin_readcyclecounter();
  __shared__ float xwg_p[32 + 64 + 64];               // the workgroup's 64 scaled coordinates, zero margins (w1 owners)
  float* xwg = xwg_p + 32;
  __shared__ float imgs[2][kMuMaxBatch][kMuMaxKR];    // the image columns the workgroup's w1 rows touch, by step parity
  __shared__ int labs[2][kMuMaxBatch];
  __shared__ __attribute__((aligned(16))) float Hs[kMuMaxBatch][kMuMaxH];          // hidden activations
  __shared__ __attribute__((aligned(16))) float dHs[kMuMaxBatch][kMuMaxH];
  __shared__ __attribute__((aligned(16))) float dZs[kMuMaxBatch][kMuMaxO];         // logits, then dZ
  __shared__ float small[kMuMaxH + kMuMaxH * kMuMaxO + kMuMaxO];   // b1 | w2 | b2 (scaled)
  __shared__ __attribute__((aligned(16))) float w2p[kMuMaxH][12];   // FAST: w2 rows padded to 48 bytes
  __shared__ float red[4][kMuMaxR];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int wg = blockIdx.x;
  const int H = a.H, O = a.O, Bn = a.batch, n_in = a.n_in;
  const int NO = Bn * H;                              // all-reduced outputs per evaluation
  const int NSM = H + H * O + O;
  const unsigned salt = a.use_salt ? ((a.ws->seq + 1u) & 0x7fffu) << 16 : 0u;
  unsigned* status = &a.ws->status;

  // ---- this wave's tile ---------------------------------------------------
  const int ti = wg * 4 + wv;
  const int ntiles = a.tile_begin[4];
  const bool tile_real = ti < ntiles;
  int var = 0;
  while (var < 3 && ti >= a.tile_begin[var + 1]) ++var;
  const int tile_in_var = tile_real ? ti - a.tile_begin[var] : 0;
  const int jl = tile_in_var * kTile + c;             // coordinate index inside the variable
  const bool live = tile_real && jl < a.n[var];
  // workgroup's w1 range: flat [j0, j0 + 64) -> image columns [k0, k0 + KR)
  const int j0 = wg * 64;
  const bool owns_w1 = wg < a.nw1;
  const int k0 = j0 / H;
  const int k_end = (min(j0 + 64, a.n[0]) - 1) / H;   // inclusive
  const int KR = owns_w1 ? k_end - k0 + 1 : 0;

  // bf16x3 gate GEMM, 6-product form, fragments pinned to AGPRs.  RNNProp's four chunks would be 240 of them; its feature
  // chunk is read from LDS instead (LstmCoreHyb, 15 KB) -- the 60 registers are what keeps the step loop out of scratch
#ifndef L2O_MU_LDS_CHUNK
#define L2O_MU_LDS_CHUNK 1
#endif
  constexpr bool kHyb = L2O_MU_LDS_CHUNK && PRE != L2O_PRE_IDENTITY;      // (LogAndSign: 40 registers of input-weight rows)
  constexpr int kHybCh = PRE == L2O_PRE_FC_ELU ? bx::kChL1X : bx::kChL2A;
  using Core = typename std::conditional<kHyb, LstmCoreHyb<PRE, kHybCh>, LstmCore<PRE, true, false>>::type;
  Core core;
  core.load(a.np.wpack, lane);
  core.pin();
  __shared__ __attribute__((aligned(16))) float bias_s[Core::kBiasFloats];   // the gate biases = accumulator inits
  core.stage_bias(bias_s, a.np.wpack, tid, 256, q);          // (ordered by the __syncthreads() of the prologue below)
  if constexpr (kHyb) {
    __shared__ __attribute__((aligned(16))) float chunk_s[LstmCoreHyb<PRE, kHybCh>::kChunkFloats];
    core.stage_chunk(chunk_s, a.np.wpack, tid, 256, lane);
  }
  f32x4 acc1[kNT], acc2[kNT];
  TileState s;
  float* st_tile = a.st[var] + (size_t)tile_in_var * kStateFloatsPerTile;
  if (tile_real) load_tile_state(s, st_tile, lane);
  else {
#pragma unroll
    for (int t5 = 0; t5 < kNT; ++t5) s.h1[t5] = s.c1[t5] = s.h2[t5] = s.c2[t5] = 0.0f;
  }
  float xv = live ? a.x[var][jl] : 0.0f;
  const float sc = (live && a.xscale[var]) ? a.xscale[var][jl] : 1.0f;
  float mv = 0.0f, vv = 0.0f;
  if (PRE == L2O_PRE_FC_ELU && live) { mv = a.m[var][jl]; vv = a.v[var][jl]; }
  float p1h = a.p1_hi, p1l = a.p1_lo, p2h = a.p2_hi, p2l = a.p2_lo;
  const int sm_off = var == 1 ? 0 : (var == 2 ? H : H + H * O);        // slot of a small-parameter coordinate
  bool dead = false;
  if (a.ws->fault != 0) {                              // (test hook: the injected timeout, see MlpWs)
    dead = true;
    if (tid == 0) atomicExch(status, 2u);
  }

  // image columns + labels of evaluation `t` into the parity buffer (synchronous form: prologue)
  auto load_eval = [&](int t, int par) {
    const int* ix = a.idx + (size_t)t * Bn;
    for (int e = tid; e < Bn * KR; e += 256) {
      const int sidx = e / KR, kk = e - sidx * KR;
      imgs[par][sidx][kk] = a.images[(size_t)ix[sidx] * n_in + k0 + kk];
    }
    for (int sidx = tid; sidx < Bn; sidx += 256) labs[par][sidx] = a.labels[ix[sidx]];
  };
  for (int e = tid; e < 160; e += 256) xwg_p[e] = 0.0f;
  for (int e = tid; e < 2 * kMuMaxBatch * kMuMaxKR; e += 256) (&imgs[0][0][0])[e] = 0.0f;
  for (int e = tid; e < kMuMaxH * 12; e += 256) (&w2p[0][0])[e] = 0.0f;
  __syncthreads();
  load_eval(0, 0);
  __syncthreads();
  // ---- XCD handshake (hierarchical all-reduce): every workgroup publishes its XCC_ID, reads everybody's and checks
  // that workgroup i sits on the XCD of workgroup i % 8.  All workgroups read the same table: one consistent decision.
  __shared__ int hier_s;
  __shared__ unsigned xcc_s[256];
  __shared__ float redh[16][kMuHierR1Max];             // stage A: [source chunk][output] partial sums
  __shared__ float redx[kMuHierG][kMuHierR1Max];       // stage B: [XCD][output] per-XCD partial sums
  bool hier = false;
  if constexpr (FAST) {
    if (a.hier_R1 > 0) {
      const unsigned kHs = 0x80000000u | salt | 0xfffeu;
      if (tid == 0) {
        unsigned my_xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc));
        __hip_atomic_store(a.HS + wg, ((unsigned long long)kHs << 32) | (my_xcc & 0xfu), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        hier_s = 1;
      }
      __syncthreads();
      if (tid < a.nwg) xcc_s[tid] = __float_as_uint(mu_poll(a.HS + tid, kHs, dead, status)) & 0xfu;
      __syncthreads();
      if (tid < a.nwg && (dead || xcc_s[tid] != xcc_s[tid & (kMuHierG - 1)])) hier_s = 0;
      __syncthreads();
      hier = hier_s != 0;
    }
  }

  const float invB = 1.0f / (float)Bn;
  // this thread's slots of the image-column panel: slot e = tid + 256 u -> (sample e / KR, k-row e % KR).  The two per-step
  // loops over the panel divided by the runtime KR ten times per step (~350 instructions); round 4: one multiply by a
  // 16-bit reciprocal (exact for e < 2^12, KR <= 10)
  constexpr int kPre = (kMuMaxBatch * kMuMaxKR + 255) / 256;
  const unsigned kr_rcp = KR > 0 ? (65536u + (unsigned)KR - 1u) / (unsigned)KR : 0u;
  auto slot_of = [&](int u) {
    const int e = tid + 256 * u;
    if (e >= Bn * KR) return -1;
    const int sidx = (int)(((unsigned)e * kr_rcp) >> 16);
    return (sidx << 8) | (e - sidx * KR);
  };
  core.init(s, q);
  core.preload(acc1, acc2);                                 // accumulator inits of the first step (the biases)
  PhaseClock pc;
  pc.start();
  // (round 5) where this thread's <= 3 granule pairs of partial pre-activations go: element offsets into P for the protocol
  // in use, computed ONCE (they hold a runtime division); every other per-thread address of the step loop is re-derived from
  // an opaque copy of the thread index each step.  Before: LICM hoisted ~30 such 64-bit addresses out of the step loop,
  // the register allocator spilled them (60 dwords of scratch in a kernel at 512 registers), and every reload came with an
  // `s_waitcnt vmcnt(0)` -- which on gfx9 also waits for the ACKs of the granule stores just issued and for the next
  // minibatch's image loads in flight: the store phase serialised on its own write acknowledgements.
  unsigned poff[3] = {0u, 0u, 0u};
  if constexpr (FAST) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int o = 2 * (tid + 256 * u);
      if (hier) {
        const int mr = o / a.hier_R1;
        poff[u] = (unsigned)((((wg & (kMuHierG - 1)) * kMuHierM + mr) * kMuHierM + (wg >> 3)) * a.hier_R1 + (o - mr * a.hier_R1));
      } else {
        const int r = o / a.R;
        poff[u] = (unsigned)((r * a.nw1 + wg) * a.R + (o - r * a.R));
      }
    }
  }
  const int tid_outer = tid;
  const long long loop_t0 = __builtin_readcyclecounter();
  for (int t = 0;; ++t) {
    int oz_step = 0;
    asm volatile("" : "+v"(oz_step));
    const int tid = tid_outer + oz_step;                // (opaque per step: see poff above)
    const int par = t & 1;
    const unsigned tag = salt | ((unsigned)t + 1u);
    const float xsv = xv * sc;
    // ---- publish: w1 owners into the workgroup's LDS row, small-parameter owners as granules
    if (q == 0 && tile_real) {
      if (var == 0) xwg[wv * kTile + c] = live ? xsv : 0.0f;
      else if (live)
        __hip_atomic_store(a.Sm + (size_t)par * NSM + sm_off + jl, mu_granule(xsv, tag), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    if (q == 0 && !tile_real && owns_w1) xwg[wv * kTile + c] = 0.0f;
    float pre_img[kPre];
    int pre_lab = 0;
    const bool have_next = t < a.T;
    // the next evaluation's image columns / labels -> registers: requested once the sums of THIS evaluation are in
    // (vector-memory returns in order: ahead of the polls the gather would delay them, and in front of the step's
    // first barrier its two dependent latencies sat on the critical path), completed under the forward tail
    auto prefetch_next = [&]() {
#ifdef L2O_MU_ABL_NOPREFETCH   // (timing ablation: no minibatch gather -- wrong numerics; what do the two dependent loads cost?)
#pragma unroll
      for (int u = 0; u < kPre; ++u) pre_img[u] = 0.25f;
      return;
#endif
      if (have_next) {
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
          pre_img[u] = 0.0f;
          const int sl = slot_of(u);
          if (sl >= 0) pre_img[u] = a.images[(size_t)a.idx[(size_t)(t + 1) * Bn + (sl >> 8)] * n_in + k0 + (sl & 0xff)];
        }
        if (tid < Bn) pre_lab = a.labels[a.idx[(size_t)(t + 1) * Bn + tid]];
      }
    };
    mu_barrier();                                   // xwg complete
    pc.mark(0);
    if constexpr (FAST) {
      constexpr int FH = 20, FO = 10, FB = 64, FNO = FB * FH;
      const int R = a.R;                               // even (host check)
      // ---- partial hidden pre-activations, two adjacent outputs (s, h), (s, h + 1) per 16-byte granule pair,
      // stored into the INBOX of the workgroup that reduces them: P[r][source][R]
      if (owns_w1) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const int pr = tid + 256 * u;
          if (pr < FNO / 2) {
            const int o = 2 * pr, sidx = o / FH, h = o - sidx * FH;
            float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll
            for (int kk = 0; kk < 5; ++kk) {           // (k-rows beyond KR: zero image columns AND zero margin of xwg)
              const float im = imgs[par][sidx][kk];
              const int jj = (k0 + kk) * FH + h - j0;  // in [-19, 99]: inside the zero margins of xwg_p
              acc0 = __builtin_fmaf(im, xwg[jj], acc0);
              acc1 = __builtin_fmaf(im, xwg[jj + 1], acc1);
            }
            // hier: the inbox of the reducer of these outputs on THIS XCD (plain store: the line stays in that L2); flat:
            // the inbox of the reducing workgroup, write-through
            if (hier) mu_store2_local_at(a.P, 8u * poff[u], acc0, acc1, tag);
            else mu_store2_at(a.P, 8u * poff[u], acc0, acc1, tag);
          }
        }
      }
      // the 60 recurrent MFMAs of the optimizer step (chunks L2B, L1H: fed by the previous step's h2, h1) ride under
      // the first hop of the all-reduce -- the first poll below would only fail anyway
      core.template issue_l2_prev<0, Core::kTotal>(s, acc2);
      core.template issue_l1_prev<0, Core::kTotal>(s, acc1);
      // ---- the small parameters (published at the start of the step: one hop, long under way)
      if (tid < FH + FH * FO + FO) {
        const float val = mu_poll(a.Sm + (size_t)par * NSM + tid, tag, dead, status);
        small[tid] = val;
        if (tid >= FH && tid < FH + FH * FO) { const int e = tid - FH; w2p[e / FO][e % FO] = val; }
      }
      pc.mark(1);
      if (hier) {
        // ---- (A) local reduce: member (g, mr) sums outputs [mr R1, mr R1 + R1) over the w1 owners of ITS XCD.
        // thread = (output pair p, source chunk ch): sources ch, ch + nch, ... in ascending order, then the chunks in
        // ascending order -- a fixed summation order.  (B) publish the XCD's partial through the fabric, collect the 8
        // XCDs' partials of the same outputs, add them in the order g = 0..7 on EVERY XCD.  (C) the sums -> this XCD's copy.
        // (indices through opaque zeros: LICM otherwise hoists every per-thread 64-bit address of this block out of the
        //  step loop and keeps it in registers for the whole unroll -- 60 more spilled registers in a kernel at its limit)
        int oz = 0, soz = 0;
        asm volatile("" : "+v"(oz));
        asm volatile("" : "+s"(soz));
        const int tid = threadIdx.x + oz, wg = blockIdx.x + soz;
        const int R1 = a.hier_R1, g = wg & (kMuHierG - 1), mr = wg >> 3;
        const int cnt = (a.nw1 - 1 - g) / kMuHierG + 1;          // w1 owners in group g
        const int o0 = mr * R1;
        const int nr = min(R1, FNO - o0);                        // <= 0: no outputs left for this member (workgroup-uniform)
        if (owns_w1 && nr > 0) {
          const int np = nr / 2, nch = min(256 / np, 16);
          const int pq = tid % np, ch = tid / np;
          float s0 = 0.0f, s1 = 0.0f;
          if (ch < nch) {
            const unsigned long long* inbox = a.P + (((size_t)g * kMuHierM + mr) * kMuHierM) * R1 + 2 * pq;
            for (int base = ch; base < cnt; base += 3 * nch) {   // up to 3 sources in flight per thread, then their tags
              mu_u32x4 d[3];
#pragma unroll
              for (int k = 0; k < 3; ++k)
                if (base + k * nch < cnt) d[k] = mu_load2(inbox + (size_t)(base + k * nch) * R1);
              mu_wait_loads();
#pragma unroll
              for (int k = 0; k < 3; ++k)
                if (base + k * nch < cnt) {
                  d[k] = mu_poll2(inbox + (size_t)(base + k * nch) * R1, d[k], tag, dead, status);
                  s0 += __uint_as_float(d[k][0]);
                  s1 += __uint_as_float(d[k][2]);
                }
            }
            redh[ch][2 * pq] = s0;
            redh[ch][2 * pq + 1] = s1;
          }
          mu_barrier();
          float x0 = 0.0f, x1 = 0.0f;
          if (tid < np) {
            for (int k = 0; k < nch; ++k) { x0 += redh[k][2 * tid]; x1 += redh[k][2 * tid + 1]; }
            mu_store2(a.X + (((size_t)par * kMuHierG + g) * kMuHierM + mr) * R1 + 2 * tid, x0, x1, tag);   // (B) the one fabric hop
            redx[g][2 * tid] = x0;
            redx[g][2 * tid + 1] = x1;
          }
          // thread = (XCD g2, pair): ONE granule pair each from the other XCDs' partials of these outputs
          if (tid < kMuHierG * np) {
            const int g2 = tid / np, p2 = tid - g2 * np;
            if (g2 != g) {
              const unsigned long long* xp = a.X + (((size_t)par * kMuHierG + g2) * kMuHierM + mr) * R1 + 2 * p2;
              mu_u32x4 d = mu_load2(xp);
              mu_wait_loads();
              d = mu_poll2(xp, d, tag, dead, status);
              redx[g2][2 * p2] = __uint_as_float(d[0]);
              redx[g2][2 * p2 + 1] = __uint_as_float(d[2]);
            }
          }
          mu_barrier();
          if (tid < np) {                                          // the same order g = 0..7 on every XCD: identical sums
            float t0 = 0.0f, t1 = 0.0f;
#pragma unroll
            for (int g2 = 0; g2 < kMuHierG; ++g2) { t0 += redx[g2][2 * tid]; t1 += redx[g2][2 * tid + 1]; }
            mu_store2_local(a.S1 + ((size_t)par * kMuHierG + g) * kMuHierS + o0 + 2 * tid, t0, t1, tag);   // (C)
          }
        } else {
          mu_barrier();
          mu_barrier();
        }
      } else
      // ---- reduce-scatter: thread = source workgroup, R granules contiguous in this workgroup's inbox
      {
        const int o0 = wg * R;
        const int nr = min(R, FNO - o0);               // <= 0: nothing to reduce here (workgroup-uniform, even)
        if (nr > 0) {
          float part[kMuMaxR];
#pragma unroll
          for (int r = 0; r < kMuMaxR; ++r) part[r] = 0.0f;
          for (int src = tid; src < a.nw1; src += 256) {
            const unsigned long long* pp = a.P + ((size_t)wg * a.nw1 + src) * R;
            mu_u32x4 g[kMuMaxR / 2];
#pragma unroll
            for (int r = 0; r < kMuMaxR / 2; ++r)
              if (2 * r < nr) g[r] = mu_load2(pp + 2 * r);
            mu_wait_loads();
#pragma unroll
            for (int r = 0; r < kMuMaxR / 2; ++r)
              if (2 * r < nr) {
                g[r] = mu_poll2(pp + 2 * r, g[r], tag, dead, status);
                part[2 * r] += __uint_as_float(g[r][0]);
                part[2 * r + 1] += __uint_as_float(g[r][2]);
              }
          }
#pragma unroll
          for (int r = 0; r < kMuMaxR; ++r)
            if (r < nr) {
              const float ws_ = wave_sum64(part[r]);
              if (lane == 0) red[wv][r] = ws_;
            }
          mu_barrier();
          if (2 * tid < nr)
            mu_store2(a.S + (size_t)par * FNO + o0 + 2 * tid,
                      (red[0][2 * tid] + red[1][2 * tid]) + (red[2][2 * tid] + red[3][2 * tid]),
                      (red[0][2 * tid + 1] + red[1][2 * tid + 1]) + (red[2][2 * tid + 1] + red[3][2 * tid + 1]), tag);
        } else {
          mu_barrier();
        }
      }
      pc.mark(2);
      // ---- gather the sums (pairs), bias + activation fused into the LDS write
      {
        const unsigned long long* Sp = mu_uniform(hier ? a.S1 + ((size_t)par * kMuHierG + (wg & (kMuHierG - 1))) * kMuHierS
                                                       : a.S + (size_t)par * FNO);
        mu_u32x4 g[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const int pr = tid + 256 * u;
          if (pr < FNO / 2) g[u] = mu_load2_at(Sp, 16u * (unsigned)pr);
        }
        mu_wait_loads();
        // (small[] was written before the barrier inside the reduce phase)
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const int pr = tid + 256 * u;
          if (pr < FNO / 2) {
            g[u] = mu_poll2_at(Sp, 16u * (unsigned)pr, g[u], tag, dead, status);
            const int o = 2 * pr, sidx = o / FH, h = o - sidx * FH;
            const float a0 = __uint_as_float(g[u][0]) + small[h], a1 = __uint_as_float(g[u][2]) + small[h + 1];
            Hs[sidx][h] = a.act == 0 ? sigmoidf_(a0) : fmaxf(a0, 0.0f);
            Hs[sidx][h + 1] = a.act == 0 ? sigmoidf_(a1) : fmaxf(a1, 0.0f);
          }
        }
      }
      mu_barrier();
      pc.mark(3);
      prefetch_next();
      pc.mark(9);                                      // (profiling build: the next minibatch's loads are requested)
      // ---- forward tail + dH on the fp32 matrix cores (round 4).  Until round 3 every wave evaluated the layer-2 forward,
      // the softmax and dH for ALL 64 samples with 400 dependent FMAs against w2 rows read from LDS one s_waitcnt at a time
      // (5.0 k + 2.7 k of the step's 27 k ticks, four-fold redundant).  Now wave w owns samples 16 w .. 16 w + 15 and runs
      // the two small products as v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation):
      //   logits  Z[class][sample] = b2 + sum_h w2[h][class] H[sample][h]    M = classes (10 of 16 rows), K = 20: 5 MFMAs
      //           D lands as lane (sample c, q) <- classes 4q + r, r = 0..3: the softmax is in-lane over r and two swaps over q
      //   dH      dHraw[h][sample] = sum_o w2[h][o] dZ[o][sample]            M = hidden (2 tiles), K-slot (kk, q) := class 4q + kk,
      //           so the B operand of k-step kk IS register kk of the dZ the lane has just formed -- no transpose: 8 MFMAs
      // dZ, dH go to LDS for the gradient phase (every workgroup needs all 64 samples of both) exactly as before.
      {
        const int s_l = 16 * wv + c;                   // this lane's sample
        f32x4 zacc;
#pragma unroll
        for (int r = 0; r < 4; ++r) zacc[r] = 4 * q + r < FO ? small[FH + FH * FO + 4 * q + r] : 0.0f;
#pragma unroll
        for (int kk = 0; kk < FH / 4; ++kk) {
          const int h = 4 * kk + q;
          const float av = c < FO ? small[FH + h * FO + c] : 0.0f;     // A[row = class c][k]: w2[h][c]
          zacc = mfma16(av, Hs[s_l][h], zacc);                         // B[k][col = sample c]
        }
        pc.drain1(zacc);
        pc.mark(10);                                   // (profiling build: the logits' five MFMAs)
        constexpr float kLog2e = 1.4426950408889634f;
        const int lab = labs[par][s_l];
        float zmax = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 4; ++r) zmax = 4 * q + r < FO ? fmaxf(zmax, zacc[r]) : zmax;
        {                                                               // max over the four q lanes of the sample
          u32x2 sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(zmax), __float_as_uint(zmax), false, false);
          zmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
          sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(zmax), __float_as_uint(zmax), false, false);
          zmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        float se = 0.0f, zl = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = 4 * q + r < FO;
          se += ok ? fast_exp2((zacc[r] - zmax) * kLog2e) : 0.0f;
          zl += (ok && 4 * q + r == lab) ? zacc[r] : 0.0f;
        }
        se = quad_q_sum(se);
        zl = quad_q_sum(zl);
        const float lse = zmax + __builtin_amdgcn_logf(se) * 0.6931471805599453f;      // v_log_f32 (log2), 1 ulp
        f32x4 dzv;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          dzv[r] = 4 * q + r < FO ? (fast_exp2((zacc[r] - lse) * kLog2e) - (4 * q + r == lab ? 1.0f : 0.0f)) * invB : 0.0f;
        *reinterpret_cast<f32x4*>(&dZs[s_l][4 * q]) = dzv;             // (classes 10..15: zeros)
        // loss: the wave's 16 samples (one copy per sample: the q == 0 row), the four waves meet in LDS
        const float lw = row_sum16(lse - zl);
        if (lane == 0) red[wv][0] = lw;
        if (t == a.T && !HIST) {
          mu_barrier();
          if (wg == 0 && tid == 0) a.fx[t] = ((red[0][0] + red[1][0]) + (red[2][0] + red[3][0])) * invB;
          pc.mark(4);
          break;
        }
        pc.mark(4);
        f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int o = 4 * q + kk;                                     // K-slot (kk, q) <-> class 4q + kk
          const float a0 = o < FO ? w2p[c][o] : 0.0f;                   // A[row = hidden c][k]
          const float a1 = (o < FO && c < FH - 16) ? w2p[16 + c][o] : 0.0f;
          d0 = mfma16(a0, dzv[kk], d0);
          d1 = mfma16(a1, dzv[kk], d1);
        }
        pc.drain1(d0);
        pc.drain1(d1);
        pc.mark(11);                                   // (profiling build: dH's eight MFMAs)
        // D: lane (sample c, q) <- hidden 4q + r (tile 0), 16 + 4q + r (tile 1: q == 0 only)
        const f32x4 hv0 = *reinterpret_cast<const f32x4*>(&Hs[s_l][4 * q]);
        f32x4 o0;
#pragma unroll
        for (int r = 0; r < 4; ++r) o0[r] = a.act == 0 ? d0[r] * hv0[r] * (1.0f - hv0[r]) : (hv0[r] > 0.0f ? d0[r] : 0.0f);
        *reinterpret_cast<f32x4*>(&dHs[s_l][4 * q]) = o0;
        if (q == 0) {
          const f32x4 hv1 = *reinterpret_cast<const f32x4*>(&Hs[s_l][16]);
          f32x4 o1;
#pragma unroll
          for (int r = 0; r < 4; ++r) o1[r] = a.act == 0 ? d1[r] * hv1[r] * (1.0f - hv1[r]) : (hv1[r] > 0.0f ? d1[r] : 0.0f);
          *reinterpret_cast<f32x4*>(&dHs[s_l][16]) = o1;
        }
      }
    } else {
    // ---- partial hidden pre-activations over the workgroup's own w1 coordinates
    if (owns_w1) {
      unsigned long long* Pw = a.P + (size_t)wg * NO;
      for (int o = tid; o < NO; o += 256) {
        const int sidx = o / H, h = o - sidx * H;
        float acc = 0.0f;
        for (int kk = 0; kk < KR; ++kk) {
          const int jj = (k0 + kk) * H + h - j0;       // position inside the workgroup's 64 coordinates
          if (jj >= 0 && jj < 64) acc = __builtin_fmaf(imgs[par][sidx][kk], xwg[jj], acc);
        }
        __hip_atomic_store(Pw + o, mu_granule(acc, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    core.template issue_l2_prev<0, Core::kTotal>(s, acc2);
    core.template issue_l1_prev<0, Core::kTotal>(s, acc1);
    pc.mark(1);                                        // partial + publish
    // ---- reduce-scatter: this workgroup sums outputs [wg R, wg R + R) over all partials, fixed order.
    // All R granules of a source are requested before the first tag is checked (one L2 / fabric round trip
    // per source, not R of them)
    {
      const int o0 = wg * a.R;
      const int nr = min(a.R, NO - o0);                // <= 0: nothing to reduce here (workgroup-uniform)
      if (nr > 0) {
        float part[kMuMaxR];
#pragma unroll
        for (int r = 0; r < kMuMaxR; ++r) part[r] = 0.0f;
        for (int src = tid; src < a.nw1; src += 256) { // (ascending source order per thread, then a fixed tree)
          const unsigned long long* pp = a.P + (size_t)src * NO + o0;
          unsigned long long g[kMuMaxR];
#pragma unroll
          for (int r = 0; r < kMuMaxR; ++r)
            if (r < nr) g[r] = __hip_atomic_load(pp + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
          for (int r = 0; r < kMuMaxR; ++r)
            if (r < nr) {
              if ((unsigned)(g[r] >> 32) != tag) g[r] = mu_granule(mu_poll(pp + r, tag, dead, status), tag);
              part[r] += __uint_as_float((unsigned)g[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < kMuMaxR; ++r)
          if (r < nr) {
            const float ws_ = wave_sum64(part[r]);
            if (lane == 0) red[wv][r] = ws_;
          }
        mu_barrier();
        if (tid < nr)
          __hip_atomic_store(a.S + (size_t)par * NO + o0 + tid,
                             mu_granule((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]), tag),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    pc.mark(2);                                        // reduce-scatter
    // ---- gather the sums and the small parameters (all requests first, then the tag checks)
    {
      const unsigned long long* Sp = a.S + (size_t)par * NO;
      const unsigned long long* Smp = a.Sm + (size_t)par * NSM;
      unsigned long long g[kMuMaxG], gs = 0;
#pragma unroll
      for (int u = 0; u < kMuMaxG; ++u) {
        const int o = tid + 256 * u;
        if (o < NO) g[u] = __hip_atomic_load(Sp + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      for (int e = tid; e < NSM; e += 256) {
        gs = __hip_atomic_load(Smp + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        small[e] = (unsigned)(gs >> 32) == tag ? __uint_as_float((unsigned)gs) : mu_poll(Smp + e, tag, dead, status);
      }
#pragma unroll
      for (int u = 0; u < kMuMaxG; ++u) {
        const int o = tid + 256 * u;
        if (o < NO) {
          const float val = (unsigned)(g[u] >> 32) == tag ? __uint_as_float((unsigned)g[u]) : mu_poll(Sp + o, tag, dead, status);
          const int sidx = o / H;
          Hs[sidx][o - sidx * H] = val;
        }
      }
    }
    mu_barrier();
    pc.mark(3);                                        // gather
    prefetch_next();
    const float* b1s = small;
    const float* w2s = small + H;
    const float* b2s = small + H + H * O;
    for (int o = tid; o < NO; o += 256) {
      const int sidx = o / H, h = o - sidx * H;
      const float av = Hs[sidx][h] + b1s[h];
      Hs[sidx][h] = a.act == 0 ? 1.0f / (1.0f + expf(-av)) : fmaxf(av, 0.0f);
    }
    mu_barrier();
    for (int e = tid; e < Bn * O; e += 256) {          // logits
      const int sidx = e / O, o = e - sidx * O;
      float z = b2s[o];
      for (int h = 0; h < H; ++h) z = __builtin_fmaf(Hs[sidx][h], w2s[h * O + o], z);
      dZs[sidx][o] = z;
    }
    mu_barrier();
    float lossn = 0.0f;
    if (tid < Bn) {                                    // softmax cross-entropy of sample tid
      const int lab = labs[par][tid];
      float zmax = dZs[tid][0];
      for (int o = 1; o < O; ++o) zmax = fmaxf(zmax, dZs[tid][o]);
      float se = 0.0f;
      for (int o = 0; o < O; ++o) se += expf(dZs[tid][o] - zmax);
      const float lse = zmax + logf(se);
      lossn = lse - dZs[tid][lab];
      for (int o = 0; o < O; ++o) dZs[tid][o] = (expf(dZs[tid][o] - lse) - (o == lab ? 1.0f : 0.0f)) * invB;
    }
    lossn = wave_sum64(lossn);
    if (lane == 0) red[wv][0] = lossn;
    mu_barrier();
    if (wg == 0 && tid == 0) a.fx[t] = ((red[0][0] + red[1][0]) + (red[2][0] + red[3][0])) * invB;
    pc.mark(4);                                        // activation, layer 2, softmax, loss
    if (t == a.T && !HIST) break;
    for (int o = tid; o < NO; o += 256) {              // dH = (dZ w2^T) * act'
      const int sidx = o / H, h = o - sidx * H;
      float d = 0.0f;
      for (int oo = 0; oo < O; ++oo) d = __builtin_fmaf(dZs[sidx][oo], w2s[h * O + oo], d);
      const float hv = Hs[sidx][h];
      dHs[sidx][h] = a.act == 0 ? d * hv * (1.0f - hv) : (hv > 0.0f ? d : 0.0f);
    }
    }
    mu_barrier();
    if (FAST && wg == 0 && tid == 0) a.fx[t] = ((red[0][0] + red[1][0]) + (red[2][0] + red[3][0])) * invB;
    pc.mark(5);                                        // dH
    // ---- the gradient of this lane's coordinate: a sum over the samples, split over the four q lanes
    float gv = 0.0f;
    if (FAST && live && var == 0) {
      // static trip count (minibatch 64: 16 samples per q lane): all 32 LDS reads in flight, four partial sums --
      // the dynamic-bound loop below waited out one LDS round trip per sample (1.385 -> 1.41 G).  The same cure for
      // the forward tail and dH (79 s_waitcnt lgkmcnt: one per w2 row) does NOT pay: batched reads need 48-60 more
      // registers in a kernel that already spills 20 (1.41 -> 1.22 G; profiles/archive_r01_r03/r02u_mlp_variants.txt)
      const int k = jl / 20, h = jl - k * 20;
      float g4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 16; ++i) g4[i & 3] = __builtin_fmaf(imgs[par][q + 4 * i][k - k0], dHs[q + 4 * i][h], g4[i & 3]);
      gv = (g4[0] + g4[1]) + (g4[2] + g4[3]);
    } else if (live) {
      if (var == 0) {
        const int k = jl / H, h = jl - k * H;
        for (int sidx = q; sidx < Bn; sidx += 4) gv = __builtin_fmaf(imgs[par][sidx][k - k0], dHs[sidx][h], gv);
      } else if (var == 1) {
        for (int sidx = q; sidx < Bn; sidx += 4) gv += dHs[sidx][jl];
      } else if (var == 2) {
        const int h = jl / O, o = jl - h * O;
        for (int sidx = q; sidx < Bn; sidx += 4) gv = __builtin_fmaf(Hs[sidx][h], dZs[sidx][o], gv);
      } else {
        for (int sidx = q; sidx < Bn; sidx += 4) gv += dZs[sidx][jl];
      }
    }
    gv = quad_q_sum(gv);
    gv = live ? gv * sc : 0.0f;
    pc.mark(6);                                        // gradient
    if constexpr (HIST) {
      if (live && q == 0) a.hist_g[var][(size_t)t * a.n[var] + jl] = gv;
      if (t == a.T) break;                             // (the gradient at x_T was still needed)
      if (tile_real)                                   // the state BEFORE this step's update
        store_tile_state(s, a.hist_st[var] + ((size_t)t * (a.tile_begin[var + 1] - a.tile_begin[var]) + tile_in_var) *
                                                 kStateFloatsPerTile, lane);
    }
    // ---- optimizer step on the tile
    float in0, in1;
    if (PRE == L2O_PRE_FC_ELU) {
      rnnprop_inputs(gv, mv, vv, a.np.beta1, a.np.beta2, a.np.omb1, a.np.omb2, 1.0f - p1h, 1.0f - p2h, in0, in1);
      if (HIST && live && q == 0) {                    // the moments AFTER the step
        a.hist_m[var][(size_t)(t + 1) * a.n[var] + jl] = mv;
        a.hist_v[var][(size_t)(t + 1) * a.n[var] + jl] = vv;
      }
      if (!live) { in0 = 0.0f; in1 = 0.0f; }
      float hi = p1h * a.np.beta1, er = __builtin_fmaf(p1h, a.np.beta1, -hi);
      float lo = __builtin_fmaf(p1l, a.np.beta1, er), sum = hi + lo;
      p1l = lo - (sum - hi); p1h = sum;
      hi = p2h * a.np.beta2; er = __builtin_fmaf(p2h, a.np.beta2, -hi);
      lo = __builtin_fmaf(p2l, a.np.beta2, er); sum = hi + lo;
      p2l = lo - (sum - hi); p2h = sum;
    } else {
      preprocess_grad<PRE>(gv, a.np.k_inv_ln2, a.np.exp_k, in0, in1);
    }
    float d = core.template finish<false>(s, acc1, acc2, in0, in1, q, pc);
    core.refresh(s);
    if (a.np.tanh_output) d = tanhf_(d);
    xv = __builtin_fmaf(d, a.np.scale, xv);
    pc.mark(7);                                        // LSTM tile step
    // ---- the prefetched next evaluation -> the other parity buffer (its previous readers finished two barriers ago)
#pragma unroll
    for (int u = 0; u < kPre; ++u) {
      const int sl = slot_of(u);
      if (sl >= 0) imgs[par ^ 1][sl >> 8][sl & 0xff] = pre_img[u];
    }
    if (tid < Bn) labs[par ^ 1][tid] = pre_lab;
    pc.mark(8);
  }
#ifdef L2O_PROFILE_PHASES
#ifndef L2O_PROFILE_WG
#define L2O_PROFILE_WG 0
#endif
  if (wg == L2O_PROFILE_WG && tid == 0) pc.dump(a.ws->phases);
#endif
  if (wg == 0 && tid == 0) a.ws->ticks = __builtin_readcyclecounter() - loop_t0;

  if (live && q == 0) {
    a.x[var][jl] = xv;
    if (PRE == L2O_PRE_FC_ELU) { a.m[var][jl] = mv; a.v[var][jl] = vv; }
  }
  if (tile_real) store_tile_state(s, st_tile, lane);
  // every workgroup read ws->seq at its start and none can finish before all have started (the first
  // all-reduce needs every partial): workgroup 0 may advance the sequence word now
  if (wg == 0 && tid == 0) {
    a.ws->seq = a.ws->seq + 1u;
    a.ws->ticks_total = __builtin_readcyclecounter() - kernel_t0;
  }
}
