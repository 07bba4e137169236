// l2o_mlp_xcd.h -- the fused persistent unroll of the neural optimizee (l2o_mlp_unroll.h; DM/problems.py:246-288 stepped
// by ONE coordinate-wise LSTM optimizer, DM/meta_rnnprop_train.py:371-423) with every optimizee INSTANCE confined to ONE
// XCD: up to eight independent unrolls per launch, one per XCD (round 6).
// Included by l2o_kernels.hip after l2o_mlp_unroll.h (its granule helpers); written for gfx950 only.
//
// Why: k_mlp_unroll puts ONE instance on the whole chip -- 996 tiles on 996 SIMDs of 249 workgroups on 8 XCDs -- and its
// step is three exchange hops (local reduce, fabric, local gather) around 670 cycles of LSTM work per tile: 54 % of the wave
// cycles are parked in polls (profiles/archive_r05/r05_counters_c5.json), roofline.frac 0.08.  That decomposition buys the lowest
// LATENCY of one unroll; it is the wrong one for THROUGHPUT (a meta-training batch of optimizees, BASELINE config 5's
// replicas), because a replica needs no partner outside its own L2:
//
//   instance j  <->  XCD j: the 32 workgroups (one per CU, eight waves) that find themselves on XCD j claim a member slot
//                    with one atomic and run instance j; workgroups on an XCD without an instance exit.  No placement is
//                    ASSUMED -- a team that does not fill (a masked or partitioned device) times out in its first
//                    bounded poll and raises the sticky status word, the host re-runs on k_mlp_unroll.
//   member m    owns the 32 consecutive tiles [32 m, 32 m + 32) of the instance's 996 (four per wave: the k_unroll_cu8
//                    arrangement -- packed bf16x3 fragments in LDS, the LSTM state of the wave's four tiles in registers,
//                    x / scale / moments of the 512 coordinates in LDS)
//   per step    partial  P_m[s][h] = sum over the member's own w1 rows of img[s][k] w1[k][h] on the fp32 matrix cores
//                        (7 v_mfma_f32_16x16x4_f32 per wave, exact fp32 products) -> the inbox of the member that reduces
//                        sample pair s / 2, PLAIN stores: the lines stay in the XCD's L2
//               reduce   member r adds the <= 31 partials of its 40 outputs in ascending source order, publishes the sums
//               gather   every member polls the 1 280 sums (L1-bypassing loads: they hit the same L2), bias + sigmoid fused
//               tail     logits / softmax / dZ / dH on the matrix cores (waves 0-3, as k_mlp_unroll's tail); the NEXT minibatch's
//                        indices / image columns are requested at the head of the step / behind the gather, landed before the LSTM phase
//               gradient G[k][h] = sum_s img[s][k] dH[s][h] for the member's 512 coordinates: 16 MFMAs per wave (waves 0-3)
//               LSTM     every wave steps its four tiles (bx::tile_step_w on the LDS fragments), x += delta
//   Two hops through ONE coherent L2 instead of three hops of which one crosses the fabric; no XCC_ID table, no flat
//   fallback protocol, no cross-XCD traffic at all.  What it costs: eight tile-steps per SIMD and step instead of one, so
//   the LATENCY of a single unroll is worse than k_mlp_unroll's -- the form is selected per call (l2o_mlp_unroll_multi).
#pragma once

namespace l2o {

constexpr int kMxMembers = 32;                    // workgroups per instance = the CUs of one XCD
constexpr int kMxThreads = 512;                   // form 8: eight waves, two per SIMD, four tiles per wave
constexpr int kMxThreads4 = 256;                  // form 4: four waves, one per SIMD, eight tiles per wave stepped in PAIRS
constexpr int kMxSlots = 32;                      // tiles per member (four per wave)
constexpr int kMxCoords = kMxSlots * kTile;       // 512 coordinate positions per member
constexpr int kMxMaxInst = 8;                     // instances per launch = XCDs
constexpr int kMxB = 64, kMxH = 20, kMxO = 10;    // the reference's shape (DM/util.py:146-155: mnist, batch 64... the FAST form)
constexpr int kMxNO = kMxB * kMxH;                // all-reduced outputs per evaluation
constexpr int kMxR = kMxNO / kMxMembers;          // outputs per reducer: 40 = two samples x 20 hidden units
constexpr int kMxKR = 28;                         // image columns a member's 512 w1 coordinates touch (<= 27), 7 MFMA k-steps
constexpr int kMxNSM = kMxH + kMxH * kMxO + kMxO; // b1 | w2 | b2
constexpr int kMxNSMp = (kMxNSM + 1) & ~1;
constexpr int kMxXwFront = 32, kMxXwBack = 64;    // zero margins of the member's scaled w1 row (index -19 .. 559)

struct MxInst {                // one optimizee instance (replica)
  const int* idx;              // [T + 1][64] minibatch indices
  float* x[4];                 // w1 [n_in, 20], b1 [20], w2 [20, 10], b2 [10]   in-out
  float* st[4];                // packed LSTM state per variable                 in-out
  float* m[4];                 // RNNProp moments per variable                   in-out
  float* v[4];
  const float* xscale[4];      // per-coordinate scale or NULL
  float* fx;                   // [T + 1]
  unsigned long long* P;       // [32 reducers][32 sources][40]  partial pre-activations (granules)
  unsigned long long* S;       // [2][1280]                      their sums, by step parity
  unsigned long long* Sm;      // [2][kMxNSMp]                   b1, w2, b2 (scaled), by step parity
};

struct MlpXcdArgs {
  NetParams np;
  int n_in, act, T, ninst;
  const float* images;
  const int* labels;
  int n[4];                    // coordinates per variable
  int tile_begin[5];           // running tile count
  int nw1;                     // members that own w1 coordinates
  float p1_hi, p1_lo, p2_hi, p2_lo;
  MlpWs* ws;
  unsigned* team;              // [8] arrival counters per XCD (zeroed by the launch)
  MxInst inst[kMxMaxInst];
};

struct MlpXcdLds {             // offsets (floats) into the dynamic LDS image behind the fragments and the input-weight rows
  int imgs, labs, xw, xL, scL, mL, vL, gL, Hs, dHs, dZs, small, w2p, red, redr, total;
};
__host__ __device__ constexpr MlpXcdLds mlp_xcd_lds(int frag_floats, int win_floats) {
  MlpXcdLds L{};
  int o = frag_floats + win_floats;
  L.imgs = o; o += 2 * kMxB * kMxKR;
  L.labs = o; o += 2 * kMxB;
  L.xw = o; o += kMxXwFront + kMxCoords + kMxXwBack;
  L.xL = o; o += kMxCoords;
  L.scL = o; o += kMxCoords;
  L.mL = o; o += kMxCoords;
  L.vL = o; o += kMxCoords;
  L.gL = o; o += kMxCoords;
  L.Hs = o; o += kMxB * kMxH;
  L.dHs = o; o += kMxB * kMxH;
  L.dZs = o; o += kMxB * 16;
  L.small = o; o += 256;
  L.w2p = o; o += kMxH * 12;
  L.red = o; o += 16;
  L.redr = o; o += kMxMembers * kMxR;
  L.total = o;
  return L;
}

// Granule loads of this kernel: request AND wait inside ONE asm statement.  l2o_mlp_unroll.h issues the load in one statement
// and the s_waitcnt in another; between the two the compiler is free to copy the (not yet written) destination and to
// hand its registers to another value, which the returning load then overwrites -- the first build of this kernel hung
// or passed depending on an unrelated diagnostic store (round 6).  Two independent loads that should overlap go out in
// one statement too.
__device__ __forceinline__ mu_u32x4 mx_load2_wait(const unsigned long long* p) {
  mu_u32x4 d;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(d) : "v"(p) : "memory");
  return d;
}
__device__ __forceinline__ void mx_load2x2_wait(const unsigned long long* p0, const unsigned long long* p1, mu_u32x4& d0, mu_u32x4& d1) {
  asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
               : "=&v"(d0), "=&v"(d1) : "v"(p0), "v"(p1) : "memory");
}
__device__ __forceinline__ void mx_load2x3_wait(const unsigned long long* p0, const unsigned long long* p1, const unsigned long long* p2,
                                                mu_u32x4& d0, mu_u32x4& d1, mu_u32x4& d2) {
  asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %4, off sc1\n\tglobal_load_dwordx4 %2, %5, off sc1\n\t"
               "s_waitcnt vmcnt(0)"
               : "=&v"(d0), "=&v"(d1), "=&v"(d2) : "v"(p0), "v"(p1), "v"(p2) : "memory");
}
// two granules at p until both carry `tag` (bounded, with back-off)
__device__ __forceinline__ mu_u32x4 mx_poll2(const unsigned long long* p, mu_u32x4 d, unsigned tag, bool& dead, unsigned* status) {
  int spins = 0;
L2O_MU_POLL_PRAGMA
  while ((d[1] != tag || d[3] != tag) && !dead) {
    if (++spins > (1 << 17)) { dead = true; atomicExch(status, 2u); break; }
    __builtin_amdgcn_s_sleep(L2O_MU_SLEEP2);
    d = mx_load2_wait(p);
  }
  return d;
}
// a store of matrix-core results: the hazard recogniser does not see the VMEM read inside an inline asm, so the wait states
// between the last MFMA that wrote `v` and the store are spelled out (18 covers the 8-pass v_mfma_f32_16x16x4_f32)
__device__ __forceinline__ void mx_settle(f32x4& v) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(v)); }
// two granules with a PLAIN store (same-XCD readers: the line stays in this L2) + the wait state gfx9 wants between a store
// of more than 8 bytes and a VALU write of its data registers (the compiler does not see the store inside the asm)
__device__ __forceinline__ void mx_store2(unsigned long long* p, float v0, float v1, unsigned tag) {
  mu_u32x4 d = {__float_as_uint(v0), tag, __float_as_uint(v1), tag};
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
}

#ifndef L2O_MX_FRAG_DEPTH
#define L2O_MX_FRAG_DEPTH 4    // fragment reads in flight ahead of their MFMA (bx::issue_pipelined; 238-249 registers, no spills)
#endif
template <int PRE>
using MxW = bx::NetWBLF<PRE, L2O_MX_FRAG_DEPTH>;

template <int PRE>
static size_t mlp_xcd_lds_bytes() {
  return sizeof(float) * (size_t)mlp_xcd_lds(LstmCoreLds<PRE, MxW<PRE>>::kFragWords, MxW<PRE>::kWinFloats).total;
}

// WV = waves per member: 8 (two per SIMD, four tiles each, one tile step at a time) or 4 (one per SIMD, eight tiles each,
// stepped two at a time: every fragment read from LDS feeds two MFMAs and the wave has two independent dependent chains to
// issue from -- what a lone wave per SIMD otherwise lacks)
template <int PRE, int WV>
__global__ __launch_bounds__(64 * WV) __attribute__((amdgpu_waves_per_eu(WV == 4 ? 1 : 2, WV == 4 ? 1 : 8))) void k_mlp_xcd(MlpXcdArgs a) {
  constexpr int kThreads = 64 * WV, kMxWaves = WV, kTPW = kMxSlots / WV;
  const long long kernel_t0 = __builtin_readcyclecounter();
  extern __shared__ __attribute__((aligned(16))) float mx_smem[];
  using Core = LstmCoreLds<PRE, MxW<PRE>>;
  constexpr MlpXcdLds LY = mlp_xcd_lds(Core::kFragWords, MxW<PRE>::kWinFloats);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4;

  // ---- team formation: which XCD is this workgroup on, and which member of that XCD's instance is it ----------------
  __shared__ int team_s[2];
  if (tid == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xfu;
    int slot = -1;
    if (xcc < (unsigned)a.ninst) slot = (int)atomicAdd(a.team + xcc, 1u);
    team_s[0] = (int)xcc;
    team_s[1] = slot;
  }
  __syncthreads();
  const int inst = __builtin_amdgcn_readfirstlane(team_s[0]);
  const int mem = __builtin_amdgcn_readfirstlane(team_s[1]);
  if (mem < 0 || mem >= kMxMembers) return;       // an XCD without an instance, or a 33rd arrival
  const MxInst& I = a.inst[inst];
  unsigned* status = &a.ws->status;
  // which wait gave up FIRST (diagnostics; workspace header pad[0]: step * 16 + phase, pad[1]: instance * 32 + member)
  bool dead_seen = false;
  auto note_dead = [&](bool dead, int t, int phase) __attribute__((always_inline)) {
    if (dead && !dead_seen) {
      dead_seen = true;
      if (atomicCAS(&a.ws->pad[0], 0u, (unsigned)(t * 16 + phase)) == 0u) a.ws->pad[1] = (unsigned)(inst * kMxMembers + mem);
    }
  };

  float* frs = mx_smem;
  float* winL = frs + Core::kFragWords;
  float (*imgs)[kMxB][kMxKR] = reinterpret_cast<float (*)[kMxB][kMxKR]>(mx_smem + LY.imgs);
  int (*labs)[kMxB] = reinterpret_cast<int (*)[kMxB]>(mx_smem + LY.labs);
  float* xw = mx_smem + LY.xw + kMxXwFront;       // the member's scaled w1 coordinates, zero margins
  float* xL = mx_smem + LY.xL;                    // x per coordinate position (16 slot + c)
  float* scL = mx_smem + LY.scL;
  float* mL = mx_smem + LY.mL;
  float* vL = mx_smem + LY.vL;
  float* gL = mx_smem + LY.gL;                    // the w1 gradients of the member's coordinates
  float (*Hs)[kMxH] = reinterpret_cast<float (*)[kMxH]>(mx_smem + LY.Hs);
  float (*dHs)[kMxH] = reinterpret_cast<float (*)[kMxH]>(mx_smem + LY.dHs);
  float (*dZs)[16] = reinterpret_cast<float (*)[16]>(mx_smem + LY.dZs);
  float* small = mx_smem + LY.small;              // b1 | w2 | b2 (scaled)
  float (*w2p)[12] = reinterpret_cast<float (*)[12]>(mx_smem + LY.w2p);
  float* red = mx_smem + LY.red;
  float (*redr)[kMxR] = reinterpret_cast<float (*)[kMxR]>(mx_smem + LY.redr);

  const int n_in = a.n_in;
  const int ntiles = a.tile_begin[4];
  // the member's w1 range: flat [j0, j1) -> image columns [k0, k0 + 28)
  const int j0 = mem * kMxCoords;
  const bool owns_w1 = mem < kMxMembers - 1 && j0 < a.n[0];
  const int k0 = j0 / kMxH;
  // slot -> tile of the instance (or ntiles: none).  Members 0 .. 30 hold the w1 tiles, 32 consecutive ones each; the LAST
  // member holds b1 | w2 | b2 (16 tiles) and nothing else: it is the only one whose gradients are VALU sums over the samples
  // and whose coordinates everybody polls at the start of a step -- with half a member's LSTM work it publishes early
  // (first build: those tiles rode behind member 30's w1 tiles and every member waited ~5 k cycles for them each step)
  const int nt1 = a.tile_begin[1];
  auto tile_of = [&](int slot) {
    const int ti = mem == kMxMembers - 1 ? nt1 + slot : mem * kMxSlots + slot;
    return (mem == kMxMembers - 1 ? ti < ntiles : ti < nt1) ? ti : ntiles;
  };
  // tile slot -> (variable, tile inside it); wave-uniform per slot
  auto var_of = [&](int ti) {
    int var = 0;
    while (var < 3 && ti >= a.tile_begin[var + 1]) ++var;
    return var;
  };

  // ---- prologue: coordinates -> LDS, LSTM state -> registers, fragments / biases / input rows -> LDS ---------------
  for (int e = tid; e < kMxXwFront + kMxCoords + kMxXwBack; e += kThreads) (xw - kMxXwFront)[e] = 0.0f;
  for (int e = tid; e < 2 * kMxB * kMxKR; e += kThreads) (&imgs[0][0][0])[e] = 0.0f;
  for (int e = tid; e < kMxH * 12; e += kThreads) (&w2p[0][0])[e] = 0.0f;
  __syncthreads();
#pragma unroll
  for (int pi = 0; pi < kMxCoords / kThreads; ++pi) {      // (static trip count: 1 | 2)
    const int pos = tid + kThreads * pi;
    const int ti = tile_of(pos >> 4);
    float xv = 0.0f, sc = 1.0f, mv = 0.0f, vv = 0.0f;
    bool w1pos = false;
    if (ti < ntiles) {
      const int var = var_of(ti);
      const int jl = (ti - a.tile_begin[var]) * kTile + (pos & 15);
      if (jl < a.n[var]) {
        xv = I.x[var][jl];
        if (I.xscale[var]) sc = I.xscale[var][jl];
        if (PRE == L2O_PRE_FC_ELU) { mv = I.m[var][jl]; vv = I.v[var][jl]; }
        w1pos = var == 0;
      }
    }
    xL[pos] = xv; scL[pos] = sc; mL[pos] = mv; vL[pos] = vv; gL[pos] = 0.0f;
    if (w1pos) xw[pos] = xv * sc;
  }
  TileState sr[kTPW];                                      // tile slots wv + WV k
#pragma unroll
  for (int k = 0; k < kTPW; ++k) {
#pragma unroll
    for (int t5 = 0; t5 < kNT; ++t5) { sr[k].h1[t5] = 0.f; sr[k].c1[t5] = 0.f; sr[k].h2[t5] = 0.f; sr[k].c2[t5] = 0.f; }
    const int ti = tile_of(wv + kMxWaves * k);
    if (ti < ntiles) {
      const int var = var_of(ti);
      load_tile_state(sr[k], I.st[var] + (size_t)(ti - a.tile_begin[var]) * kStateFloatsPerTile, lane);
    }
  }
  Core core;
  core.load(a.np.wpack, lane);
  core.stage_frags(frs, a.np.wpack, tid, kThreads, lane);
  __shared__ __attribute__((aligned(16))) float bias_s[Core::kBiasFloats];
  core.stage_bias(bias_s, a.np.wpack, tid, kThreads, q);
  bx::stage_win(core.w, winL, a.np.wpack, tid, kThreads, lane);
  float p1h = a.p1_hi, p1l = a.p1_lo, p2h = a.p2_hi, p2l = a.p2_lo;
  float om1 = 1.0f, om2 = 1.0f;
  bool dead = false;
  if (a.ws->fault != 0) {                                  // (test hook: the injected timeout, see MlpWs)
    dead = true;
    if (tid == 0) atomicExch(status, 2u);
  }
  // image columns + labels of evaluation t into parity buffer par: slot e = first + NTHR u -> (sample e / 28, column e % 28).
  // Static trip count: all index loads go out together, then all column loads (two dependent latencies per call, not 2 NU)
  auto load_eval = [&](int t, int par, int first, auto nthr_c) __attribute__((always_inline)) {
    constexpr int NTHR = decltype(nthr_c)::value;
    constexpr int NU = (kMxB * kMxKR + NTHR - 1) / NTHR;
    const int* ix = I.idx + (size_t)t * kMxB;
    int row[NU];
    float val[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int e = first + NTHR * u;
      row[u] = e < kMxB * kMxKR ? ix[e / kMxKR] : 0;
    }
    int lab = first < kMxB ? ix[first] : 0;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int e = first + NTHR * u, kk = e % kMxKR;
      val[u] = (e < kMxB * kMxKR && owns_w1 && k0 + kk < n_in) ? a.images[(size_t)row[u] * n_in + k0 + kk] : 0.0f;
    }
    if (first < kMxB) lab = a.labels[lab];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int e = first + NTHR * u;
      if (e < kMxB * kMxKR) imgs[par][e / kMxKR][e % kMxKR] = val[u];
    }
    if (first < kMxB) labs[par][first] = lab;
  };
  load_eval(0, 0, tid, std::integral_constant<int, kThreads>());
  const float invB = 1.0f / (float)kMxB;
  const int nw1 = a.nw1;
  PhaseClock pc;
  pc.start();
  __syncthreads();

  const int tid_outer = tid;
  const long long loop_t0 = __builtin_readcyclecounter();
  for (int t = 0;; ++t) {
    const int par = t & 1;
    const unsigned tag = (unsigned)t + 1u;
    // the thread index, re-made opaque every step: LICM otherwise hoists every per-lane 64-bit address of the step (inboxes,
    // sums, LDS positions of four inlined tile bodies) out of the loop and holds it for the whole unroll -- in a kernel
    // whose 80 registers of LSTM state already live there (the cure of k_mlp_unroll / k_unroll_cu8)
    int tid = tid_outer;
    asm volatile("" : "+v"(tid));
    const int cc = tid & 15, q = (tid >> 4) & 3, lane = tid & 63;
    // ---- publish: the owners of b1 / w2 / b2 broadcast their scaled coordinates as granules (plain stores: same L2)
#pragma unroll
    for (int k = 0; k < kTPW; ++k) {
      const int ti = tile_of(wv + kMxWaves * k);
      if (ti >= a.tile_begin[1] && ti < ntiles && q == 0) {
        const int var = var_of(ti);
        const int jl = (ti - a.tile_begin[var]) * kTile + cc;
        if (jl < a.n[var]) {
          const int pos = (wv + kMxWaves * k) * kTile + cc;
          const int sm_off = var == 1 ? 0 : (var == 2 ? kMxH : kMxH + kMxH * kMxO);
          __hip_atomic_store(I.Sm + (size_t)par * kMxNSMp + sm_off + jl, mu_granule(xL[pos] * scL[pos], tag), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    // ---- the NEXT evaluation's minibatch, stage A: its indices -> registers (stage B, behind the gather: the image columns;
    // stage C, in front of the LSTM phase: -> the other parity buffer).  Two dependent global latencies, neither of them
    // waited for where it is issued.
    constexpr int kPf = (kMxB * kMxKR + kThreads - 1) / kThreads;
    int pf_row[kPf], pf_lab = 0;
    float pf_val[kPf];
    const bool have_next = t < a.T;
    if (have_next) {
      const int* ix1 = I.idx + (size_t)(t + 1) * kMxB;
#pragma unroll
      for (int u = 0; u < kPf; ++u) {
        const int e = tid + kThreads * u;
        pf_row[u] = e < kMxB * kMxKR ? ix1[e / kMxKR] : 0;
      }
      if (tid < kMxB) pf_lab = ix1[tid];
    }
    pc.mark(0);
    // ---- partial hidden pre-activations on the fp32 matrix cores: wave = (sample tile st, hidden tile ht);
    // D lands as lane (sample 16 st + c, q) <- hidden units 16 ht + 4 q + r: four consecutive outputs of one sample ->
    // two granule pairs in the inbox of the member that reduces sample pair s / 2
    if (owns_w1)
#pragma unroll
    for (int ci = 0; ci < 8 / kMxWaves; ++ci) {              // eight (sample tile, hidden tile) products over the member's waves
      const int cb = wv + kMxWaves * ci;
      const int st_ = cb & 3, ht = cb >> 2;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const bool arow = ht == 0 || cc < kMxH - 16;         // hidden rows 16 .. 19 only
#pragma unroll
      for (int kk = 0; kk < kMxKR / 4; ++kk) {
        const int krow = 4 * kk + q;
        const float av = arow ? xw[(k0 + krow) * kMxH + 16 * ht + cc - j0] : 0.0f;     // A[row = hidden][k]
        acc = mfma16(av, imgs[par][16 * st_ + cc][krow], acc);                         // B[k][col = sample]
      }
      mx_settle(acc);
      if (ht == 0 || q == 0) {
        const int sidx = 16 * st_ + cc;
        unsigned long long* dst = I.P + ((size_t)((sidx >> 1) * kMxMembers + mem)) * kMxR + (sidx & 1) * kMxH + 16 * ht + 4 * q;
        mx_store2(dst, acc[0], acc[1], tag);
        mx_store2(dst + 2, acc[2], acc[3], tag);
      }
    }
    // ---- the small parameters (published at the start of the step)
    if (tid < kMxNSM) {
      const float val = mu_poll(I.Sm + (size_t)par * kMxNSMp + tid, tag, dead, status);
      small[tid] = val;
      if (tid >= kMxH && tid < kMxH + kMxH * kMxO) { const int e = tid - kMxH; w2p[e / kMxO][e % kMxO] = val; }
    }
    pc.mark(1);
    note_dead(dead, t, 2);
    // ---- reduce: this member's 40 outputs over the w1 owners: thread = (pair p, source), all loads first, then the tags
    {
      constexpr int kSrcRound = kThreads / (kMxR / 2);                                  // sources per round: 25 (form 8) | 12 (form 4)
      constexpr int kRounds = (kMxMembers - 1 + kSrcRound - 1) / kSrcRound;             // 2 | 3
      const int pq = tid % (kMxR / 2), s0 = tid / (kMxR / 2);
      const unsigned long long* inbox = I.P + ((size_t)mem * kMxMembers) * kMxR + 2 * pq;
      const bool vt = tid < kSrcRound * (kMxR / 2);
      // (all loads unconditionally, from a clamped source: no conditionally defined asm results)
      const bool v0 = vt && s0 < nw1, v1 = vt && s0 + kSrcRound < nw1, v2 = kRounds > 2 && vt && s0 + 2 * kSrcRound < nw1;
      const unsigned long long* p0 = inbox + (size_t)(v0 ? s0 : 0) * kMxR;
      const unsigned long long* p1 = inbox + (size_t)(v1 ? s0 + kSrcRound : 0) * kMxR;
      const unsigned long long* p2 = inbox + (size_t)(v2 ? s0 + 2 * kSrcRound : 0) * kMxR;
      mu_u32x4 d0, d1, d2;
      if constexpr (kRounds == 2) mx_load2x2_wait(p0, p1, d0, d1);
      else mx_load2x3_wait(p0, p1, p2, d0, d1, d2);
      if (v0) {
        d0 = mx_poll2(p0, d0, tag, dead, status);
        redr[s0][2 * pq] = __uint_as_float(d0[0]);
        redr[s0][2 * pq + 1] = __uint_as_float(d0[2]);
      }
      if (v1) {
        d1 = mx_poll2(p1, d1, tag, dead, status);
        redr[s0 + kSrcRound][2 * pq] = __uint_as_float(d1[0]);
        redr[s0 + kSrcRound][2 * pq + 1] = __uint_as_float(d1[2]);
      }
      if constexpr (kRounds > 2) {
        if (v2) {
          d2 = mx_poll2(p2, d2, tag, dead, status);
          redr[s0 + 2 * kSrcRound][2 * pq] = __uint_as_float(d2[0]);
          redr[s0 + 2 * kSrcRound][2 * pq + 1] = __uint_as_float(d2[2]);
        }
      }
      lds_barrier();
      if (tid < kMxR / 2) {                                 // ascending source order: the same sum whoever computes it
        float t0 = 0.0f, t1 = 0.0f;
        for (int s = 0; s < nw1; ++s) { t0 += redr[s][2 * tid]; t1 += redr[s][2 * tid + 1]; }
        mx_store2(I.S + (size_t)par * kMxNO + mem * kMxR + 2 * tid, t0, t1, tag);
      }
    }
    pc.mark(2);
    note_dead(dead, t, 3);
    // ---- gather the 1 280 sums (pairs), bias + activation fused into the LDS write
    {
      const unsigned long long* Sp = I.S + (size_t)par * kMxNO;
      constexpr int kG = (kMxNO / 2 + kThreads - 1) / kThreads;                         // pairs per thread: 2 | 3
      const bool w0 = tid < kMxNO / 2, w1 = tid + kThreads < kMxNO / 2, w2 = kG > 2 && tid + 2 * kThreads < kMxNO / 2;
      const unsigned long long* q0 = Sp + 2 * (w0 ? tid : 0);
      const unsigned long long* q1 = Sp + 2 * (w1 ? tid + kThreads : 0);
      const unsigned long long* q2 = Sp + 2 * (w2 ? tid + 2 * kThreads : 0);
      mu_u32x4 g0, g1, g2;
      if constexpr (kG == 2) mx_load2x2_wait(q0, q1, g0, g1);
      else mx_load2x3_wait(q0, q1, q2, g0, g1, g2);
      auto put = [&](int pr, mu_u32x4 g) {
        const int o = 2 * pr, sidx = o / kMxH, h = o - sidx * kMxH;
        const float a0 = __uint_as_float(g[0]) + small[h], a1 = __uint_as_float(g[2]) + small[h + 1];
        Hs[sidx][h] = a.act == 0 ? sigmoidf_(a0) : fmaxf(a0, 0.0f);
        Hs[sidx][h + 1] = a.act == 0 ? sigmoidf_(a1) : fmaxf(a1, 0.0f);
      };
      // (small[] was written before the barrier inside the reduce phase)
      if (w0) {
        g0 = mx_poll2(q0, g0, tag, dead, status);
        put(tid, g0);
      }
      if (w1) {
        g1 = mx_poll2(q1, g1, tag, dead, status);
        put(tid + kThreads, g1);
      }
      if constexpr (kG > 2) {
        if (w2) {
          g2 = mx_poll2(q2, g2, tag, dead, status);
          put(tid + 2 * kThreads, g2);
        }
      }
    }
    if (have_next) {                                        // stage B: the image columns of the next minibatch
#pragma unroll
      for (int u = 0; u < kPf; ++u) {
        const int e = tid + kThreads * u, kk = e % kMxKR;
        pf_val[u] = (e < kMxB * kMxKR && owns_w1 && k0 + kk < n_in) ? a.images[(size_t)pf_row[u] * n_in + k0 + kk] : 0.0f;
      }
      if (tid < kMxB) pf_lab = a.labels[pf_lab];
    }
    __syncthreads();
    pc.mark(3);
    note_dead(dead, t, 4);
    // ---- waves 0-3: forward tail + dH on the fp32 matrix cores (k_mlp_unroll's tail: wave w owns samples 16 w .. 16 w + 15)
    if (wv < 4) {
      const int s_l = 16 * wv + cc;
      f32x4 zacc;
#pragma unroll
      for (int r = 0; r < 4; ++r) zacc[r] = 4 * q + r < kMxO ? small[kMxH + kMxH * kMxO + 4 * q + r] : 0.0f;
#pragma unroll
      for (int kk = 0; kk < kMxH / 4; ++kk) {
        const int h = 4 * kk + q;
        const float av = cc < kMxO ? small[kMxH + h * kMxO + cc] : 0.0f;    // A[row = class c][k]: w2[h][c]
        zacc = mfma16(av, Hs[s_l][h], zacc);                                  // B[k][col = sample c]
      }
      constexpr float kLog2e = 1.4426950408889634f;
      const int lab = labs[par][s_l];
      float zmax = -3.0e38f;
#pragma unroll
      for (int r = 0; r < 4; ++r) zmax = 4 * q + r < kMxO ? fmaxf(zmax, zacc[r]) : zmax;
      {
        u32x2 sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(zmax), __float_as_uint(zmax), false, false);
        zmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(zmax), __float_as_uint(zmax), false, false);
        zmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      }
      float se = 0.0f, zl = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = 4 * q + r < kMxO;
        se += ok ? fast_exp2((zacc[r] - zmax) * kLog2e) : 0.0f;
        zl += (ok && 4 * q + r == lab) ? zacc[r] : 0.0f;
      }
      se = quad_q_sum(se);
      zl = quad_q_sum(zl);
      const float lse = zmax + __builtin_amdgcn_logf(se) * 0.6931471805599453f;
      f32x4 dzv;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        dzv[r] = 4 * q + r < kMxO ? (fast_exp2((zacc[r] - lse) * kLog2e) - (4 * q + r == lab ? 1.0f : 0.0f)) * invB : 0.0f;
      *reinterpret_cast<f32x4*>(&dZs[s_l][4 * q]) = dzv;                      // (classes 10..15: zeros)
      const float lw = row_sum16(lse - zl);
      if (lane == 0) red[wv] = lw;
      if (t < a.T) {
        f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int o = 4 * q + kk;                                            // K-slot (kk, q) <-> class 4 q + kk
          const float a0 = o < kMxO ? w2p[cc][o] : 0.0f;                       // A[row = hidden c][k]
          const float a1 = (o < kMxO && cc < kMxH - 16) ? w2p[16 + cc][o] : 0.0f;
          d0 = mfma16(a0, dzv[kk], d0);
          d1 = mfma16(a1, dzv[kk], d1);
        }
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
    }
    __syncthreads();
    if (mem == 0 && tid == 0) I.fx[t] = ((red[0] + red[1]) + (red[2] + red[3])) * invB;
    pc.mark(4);
    if (t == a.T) break;
    // ---- the w1 gradients of the member's coordinates on the matrix cores: wave = (row tile mt, hidden tile nt), K = the
    // 64 samples; D lands as lane (hidden 16 nt + c, q) <- image columns 16 mt + 4 q + r
    if (wv < 4 && owns_w1) {
      const int mt = wv & 1, nt = wv >> 1;
      const bool brow = nt == 0 || cc < kMxH - 16;
      const bool acol = 16 * mt + cc < kMxKR;
      f32x4 gacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < kMxB / 4; ++kk) {
        const int sidx = 4 * kk + q;
        const float av = acol ? imgs[par][sidx][16 * mt + cc] : 0.0f;        // A[row = image column][k = sample]
        const float bv = brow ? dHs[sidx][16 * nt + cc] : 0.0f;               // B[k = sample][col = hidden]
        gacc = mfma16(av, bv, gacc);
      }
      if (brow) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int local = (k0 + 16 * mt + 4 * q + r) * kMxH + 16 * nt + cc - j0;
          if (16 * mt + 4 * q + r < kMxKR && local >= 0 && local < kMxCoords) gL[local] = gacc[r];
        }
      }
    }
    // stage C: the next minibatch -> the other parity buffer (nobody reads it before the next step's partial phase)
    if (have_next) {
#pragma unroll
      for (int u = 0; u < kPf; ++u) {
        const int e = tid + kThreads * u;
        if (e < kMxB * kMxKR) imgs[par ^ 1][e / kMxKR][e % kMxKR] = pf_val[u];
      }
      if (tid < kMxB) labs[par ^ 1][tid] = pf_lab;
    }
    __syncthreads();
    pc.mark(5);
    // ---- optimizer network on this wave's four tiles
    if (PRE == L2O_PRE_FC_ELU) { om1 = 1.0f - p1h; om2 = 1.0f - p2h; }
    // the gradients of b1 / w2 / b2 (16 tiles of the whole instance: members 30 and 31): a sum over the samples, split over
    // the q lanes, into the same gL slots the w1 tiles read (same wave writes and reads: LDS operations of a wave are ordered)
#pragma unroll 1
    for (int k = 0; k < kTPW; ++k) {
      const int slot = wv + kMxWaves * k;
      const int ti = tile_of(slot);
      if (ti < a.tile_begin[1] || ti >= ntiles) continue;
      const int var = var_of(ti);
      const int jl = (ti - a.tile_begin[var]) * kTile + cc;
      float gv = 0.0f;
      if (jl < a.n[var]) {
        if (var == 1) {
          for (int sidx = q; sidx < kMxB; sidx += 4) gv += dHs[sidx][jl];
        } else if (var == 2) {
          const int h = jl / kMxO, o = jl - h * kMxO;
          for (int sidx = q; sidx < kMxB; sidx += 4) gv = __builtin_fmaf(Hs[sidx][h], dZs[sidx][o], gv);
        } else {
          for (int sidx = q; sidx < kMxB; sidx += 4) gv += dZs[sidx][jl];
        }
      }
      gv = quad_q_sum(gv);
      if (q == 0) gL[slot * kTile + cc] = gv;
    }
    if constexpr (WV == 8) {                                 // one tile step at a time, two waves per SIMD fill each other's stalls
      // (measured and removed, profiles/r06_c5_xcd_forms_ab.txt: starting waves 4-7 of a member 0.9 / 1.8 / 2.7 k cycles late --
      //  the two waves of a SIMD leave the barrier together and run the same sequence -- costs exactly the delay: 4.68 / 4.71 /
      //  4.82 ms against 4.57)
      auto do_tile = [&](int k, TileState& s) __attribute__((always_inline)) {
        const int slot = wv + kMxWaves * k;
        const int ti = tile_of(slot);
        if (ti >= ntiles) return;
        const bool is_w1 = ti < a.tile_begin[1];
        const int var = var_of(ti);
        const bool live = (ti - a.tile_begin[var]) * kTile + cc < a.n[var];
        const int pos = slot * kTile + cc;
        const float sc = scL[pos], xj = xL[pos];
        const float gv = live ? gL[pos] * sc : 0.0f;
        float in0, in1;
        if (PRE == L2O_PRE_FC_ELU) {
          float m = mL[pos], v = vL[pos];
          rnnprop_inputs(gv, m, v, a.np.beta1, a.np.beta2, a.np.omb1, a.np.omb2, om1, om2, in0, in1);
          if (!live) { in0 = 0.0f; in1 = 0.0f; }
          if (q == 0) { mL[pos] = m; vL[pos] = v; }
        } else {
          preprocess_grad<PRE>(gv, a.np.k_inv_ln2, a.np.exp_k, in0, in1);
        }
        float d = bx::tile_step_w<PRE, true, MxW<PRE>>(core.w, s, in0, in1, q);
        if (a.np.tanh_output) d = tanhf_(d);
        const float xn = __builtin_fmaf(d, a.np.scale, xj);
        if (live && q == 0) {
          xL[pos] = xn;
          if (is_w1) xw[pos] = xn * sc;
        }
      };
      static_for<0, kTPW>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        do_tile(k, sr[k]);
      });
    } else {                                                 // a lone wave per SIMD: two tiles per step (bx::tile_step2_w)
    // what a tile's lanes bring to / take from its step: gradient -> the net's two inputs (the moments ride in LDS), and
    // the update of x / x s behind it
    struct TileIo { int pos; bool live, is_w1, real; float sc, xj, in0, in1; };
    auto tile_in = [&](int k) __attribute__((always_inline)) {
      TileIo io;
      const int slot = wv + kMxWaves * k;
      const int ti = tile_of(slot);
      io.real = ti < ntiles;
      io.is_w1 = ti < a.tile_begin[1];
      const int var = io.real ? var_of(ti) : 0;
      io.live = io.real && (ti - a.tile_begin[var]) * kTile + cc < a.n[var];
      io.pos = slot * kTile + cc;
      io.sc = scL[io.pos];
      io.xj = xL[io.pos];
      const float gv = io.live ? gL[io.pos] * io.sc : 0.0f;
      if (PRE == L2O_PRE_FC_ELU) {
        float m = mL[io.pos], v = vL[io.pos];
        rnnprop_inputs(gv, m, v, a.np.beta1, a.np.beta2, a.np.omb1, a.np.omb2, om1, om2, io.in0, io.in1);
        if (!io.live) { io.in0 = 0.0f; io.in1 = 0.0f; }
        if (q == 0 && io.real) { mL[io.pos] = m; vL[io.pos] = v; }
      } else {
        preprocess_grad<PRE>(gv, a.np.k_inv_ln2, a.np.exp_k, io.in0, io.in1);
      }
      return io;
    };
    auto tile_out = [&](const TileIo& io, float d) __attribute__((always_inline)) {
      if (a.np.tanh_output) d = tanhf_(d);
      const float xn = __builtin_fmaf(d, a.np.scale, io.xj);
      if (io.live && q == 0) {
        xL[io.pos] = xn;
        if (io.is_w1) xw[io.pos] = xn * io.sc;
      }
    };
      static_for<0, kTPW / 2>([&](auto kc) {
        constexpr int k = 2 * decltype(kc)::value;
        if (tile_of(wv + kMxWaves * k) < ntiles) {           // (a wave's tiles are real from slot 0 up: k real or neither)
          const TileIo ia = tile_in(k), ib = tile_in(k + 1);
          float da, db;
          bx::tile_step2_w<PRE, MxW<PRE>>(core.w, sr[k], sr[k + 1], ia.in0, ia.in1, ib.in0, ib.in1, q, da, db);
          tile_out(ia, da);
          if (ib.real) tile_out(ib, db);
        }
      });
    }
    if (PRE == L2O_PRE_FC_ELU) {                            // beta^k as a float-float running product
      float hi = p1h * a.np.beta1, er = __builtin_fmaf(p1h, a.np.beta1, -hi);
      float lo = __builtin_fmaf(p1l, a.np.beta1, er), sum = hi + lo;
      p1l = lo - (sum - hi); p1h = sum;
      hi = p2h * a.np.beta2; er = __builtin_fmaf(p2h, a.np.beta2, -hi);
      lo = __builtin_fmaf(p2l, a.np.beta2, er); sum = hi + lo;
      p2l = lo - (sum - hi); p2h = sum;
    }
    pc.mark(6);
    __syncthreads();                                        // the scaled coordinates of the next step are complete
    pc.mark(7);
  }
#ifdef L2O_PROFILE_PHASES
  if (inst == 0 && mem == 0 && tid == 0) pc.dump(a.ws->phases);
#endif
  if (inst == 0 && mem == 0 && tid == 0) a.ws->ticks = __builtin_readcyclecounter() - loop_t0;

  // ---- write back: x, moments, LSTM state -------------------------------------------------------
#pragma unroll
  for (int pi = 0; pi < kMxCoords / kThreads; ++pi) {      // (static trip count: 1 | 2)
    const int pos = tid + kThreads * pi;
    const int ti = tile_of(pos >> 4);
    if (ti < ntiles) {
      const int var = var_of(ti);
      const int jl = (ti - a.tile_begin[var]) * kTile + (pos & 15);
      if (jl < a.n[var]) {
        I.x[var][jl] = xL[pos];
        if (PRE == L2O_PRE_FC_ELU) { I.m[var][jl] = mL[pos]; I.v[var][jl] = vL[pos]; }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < kTPW; ++k) {
    const int ti = tile_of(wv + kMxWaves * k);
    if (ti < ntiles) {
      const int var = var_of(ti);
      store_tile_state(sr[k], I.st[var] + (size_t)(ti - a.tile_begin[var]) * kStateFloatsPerTile, lane);
    }
  }
  if (inst == 0 && mem == 0 && tid == 0) a.ws->ticks_total = __builtin_readcyclecounter() - kernel_t0;
}

}  // namespace l2o
