// l2o_kernels.hip -- gfx950 (MI355X / CDNA4) kernels + the C ABI of include/l2o_abi.h.
//
// Reference semantics (file:line under /root/reference/Model_Free_L2O/
// "L2O-DM and L2O-RNNProp"/, shorthand DM/):
//   unroll loop        DM/meta.py:319-389, DM/meta_rnnprop_eval.py (time_step/update)
//   optimizer nets     DM/networks.py:157-300
//   preprocess         DM/preprocess.py:26-70
//   optimizees         DM/problems.py:41-213
// Written for gfx950 only: wave64, v_mfma_f32_16x16x4_f32, 160 KiB LDS per CU.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "l2o_common.h"
#include "l2o_lstm_bx3.h"

using namespace l2o;

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";
// which kernel form the last l2o_unroll* / l2o_mlp_unroll* call of this thread launched (l2o_last_unroll_form, ABI v12):
// L2O_FORM_* | dispatches << 8.  Diagnostic only, like g_err -- no launch depends on it.
static thread_local int g_last_form = 0;
static inline void note_form(int form, int dispatches = 1) { g_last_form = form | (dispatches << 8); }

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
#define HIP_TRY(expr)                                                              \
  do {                                                                             \
    hipError_t e_ = (expr);                                                        \
    if (e_ != hipSuccess) return fail(L2O_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

static inline int tiles_per_problem(int64_t D) { return (int)((D + kTile - 1) / kTile); }

// ---------------------------------------------------------------------------
// kernel parameter blocks (plain structs passed by value)
// ---------------------------------------------------------------------------
struct NetParams {
  const float* wpack;
  float scale;
  float k_inv_ln2;   // ln2 / k   (LogAndSign: log(.)/k == log2(.) * ln2/k)
  float exp_k;       // fp32(e^k)
  float beta1, beta2, omb1, omb2;   // fp32(beta), fp32(1 - beta)
  int tanh_output;
};

struct ProbParams {
  int kind, B_local, D, M;
  int w_shared;      // W is one [M, D] matrix for every problem
  int hvp;           // Hessian-vector mode (l2o_problem_hvp): the residual is W xs (no y), no separable terms
  float inv_bg;      // 1 / B_global
  float l1, alpha;
  float twopi;       // 2*pi (rastrigin) or 2*3.1415926 (square_cos, DM/problems.py:989)
  const float* W;
  const float* y;
  const float* C;
  const float* x_scale;
};

// ---------------------------------------------------------------------------
// K1: step-granular optimizer step on a gradient panel (state in HBM).
// One wave per 16-coordinate tile, grid-stride over tiles; weights live in VGPRs.
// HBM traffic per coordinate: 320 B state read + 320 B write + g + x r/w.
// ---------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// s_waitcnt vmcnt(n) for a run-time n from the small set the step kernel needs
__device__ __forceinline__ void wait_vmcnt_le(int n) {
#define L2O_VMCASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
  switch (n) {
    L2O_VMCASE(0) L2O_VMCASE(5) L2O_VMCASE(7) L2O_VMCASE(9) L2O_VMCASE(10) L2O_VMCASE(12) L2O_VMCASE(14)
    L2O_VMCASE(15) L2O_VMCASE(17) L2O_VMCASE(18) L2O_VMCASE(19) L2O_VMCASE(22) L2O_VMCASE(23) L2O_VMCASE(24)
    L2O_VMCASE(28) L2O_VMCASE(29) L2O_VMCASE(33)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
#undef L2O_VMCASE
}

constexpr int kStepRing = 4;       // LDS ring slots per wave
constexpr int kStepAhead = 3;      // tiles in flight ahead of the one being computed

// up to kMaxStepSegs (variable, state) panels updated by one launch of the same network
constexpr int kMaxStepSegs = 8;
struct StepSegs {
  int n;
  int tile_end[kMaxStepSegs];     // running tile count (exclusive end) per segment
  int tpp[kMaxStepSegs], D[kMaxStepSegs];
  const float* g[kMaxStepSegs];
  float* m[kMaxStepSegs];
  float* v[kMaxStepSegs];
  float* st[kMaxStepSegs];
  float* x[kMaxStepSegs];
  float* st_out[kMaxStepSegs];    // where the new state / moments go (== st / m / v unless the caller records history)
  float* m_out[kMaxStepSegs];
  float* v_out[kMaxStepSegs];
};

template <int PRE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_cwlstm_step(
    NetParams np, StepSegs sg, float om1, float om2) {
  // One wave per SIMD (the bf16x3 weights take 180-240 registers), so memory latency is
  // hidden by DEPTH instead of occupancy: a wave walks its tiles with the next three tiles'
  // inputs in flight as LDS-DMA (global_load_lds: packed state 5 x dwordx4 per lane, and
  // g / x / m / v one dword each -- no staging registers), consumed after a counted
  // s_waitcnt.  VMEM operations retire in issue order, so "at most N outstanding" with
  // N = (later prefetches) x (loads per prefetch) + (later tiles' state stores) guarantees
  // that this tile has landed; the masked x / m / v stores only make the wait stricter.
  constexpr int NL = PRE == L2O_PRE_FC_ELU ? 9 : 7;              // VMEM loads per prefetched tile
  constexpr int kSlotF4 = 5 * 64 + 64;                           // float4 per slot: state + {g,x,m,v} rows
  __shared__ float4 sbuf[4][kStepRing][kSlotF4];
  __shared__ __attribute__((aligned(16))) float bias_s[bx::kBiasWords];   // the gate biases = accumulator inits
  bx::stage_bias(bias_s, np.wpack, PRE, threadIdx.x, blockDim.x);       // (ordered by the __syncthreads() of the fragment staging)
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // scalar: the waits below branch on it
  const int c = lane & 15, q = lane >> 4;
  const int wave = blockIdx.x * (blockDim.x >> 6) + wv;
  const int nwaves = gridDim.x * (blockDim.x >> 6);
  const int ntiles = sg.tile_end[sg.n - 1];
  // ---- the bf16x3 weight fragments (60 / 80 KB) reach the four waves through LDS: one DMA copy per
  // workgroup instead of four L2 reads; the staging area is the (not yet used) prefetch ring
  constexpr bool PK = bx::packed_default(PRE);
  bx::NetWB<PRE, PK> w;
  {
    constexpr int kFrags = bx::nchunks(PRE) * kNT * bx::frags(PK);   // 1 KB each = one wave-wide dwordx4
    static_assert(kFrags * 1024 <= (int)sizeof(sbuf), "staging area");
    float4* stage = &sbuf[0][0][0];
    const float* src = np.wpack + bx::frag_off(PRE, PK, 0, 0, 0) + lane * 4;
#pragma unroll
    for (int f = 0; f < (kFrags + 3) / 4; ++f) {
      const int fr = 4 * f + wv;
      if (fr < kFrags)
        __builtin_amdgcn_global_load_lds((gbl_void*)(src + fr * bx::kFragWords), (lds_void*)(stage + fr * 64), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int ch = 0; ch < bx::NetWB<PRE, PK>::NCH; ++ch)
#pragma unroll
      for (int t = 0; t < kNT; ++t)
#pragma unroll
        for (int s3 = 0; s3 < bx::frags(PK); ++s3) {
          const float4 v4 = stage[((ch * kNT + t) * bx::frags(PK) + s3) * 64 + lane];
          w.a[ch][t][s3] = __builtin_bit_cast(bx::u32x4, v4);
        }
    __syncthreads();                                               // everyone holds its copy: the ring may be used
#pragma unroll
    for (int ch = 0; ch < bx::NetWB<PRE, PK>::NCH; ++ch)           // fragments -> AGPRs (MFMA reads them there)
#pragma unroll
      for (int t = 0; t < kNT; ++t)
#pragma unroll
        for (int s3 = 0; s3 < bx::frags(PK); ++s3) asm volatile("" : "+a"(w.a[ch][t][s3]));
  }
  if (wave >= ntiles) return;
  const int ntl = (ntiles - wave + nwaves - 1) / nwaves;         // tiles of this wave
  // global tile index -> segment (wave-uniform scalar work)
  struct Loc { const float* g; float *m, *v, *st, *x, *st_out, *m_out, *v_out; int tile, tpp, D; };
  Loc L0;                                                        // the common single-panel launch: fields read once
  L0.tile = 0; L0.g = sg.g[0]; L0.m = sg.m[0]; L0.v = sg.v[0]; L0.st = sg.st[0]; L0.x = sg.x[0];
  L0.st_out = sg.st_out[0]; L0.m_out = sg.m_out[0]; L0.v_out = sg.v_out[0];
  L0.tpp = sg.tpp[0]; L0.D = sg.D[0];
  const bool single = sg.n == 1;
  auto locate = [&](int tile) {
    Loc L = L0;
    if (single) {
      L.tile = tile;
      return L;
    }
    int sidx = 0;
    while (sidx + 1 < sg.n && tile >= sg.tile_end[sidx]) ++sidx;
    L.tile = tile - (sidx ? sg.tile_end[sidx - 1] : 0);
    L.g = sg.g[sidx]; L.m = sg.m[sidx]; L.v = sg.v[sidx]; L.st = sg.st[sidx]; L.x = sg.x[sidx];
    L.st_out = sg.st_out[sidx]; L.m_out = sg.m_out[sidx]; L.v_out = sg.v_out[sidx];
    L.tpp = sg.tpp[sidx]; L.D = sg.D[sidx];
    return L;
  };
  bx::load_netw<PRE, false>(w, np.wpack, lane);                  // the small fp32 weights
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // weights landed: the VMEM queue is empty
  bx::set_bias(w, bias_s, q);
  // dead lanes (j >= D in the last tile of a problem) read the problem's last coordinate:
  // branch-free loads, masked where they are used
  auto prefetch = [&](int k, int kslot) {
    const Loc L = locate(wave + k * nwaves);
    const int tile = L.tile, tpp = L.tpp, D = L.D;
    const float *g = L.g, *x = L.x, *mbuf = L.m, *vbuf = L.v, *st = L.st;
    float4* slot = sbuf[wv][kslot % kStepRing];
    const float* src = st + (size_t)tile * kStateFloatsPerTile + lane * 4;
#pragma unroll
    for (int jj = 0; jj < 5; ++jj)
      __builtin_amdgcn_global_load_lds((gbl_void*)(src + jj * 256), (lds_void*)(slot + jj * 64), 16, 0, 0);
    const int b_ = tile / tpp, tw_ = tile - b_ * tpp;
    const size_t idx_ = (size_t)b_ * D + min(tw_ * kTile + c, D - 1);
    float* aux = reinterpret_cast<float*>(slot + 5 * 64);
    __builtin_amdgcn_global_load_lds((gbl_void*)(g + idx_), (lds_void*)(aux), 4, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_void*)(x + idx_), (lds_void*)(aux + 64), 4, 0, 0);
    if (PRE == L2O_PRE_FC_ELU) {
      __builtin_amdgcn_global_load_lds((gbl_void*)(mbuf + idx_), (lds_void*)(aux + 128), 4, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_void*)(vbuf + idx_), (lds_void*)(aux + 192), 4, 0, 0);
    }
  };
  // unconditional prologue (a wave with fewer than three tiles re-fetches its last one): a
  // static VMEM count keeps hipcc from draining the queue at the loop entry
#pragma unroll
  for (int k = 0; k < kStepAhead; ++k) prefetch(min(k, ntl - 1), k);
  for (int k = 0; k < ntl; ++k) {
    const Loc L = locate(wave + k * nwaves);
    const int tile = L.tile, tpp = L.tpp, D = L.D;
    float *x = L.x, *mbuf = L.m_out, *vbuf = L.v_out, *st = L.st_out;
    const int b = tile / tpp, tw = tile - b * tpp;
    const int j = tw * kTile + c;
    const bool live = j < D;
    const size_t idx = (size_t)b * D + j;
#ifndef L2O_ABLATE_STEP_LOAD
    wait_vmcnt_le(NL * min(kStepAhead - 1, ntl - 1 - k) + 5 * min(k, kStepAhead));
#endif
    // the slot is read with inline-asm ds_reads: hipcc cannot tell the ring slots apart and
    // would drain the whole VMEM queue (vmcnt(0)) before an ordinary LDS read of sbuf
    const unsigned slot_addr =
        (unsigned)(size_t)(__attribute__((address_space(3))) char*)(&sbuf[wv][k % kStepRing][0]) + lane * 16;
    const unsigned aux_addr = slot_addr - lane * 12 + 5 * 1024;
    TileState s;
    float gv, xv, m = 0.f, v = 0.f;
    {
      float4 v0, v1, v2, v3, v4;
      if (PRE == L2O_PRE_FC_ELU)
        asm volatile(
            "ds_read_b128 %0, %9\n\tds_read_b128 %1, %9 offset:1024\n\tds_read_b128 %2, %9 offset:2048\n\t"
            "ds_read_b128 %3, %9 offset:3072\n\tds_read_b128 %4, %9 offset:4096\n\t"
            "ds_read_b32 %5, %10\n\tds_read_b32 %6, %10 offset:256\n\tds_read_b32 %7, %10 offset:512\n\t"
            "ds_read_b32 %8, %10 offset:768\n\ts_waitcnt lgkmcnt(0)"
            : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(gv), "=&v"(xv), "=&v"(m), "=&v"(v)
            : "v"(slot_addr), "v"(aux_addr)
            : "memory");
      else
        asm volatile(
            "ds_read_b128 %0, %7\n\tds_read_b128 %1, %7 offset:1024\n\tds_read_b128 %2, %7 offset:2048\n\t"
            "ds_read_b128 %3, %7 offset:3072\n\tds_read_b128 %4, %7 offset:4096\n\t"
            "ds_read_b32 %5, %8\n\tds_read_b32 %6, %8 offset:256\n\ts_waitcnt lgkmcnt(0)"
            : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(gv), "=&v"(xv)
            : "v"(slot_addr), "v"(aux_addr)
            : "memory");
      // slot k is in registers now; slot (k+3)%4 = (k-1)%4 is free for the next prefetch
      const float e[20] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y,
                           v2.z, v2.w, v3.x, v3.y, v3.z, v3.w, v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int t = 0; t < kNT; ++t) { s.h1[t] = e[t]; s.c1[t] = e[5 + t]; s.h2[t] = e[10 + t]; s.c2[t] = e[15 + t]; }
    }
    if (!live) { gv = 0.0f; m = 0.0f; v = 0.0f; }
#ifndef L2O_ABLATE_STEP_LOAD
    if (k + kStepAhead < ntl) prefetch(k + kStepAhead, k + kStepAhead);
#endif
    float in0, in1;
    if (PRE == L2O_PRE_FC_ELU) {
      rnnprop_inputs(gv, m, v, np.beta1, np.beta2, np.omb1, np.omb2, om1, om2, in0, in1);
      if (!live) { in0 = 0.0f; in1 = 0.0f; }
      if (live && q == 0) { mbuf[idx] = m; vbuf[idx] = v; }
    } else {
      preprocess_grad<PRE>(gv, np.k_inv_ln2, np.exp_k, in0, in1);
    }
    float d = bx::tile_step<PRE>(w, s, in0, in1, q);
    if (np.tanh_output) d = tanhf_(d);
    d *= np.scale;
    if (live && q == 0) x[idx] = xv + d;
#ifndef L2O_ABLATE_STEP_STORE
    store_tile_state(s, st + (size_t)tile * kStateFloatsPerTile, lane);
#else
    if (d == 1234.5f) store_tile_state(s, st + (size_t)tile * kStateFloatsPerTile, lane);
#endif
  }
}

// layers == (): StandardDeepLSTM with no cores = Linear on the preprocessed gradient
// (DM/networks.py:225-232 with an empty DeepRNN; used by the meta_test.py:50-69 KAT).
// lw = {w0, w1, b}
template <int PRE>
__global__ void k_linear_step(NetParams np, const float* __restrict__ g, float* __restrict__ x, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float w0 = np.wpack[0], w1 = np.wpack[1], bl = np.wpack[2];
  float in0, in1;
  preprocess_grad<PRE>(g[i], np.k_inv_ln2, np.exp_k, in0, in1);
  float d = __builtin_fmaf(in1, w1, in0 * w0) + bl;
  if (np.tanh_output) d = tanhf_(d);
  x[i] += d * np.scale;
}

// ---------------------------------------------------------------------------
// K2: optimizee forward + gradient for arbitrary sizes (matrix streamed from HBM/L2).
// One 256-thread workgroup per problem.
//   pass 1  r = W xs - y      rows over waves, columns over lanes (coalesced), wave reduce
//   pass 2  g = W^T r         columns over threads (coalesced), rows split over thread groups
// ---------------------------------------------------------------------------
constexpr int kFgThreads = 512;
constexpr int kFgWaves = kFgThreads / 64;

__device__ __forceinline__ float block_sum_fg(float v, float* red /* >= kFgWaves floats */) {
  v = wave_sum64(v);
  const int wv = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[wv] = v;
  __syncthreads();
  float s = red[0];
#pragma unroll
  for (int k = 1; k < kFgWaves; ++k) s += red[k];
  return s;
}

// VEC = true: D % 4 == 0 and 16-byte aligned rows -> dwordx4 loads, 4 rows (r pass) /
// 4 row-strided loads (g pass) in flight per lane; VEC = false: scalar fallback.
template <bool VEC>
__global__ __launch_bounds__(kFgThreads) void k_problem_fg(ProbParams pp, const float* __restrict__ x,
                                                            float* __restrict__ f_part,
                                                            float* __restrict__ gout) {
  extern __shared__ float sm[];
  const int D = pp.D, M = pp.M;
  float* xs = sm;                       // [D4]  scaled x (padded to a multiple of 4)
  const int D4 = (D + 3) & ~3;
  float* rs = xs + D4;                  // [M]   residual
  float* part = rs + ((M + 3) & ~3);    // [4 * kFgThreads] partial column sums
  float* red = part + 4 * kFgThreads;   // [kFgWaves]
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* xb = x + (size_t)b * D;
  const float* sb = pp.x_scale ? pp.x_scale + (size_t)b * D : nullptr;

  float facc = 0.0f;   // per-thread contribution to f_b
  for (int j = tid; j < D4; j += kFgThreads) {
    float xv = 0.0f;
    if (j < D) {
      xv = xb[j] * (sb ? sb[j] : 1.0f);
      if (pp.kind == L2O_PROB_SIMPLE) facc += xv * xv;
      if (pp.kind == L2O_PROB_LASSO) facc += pp.l1 * __builtin_fabsf(xv);
      if (pp.kind == L2O_PROB_RASTRIGIN || pp.kind == L2O_PROB_SQUARE_COS)
        facc += pp.alpha - pp.alpha * pp.C[(size_t)b * D + j] * l2o::cos_f(pp.twopi * xv);
    }
    xs[j] = xv;
  }
  __syncthreads();

  if (pp.kind != L2O_PROB_SIMPLE) {
    const float* Wb = pp.W + (pp.w_shared ? (size_t)0 : (size_t)b * M * D);
    const float* yb = pp.y + (size_t)b * M;
    const float coef = (pp.kind == L2O_PROB_QUADRATIC || pp.kind == L2O_PROB_SQUARE_COS) ? 1.0f : 0.5f;
    if (VEC) {
      // ---- pass 1: r = W xs - y ; each wave takes 4 rows at a time -> 4 x (D/256) dwordx4 in flight
      for (int i0 = wv * 4; i0 < M; i0 += kFgWaves * 4) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j = lane * 4; j < D; j += 256) {
          const float4 xv4 = *reinterpret_cast<const float4*>(xs + j);
          float4 w4[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int i = i0 + k < M ? i0 + k : M - 1;       // clamp: result of a clamped row is discarded
            w4[k] = *reinterpret_cast<const float4*>(Wb + (size_t)i * D + j);
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            acc[k] = __builtin_fmaf(w4[k].x, xv4.x, acc[k]);
            acc[k] = __builtin_fmaf(w4[k].y, xv4.y, acc[k]);
            acc[k] = __builtin_fmaf(w4[k].z, xv4.z, acc[k]);
            acc[k] = __builtin_fmaf(w4[k].w, xv4.w, acc[k]);
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float a = wave_sum64(acc[k]);
          if (lane == 0 && i0 + k < M) {
            const float r = a - (pp.hvp ? 0.0f : yb[i0 + k]);
            rs[i0 + k] = r;
            facc += coef * r * r;
          }
        }
      }
    } else {
      for (int i = wv; i < M; i += kFgWaves) {
        const float* row = Wb + (size_t)i * D;
        float acc = 0.0f;
        for (int j = lane; j < D; j += 64) acc = __builtin_fmaf(row[j], xs[j], acc);
        acc = wave_sum64(acc);
        if (lane == 0) {
          const float r = acc - (pp.hvp ? 0.0f : yb[i]);
          rs[i] = r;
          facc += coef * r * r;
        }
      }
    }
  }
  const float fb = block_sum_fg(facc, red);   // contains a __syncthreads: rs is visible after it
  if (tid == 0) f_part[b] = fb;
  if (gout == nullptr) return;

  float* gb = gout + (size_t)b * D;
  if (pp.kind == L2O_PROB_SIMPLE) {
    for (int j = tid; j < D; j += kFgThreads) gb[j] = 2.0f * xs[j] * pp.inv_bg * (sb ? sb[j] : 1.0f);
    return;
  }
  const float* Wb = pp.W + (pp.w_shared ? (size_t)0 : (size_t)b * M * D);
  const float cg = (pp.kind == L2O_PROB_QUADRATIC || pp.kind == L2O_PROB_SQUARE_COS) ? 2.0f : 1.0f;
  auto finish = [&](int j, float s) {
    float gj = cg * s;
    const float xv = xs[j];
    if (!pp.hvp) {
      if (pp.kind == L2O_PROB_LASSO) gj += pp.l1 * (xv > 0.f ? 1.f : (xv < 0.f ? -1.f : 0.f));
      if (pp.kind == L2O_PROB_RASTRIGIN || pp.kind == L2O_PROB_SQUARE_COS)
        gj += pp.twopi * pp.alpha * pp.C[(size_t)b * D + j] * l2o::sin_f(pp.twopi * xv);
    }
    gb[j] = gj * pp.inv_bg * (sb ? sb[j] : 1.0f);
  };
  if (VEC) {
    // ---- pass 2: g = W^T r ; a thread owns 4 adjacent columns, row groups are strided over
    // RP thread groups; 4 dwordx4 loads in flight per thread
    const int ncol4 = D >> 2;
    const int CP4 = ncol4 >= kFgThreads ? kFgThreads : ((ncol4 + 63) & ~63);
    const int RP = kFgThreads / CP4;
    const int jc = tid % CP4, rp = tid / CP4;
    for (int j0 = 0; j0 < ncol4; j0 += CP4) {
      const int j4 = j0 + jc;
      float4 a = {0.f, 0.f, 0.f, 0.f};
      if (j4 < ncol4 && rp < RP) {
        const float* col = Wb + 4 * (size_t)j4;
        int i = rp;
        for (; i + 3 * RP < M; i += 4 * RP) {
          const float4 w0 = *reinterpret_cast<const float4*>(col + (size_t)i * D);
          const float4 w1 = *reinterpret_cast<const float4*>(col + (size_t)(i + RP) * D);
          const float4 w2 = *reinterpret_cast<const float4*>(col + (size_t)(i + 2 * RP) * D);
          const float4 w3 = *reinterpret_cast<const float4*>(col + (size_t)(i + 3 * RP) * D);
          const float r0 = rs[i], r1 = rs[i + RP], r2 = rs[i + 2 * RP], r3 = rs[i + 3 * RP];
          a.x += (w0.x * r0 + w1.x * r1) + (w2.x * r2 + w3.x * r3);
          a.y += (w0.y * r0 + w1.y * r1) + (w2.y * r2 + w3.y * r3);
          a.z += (w0.z * r0 + w1.z * r1) + (w2.z * r2 + w3.z * r3);
          a.w += (w0.w * r0 + w1.w * r1) + (w2.w * r2 + w3.w * r3);
        }
        for (; i < M; i += RP) {
          const float4 w0 = *reinterpret_cast<const float4*>(col + (size_t)i * D);
          const float r0 = rs[i];
          a.x = __builtin_fmaf(w0.x, r0, a.x); a.y = __builtin_fmaf(w0.y, r0, a.y);
          a.z = __builtin_fmaf(w0.z, r0, a.z); a.w = __builtin_fmaf(w0.w, r0, a.w);
        }
      }
      __syncthreads();
      *reinterpret_cast<float4*>(part + 4 * tid) = a;
      __syncthreads();
      if (rp == 0 && j4 < ncol4) {
        float4 s4 = *reinterpret_cast<const float4*>(part + 4 * jc);
        for (int p = 1; p < RP; ++p) {
          const float4 t4 = *reinterpret_cast<const float4*>(part + 4 * (p * CP4 + jc));
          s4.x += t4.x; s4.y += t4.y; s4.z += t4.z; s4.w += t4.w;
        }
        finish(4 * j4, s4.x); finish(4 * j4 + 1, s4.y); finish(4 * j4 + 2, s4.z); finish(4 * j4 + 3, s4.w);
      }
    }
  } else {
    const int CP = D >= kFgThreads ? kFgThreads : ((D + 63) & ~63);
    const int RP = kFgThreads / CP;
    const int jc = tid % CP, rp = tid / CP;
    for (int j0 = 0; j0 < D; j0 += CP) {
      const int j = j0 + jc;
      float a0 = 0.f, a1 = 0.f;
      if (j < D && rp < RP) {
        int i = rp;
        for (; i + RP < M; i += 2 * RP) {
          a0 = __builtin_fmaf(Wb[(size_t)i * D + j], rs[i], a0);
          a1 = __builtin_fmaf(Wb[(size_t)(i + RP) * D + j], rs[i + RP], a1);
        }
        for (; i < M; i += RP) a0 = __builtin_fmaf(Wb[(size_t)i * D + j], rs[i], a0);
      }
      __syncthreads();
      part[tid] = a0 + a1;
      __syncthreads();
      if (rp == 0 && j < D) {
        float s = part[jc];
        for (int p = 1; p < RP; ++p) s += part[p * CP + jc];
        finish(j, s);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// K2b: the same forward + gradient with the matrix read ONCE.  A wave keeps the rows it visits in
// registers between their two uses: lane l holds the row's columns 4 (64 v + l) .. + 3 (v < NV: coalesced
// 1 KB pieces), r_i = dot(row, xs) - y_i is a wave sum that every lane receives, and the same registers
// then feed g += r_i * row.  Eight waves walk the rows in groups of four (4 NV dwordx4 loads in flight per
// lane); their partial gradients are added in a fixed order through LDS.  Halves the HBM traffic of the
// step-granular path's matrix-bound optimizees (config 3: 256 x 512 per problem, 128 MiB per batch).
// D % 4 == 0, D <= 256 NV.
// ---------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(kFgThreads) void k_problem_fg1(ProbParams pp, const float* __restrict__ x,
                                                             float* __restrict__ f_part,
                                                             float* __restrict__ gout) {
  extern __shared__ float sm[];
  const int D = pp.D, M = pp.M;
  float* part = sm;                     // [kFgWaves][D] partial gradients
  float* red = part + kFgWaves * D;     // [kFgWaves]
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* xb = x + (size_t)b * D;
  const float* sb = pp.x_scale ? pp.x_scale + (size_t)b * D : nullptr;
  const float* Wb = pp.W + (pp.w_shared ? (size_t)0 : (size_t)b * M * D);
  const l2o_cfp yb = (l2o_cfp)(pp.y + (size_t)b * M);
  const bool want_g = gout != nullptr;
  const float coef = (pp.kind == L2O_PROB_QUADRATIC || pp.kind == L2O_PROB_SQUARE_COS) ? 1.0f : 0.5f;

  float4 xv[NV], sv[NV], ga[NV];
  float facc = 0.0f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int j = 4 * (64 * v + lane);
    const float4 zero = {0.f, 0.f, 0.f, 0.f}, one = {1.f, 1.f, 1.f, 1.f};
    xv[v] = zero; sv[v] = one; ga[v] = zero;
    if (j < D) {
      xv[v] = *reinterpret_cast<const float4*>(xb + j);
      if (sb) sv[v] = *reinterpret_cast<const float4*>(sb + j);
      xv[v].x *= sv[v].x; xv[v].y *= sv[v].y; xv[v].z *= sv[v].z; xv[v].w *= sv[v].w;
      if (wv == 0) {                                        // the separable terms of f: once per problem
        const float xe[4] = {xv[v].x, xv[v].y, xv[v].z, xv[v].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (pp.kind == L2O_PROB_LASSO) facc += pp.l1 * __builtin_fabsf(xe[e]);
          if (pp.kind == L2O_PROB_RASTRIGIN || pp.kind == L2O_PROB_SQUARE_COS)
            facc += pp.alpha - pp.alpha * pp.C[(size_t)b * D + j + e] * l2o::cos_f(pp.twopi * xe[e]);
        }
      }
    }
  }
  // software pipeline: the next group of four rows is requested before the current one is reduced (one wave
  // has 4 NV dwordx4 per lane in flight while it computes; without it every group pays a full memory latency)
  auto load4 = [&](int i0, float4 (&w4)[4][NV]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = i0 + k < M ? i0 + k : M - 1;             // clamped rows contribute r = 0 below
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int j = 4 * (64 * v + lane);
        const float4 zero = {0.f, 0.f, 0.f, 0.f};
        w4[k][v] = j < D ? *reinterpret_cast<const float4*>(Wb + (size_t)i * D + j) : zero;
      }
    }
  };
  auto use4 = [&](int i0, const float4 (&w4)[4][NV]) {
    float acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float a = 0.0f;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        a = __builtin_fmaf(w4[k][v].x, xv[v].x, a);
        a = __builtin_fmaf(w4[k][v].y, xv[v].y, a);
        a = __builtin_fmaf(w4[k][v].z, xv[v].z, a);
        a = __builtin_fmaf(w4[k][v].w, xv[v].w, a);
      }
      acc[k] = a;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool ok = i0 + k < M;
      const float r = ok ? wave_sum64(acc[k]) - (pp.hvp ? 0.0f : yb[ok ? i0 + k : 0]) : 0.0f;
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
  {
    constexpr int STEP = kFgWaves * 4;
    float4 wa[4][NV], wb[4][NV];
    int i0 = wv * 4;
    if (i0 < M) load4(i0, wa);
    for (; i0 < M; i0 += 2 * STEP) {
      if (i0 + STEP < M) load4(i0 + STEP, wb);
      use4(i0, wa);
      if (i0 + STEP < M) {
        if (i0 + 2 * STEP < M) load4(i0 + 2 * STEP, wa);
        use4(i0 + STEP, wb);
      }
    }
  }
  if (want_g) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int j = 4 * (64 * v + lane);
      if (j < D) *reinterpret_cast<float4*>(part + wv * D + j) = ga[v];
    }
  }
  const float fb = block_sum_fg(facc, red);   // contains __syncthreads: the partial gradients are visible after it
  if (tid == 0) f_part[b] = fb;
  if (!want_g) return;
  float* gb = gout + (size_t)b * D;
  const float cg = (pp.kind == L2O_PROB_QUADRATIC || pp.kind == L2O_PROB_SQUARE_COS) ? 2.0f : 1.0f;
  for (int j = tid; j < D; j += kFgThreads) {
    float s = part[j];
#pragma unroll
    for (int w = 1; w < kFgWaves; ++w) s += part[w * D + j];
    const float sc = sb ? sb[j] : 1.0f;
    const float xj = xb[j] * sc;
    float gj = cg * s;
    if (!pp.hvp) {
      if (pp.kind == L2O_PROB_LASSO) gj += pp.l1 * (xj > 0.f ? 1.f : (xj < 0.f ? -1.f : 0.f));
      if (pp.kind == L2O_PROB_RASTRIGIN || pp.kind == L2O_PROB_SQUARE_COS)
        gj += pp.twopi * pp.alpha * pp.C[(size_t)b * D + j] * l2o::sin_f(pp.twopi * xj);
    }
    gb[j] = gj * pp.inv_bg * sc;
  }
}

// ---------------------------------------------------------------------------
// K3: the fused persistent unroll.  One workgroup per problem, one wave per
// 16-coordinate tile (<= 8 waves), T steps in ONE launch.
//   LDS  : the problem matrix TWICE -- W_b row-major and W_b^T row-major, row stride
//          S = SQ + 16 floats (S/4 == 4 or 12 mod 16 -> ds_read_b128 of a (row, 16-byte
//          column chunk) per lane is bank-conflict free) -- plus y, x*s and the residual r
//   VGPR : optimizer weights (MFMA A fragments), LSTM state, x, m, v
//   HBM  : W_b read once per launch; x/state/m/v read+written once; one float per
//          (step, problem)
// Per step and wave (lane l also plays the GEMV role (row = 16*wave + l/4, chunk = l%4)):
//   xs <- x*s ; barrier
//   r  = W xs - y           || the 25 layer-2 MFMAs fed by the previous h2
//   f_b partial ; barrier
//   g  = W^T r (own 16 coordinates) || the 25 layer-1 MFMAs fed by the previous h1
//   preprocess(g) -> input MFMAs -> gates -> layer-2 MFMAs -> gates -> Linear -> x += delta
// i.e. 50 of the 80 MFMAs of a step depend only on the previous step's state and are
// issued interleaved with the GEMV's LDS reads and FMAs (separate pipes).
// ---------------------------------------------------------------------------
struct UnrollArgs {
  NetParams np;
  ProbParams pp;
  float* x;
  float* st;
  float* m;
  float* v;
  float* fx_part;
  int T;
  float p1_hi, p1_lo, p2_hi, p2_lo;   // beta^step0 as float-float
  // optional per-step history for the meta-gradient (l2o_unroll_record): packed state BEFORE
  // step t, the gradient fed to the network at step t, RNNProp moments AFTER step t, and the
  // gradient at x_T.  All NULL for a plain unroll.
  float *hist_st, *hist_g, *hist_m, *hist_v, *hist_gfinal;
  // restart (l2o_unroll_reduce): read the iterate from x_in (x receives x_T) / start from the zero LSTM state and
  // zero moments instead of reading st, m, v -- `reset` + the first unroll in one launch, no memset / copy pass
  const float* x_in;
  int zero_state;
  long long* ticks;   // where k_unroll_lds leaves the step-loop cycle count of problem 0 (PairWs::ticks), or NULL
};

#ifdef L2O_ABLATE_BARRIER
#define L2O_SYNC() ((void)0)
#else
#define L2O_SYNC() __syncthreads()
#endif
// a.b accumulated into FOUR independent partial sums (x, y, z, w lanes of `acc`): the
// chunk loop then carries 4 short dependency chains instead of one long one
__device__ __forceinline__ void dot4(const float4 a, const float4 b, float4& acc) {
#ifdef L2O_ABLATE_GEMV
  acc.x += a.x;
  return;
#endif
  acc.x = __builtin_fmaf(a.x, b.x, acc.x);
  acc.y = __builtin_fmaf(a.y, b.y, acc.y);
  acc.z = __builtin_fmaf(a.z, b.z, acc.z);
  acc.w = __builtin_fmaf(a.w, b.w, acc.w);
}
__device__ __forceinline__ float hsum4(const float4 a) { return (a.x + a.y) + (a.z + a.w); }

template <int PRE, int KIND, int CH, bool HIST, bool EXACT = false>
__global__ __launch_bounds__(CH <= 4 ? 256 : 512) void k_unroll(UnrollArgs a) {
  // CH <= 4 <=> at most 4 waves per workgroup = one wave per SIMD: the bf16x3 gate GEMM with
  // its weights in VGPR + AGPR; CH == 8 (5..8 waves) keeps the fp32 MFMA form (as does EXACT: L2O_OPT_EXACT_GATES)
  using Core = LstmCore<PRE, (CH <= 4) && !EXACT>;
  constexpr int SQ = 16 * CH;      // padded square size of the LDS-resident matrix
  constexpr int S = SQ + 16;       // row stride (floats)
  extern __shared__ float sm[];
  const ProbParams& pp = a.pp;
  const int D = pp.D, M = pp.M;
  float* Ws = sm;                  // [SQ][S]  W   (rows i, columns j)
  float* WTs = Ws + SQ * S;        // [SQ][S]  W^T (rows j, columns i)
  float* xs = WTs + SQ * S;        // [SQ]
  float* rs = xs + SQ;             // [SQ]
  float* ys = rs + SQ;             // [SQ]
  float* fpart = ys + SQ;          // [8]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6;
  const int c = lane & 15, q = lane >> 4;      // LSTM role: coordinate c, unit group q
  const int grow = wv * kTile + (lane >> 2);   // GEMV role: matrix row (r pass) / coordinate (g pass)
  const int gq = lane & 3;                     //            16-byte column chunk inside a 64-byte group
  const int b = blockIdx.x;

  // ---- stage the problem into LDS (zero padded), both orientations ---------
  const float* Wb = pp.W + (pp.w_shared ? (size_t)0 : (size_t)b * M * D);
  for (int i = tid; i < 2 * SQ * S + 3 * SQ; i += blockDim.x) sm[i] = 0.0f;
  __syncthreads();
  for (int e = tid; e < M * D; e += blockDim.x) {
    const int i = e / D, j = e - i * D;
    const float v = Wb[e];
    Ws[i * S + j] = v;
    WTs[j * S + i] = v;
  }
  for (int i = tid; i < M; i += blockDim.x) ys[i] = pp.y[(size_t)b * M + i];

  // ---- per-lane persistent registers -------------------------------------
  Core core;
  core.load(a.np.wpack, lane);
  core.pin();   // (bf16x3 forms: fragments -> AGPRs; without it 125-133 v_accvgpr_read per step)
  __shared__ __attribute__((aligned(16))) float bias_s[Core::kBiasFloats];   // the gate biases = accumulator inits
  core.stage_bias(bias_s, a.np.wpack, tid, blockDim.x, q);
  const int j = wv * kTile + c;             // this lane's coordinate (LSTM role)
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
  float mv = 0.0f, vv = 0.0f;
  if (PRE == L2O_PRE_FC_ELU && !a.zero_state) { mv = live ? a.m[idx] : 0.0f; vv = live ? a.v[idx] : 0.0f; }
  float p1h = a.p1_hi, p1l = a.p1_lo, p2h = a.p2_hi, p2l = a.p2_lo;
  constexpr bool kSq = KIND == L2O_PROB_QUADRATIC || KIND == L2O_PROB_SQUARE_COS;
  const float coef = kSq ? 1.0f : 0.5f;
  const float cg = (KIND == L2O_PROB_QUADRATIC ? 2.0f : 1.0f) * pp.inv_bg;   // x2 folded in (exact)
  const float kTwoPi = pp.twopi;
  const float* wrow = Ws + grow * S + 4 * gq;
  const float* wtrow = WTs + grow * S + 4 * gq;
  const float* xsq = xs + 4 * gq;
  const float* rsq = rs + 4 * gq;
  const int perm_src = (4 * c) << 2;        // byte index of lane 4*c for ds_bpermute

  // software pipeline of the matrix work: acc1 always holds bias + (h1 of the previous step)
  // part of layer 1; it is produced at the END of the previous step (overlapping that step's
  // layer-2 gate math), here for step 0.
  f32x4 acc1[kNT], acc2[kNT];
  core.init(s, q);
  __syncthreads();                                           // the bias table is staged
  core.preload(acc1, acc2);
  core.template issue_l1_prev<0, Core::kTotal>(s, acc1);

  const size_t hist_n = (size_t)pp.B_local * D;
  for (int t = 0;; ++t) {
    const float xsv = xv * sc;
    if (live && q == 0) xs[j] = xsv;
    if (HIST && t < a.T)
      store_tile_state(s, a.hist_st + ((size_t)t * pp.B_local * nw + tile) * kStateFloatsPerTile, lane);
    L2O_SYNC();                                             // B1: xs complete
    // ---- r = W xs - y  ||  first 12 layer-2 MFMAs of the previous h2 -----------
    float4 racc = {0.f, 0.f, 0.f, 0.f};
    static_for<0, CH>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      const float4 wv4 = *reinterpret_cast<const float4*>(wrow + 16 * m);
      const float4 xv4 = *reinterpret_cast<const float4*>(xsq + 16 * m);
      core.template issue_l2_prev<(Core::kHalf * m) / CH, (Core::kHalf * (m + 1)) / CH>(s, acc2);
      dot4(wv4, xv4, racc);
    });
    const float r = quad_sum(hsum4(racc)) - ys[grow];
    float contrib = 0.0f;
    if (gq == 0) {
      rs[grow] = r;                       // rows >= M: W row and y are zero -> r == 0
      contrib = coef * r * r;
    }
    if (live && q == 0) {
      if (KIND == L2O_PROB_LASSO) contrib += pp.l1 * __builtin_fabsf(xsv);
      if (kCos) contrib += pp.alpha - pp.alpha * cj * l2o::cos_f(kTwoPi * xsv);
    }
    contrib = wave_sum64(contrib);
    if (lane == 0) fpart[wv] = contrib;
    L2O_SYNC();                                             // B2: rs, fpart complete
    if (tid == 0) {
      float f = fpart[0];
      for (int k = 1; k < nw; ++k) f += fpart[k];
      a.fx_part[(size_t)t * pp.B_local + b] = f;
    }
    if (t == a.T && !HIST) break;

    // ---- g = W^T r for this wave's 16 coordinates  ||  the other 13 of those MFMAs ----
    float4 gacc4 = {0.f, 0.f, 0.f, 0.f};
    static_for<0, CH>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      const float4 wt4 = *reinterpret_cast<const float4*>(wtrow + 16 * m);
      const float4 rv4 = *reinterpret_cast<const float4*>(rsq + 16 * m);
      core.template issue_l2_prev<Core::kHalf + ((Core::kTotal - Core::kHalf) * m) / CH,
                                  Core::kHalf + ((Core::kTotal - Core::kHalf) * (m + 1)) / CH>(s, acc2);
      dot4(wt4, rv4, gacc4);
    });
    const float gacc = quad_sum(hsum4(gacc4));                 // lanes 4k..4k+3 hold g of coordinate 16*wv + k
    float gv = __int_as_float(__builtin_amdgcn_ds_bpermute(perm_src, __float_as_int(gacc)));
    if (KIND == L2O_PROB_SQUARE_COS) gv *= 2.0f;            // only the ||wx-y||^2 part carries the 2
    if (KIND == L2O_PROB_LASSO) gv += pp.l1 * (xsv > 0.f ? 1.f : (xsv < 0.f ? -1.f : 0.f));
    if (kCos) gv += kTwoPi * pp.alpha * cj * l2o::sin_f(kTwoPi * xsv);
    gv = live ? gv * cg * sc : 0.0f;
    if (HIST && live && q == 0) {
      if (t < a.T) a.hist_g[(size_t)t * hist_n + idx] = gv;
      else a.hist_gfinal[idx] = gv;
    }
    if (HIST && t == a.T) break;                            // (history mode: the gradient at x_T was still needed)

    // ---- optimizer network ----------------------------------------------------
    float in0, in1;
    if (PRE == L2O_PRE_FC_ELU) {
      // beta^k as a float-float running product (k = step0 + t)
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
    PhaseClock pc;
    float d = core.template finish<true>(s, acc1, acc2, in0, in1, q, pc);
    if (a.np.tanh_output) d = tanhf_(d);
    xv = __builtin_fmaf(d, a.np.scale, xv);
  }

  if (live && q == 0) {
    a.x[idx] = xv;
    if (PRE == L2O_PRE_FC_ELU) { a.m[idx] = mv; a.v[idx] = vv; }
  }
  store_tile_state(s, st_tile, lane);
}

#include "l2o_unroll_pair.h"
#include "l2o_unroll_lds.h"

#include "l2o_unroll_cu.h"
#include "l2o_unroll_cu8.h"

#include "l2o_ilp_kernels.h"   // the plain two-CU / eight-wave unrolls: code in l2o_kernels_ilp.hip, `extern template` here
#ifndef L2O_TU_ILP             // (everything below belongs to the main translation unit only)

#include "l2o_mlp.h"

#include "l2o_mlp_unroll.h"
#include "l2o_mlp_xcd.h"
#include "l2o_mlp_deep.h"

#include "l2o_generic.h"

#include "l2o_atb.h"

#include "l2o_bwd.h"

#include "l2o_bwd_mfma.h"
#include "l2o_vecops.h"

// ---------------------------------------------------------------------------
// small utility kernels
// ---------------------------------------------------------------------------
// reference layout [B*D, 20] x 4  <->  packed tile layout
__global__ void k_state_repack(float* __restrict__ h1, float* __restrict__ c1, float* __restrict__ h2,
                               float* __restrict__ c2, float* __restrict__ st, int B, int D, int tpp,
                               int to_packed) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)B * tpp * 64 * 20;
  if (gid >= total) return;
  const int e = gid % 20;                 // element of the lane: 4*jv + wv
  const int lane = (gid / 20) % 64;
  const size_t tile = gid / (20 * 64);
  const int b = tile / tpp, tw = tile % tpp;
  const int c = lane & 15, q = lane >> 4;
  const int j = tw * kTile + c;
  const int arr = e / 5, t = e % 5, u = 4 * t + q;
  const size_t paddr = tile * kStateFloatsPerTile + ((size_t)(e >> 2) * 64 + lane) * 4 + (e & 3);
  float* ref = arr == 0 ? h1 : (arr == 1 ? c1 : (arr == 2 ? h2 : c2));
  if (j < D) {
    const size_t raddr = ((size_t)b * D + j) * kH + u;
    if (to_packed) st[paddr] = ref[raddr];
    else ref[raddr] = st[paddr];
  } else if (to_packed) {
    st[paddr] = 0.0f;
  }
}

// fx[t] = (sum_b fx_part[t][b]) / B_global, fixed order: each wave sums a strided slice
// sequentially, then a fixed butterfly.
__global__ void k_reduce_fx(const float* __restrict__ fx_part, int T1, int B_local, float inv_bg,
                            float* __restrict__ fx) {
  const int t = blockIdx.x;
  const int lane = threadIdx.x;   // 64 threads
  float acc = 0.0f;
  for (int b = lane; b < B_local; b += 64) acc += fx_part[(size_t)t * B_local + b];
  acc = wave_sum64(acc);
  if (lane == 0) fx[t] = acc * inv_bg;
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static bool net_ok_for_mfma(const l2o_net_cfg* c) {
  if (c->n_layers != 2 || c->hidden != kH) return false;
  if (c->kind == L2O_NET_CW) return c->preprocess == L2O_PRE_IDENTITY || c->preprocess == L2O_PRE_LOGSIGN;
  if (c->kind == L2O_NET_RNNPROP) return c->preprocess == L2O_PRE_FC_ELU;
  return false;
}

static NetParams make_net_params(const l2o_net_cfg* c, const float* wpack) {
  NetParams np;
  np.wpack = wpack;
  np.scale = (float)c->scale;
  np.k_inv_ln2 = c->logsign_k != 0.0 ? (float)(0.6931471805599453 / c->logsign_k) : 0.0f;
  np.exp_k = (float)std::exp(c->logsign_k);
  np.beta1 = (float)c->beta1;
  np.beta2 = (float)c->beta2;
  np.omb1 = (float)(1.0 - c->beta1);
  np.omb2 = (float)(1.0 - c->beta2);
  np.tanh_output = c->tanh_output;
  return np;
}

static ProbParams make_prob_params(const l2o_problem* p) {
  ProbParams pp;
  pp.kind = p->kind;
  pp.B_local = p->B_local;
  pp.D = p->D;
  pp.M = p->kind == L2O_PROB_SIMPLE ? 0 : p->M;
  pp.w_shared = (p->flags & L2O_PROB_W_SHARED) ? 1 : 0;
  pp.hvp = 0;
  pp.inv_bg = 1.0f / (float)p->B_global;
  pp.l1 = (float)p->l1;
  pp.alpha = p->kind == L2O_PROB_SQUARE_COS ? 10.0f : (float)p->alpha;
  pp.twopi = p->kind == L2O_PROB_SQUARE_COS ? (float)(2 * 3.1415926) : 6.2831853071795864769f;
  pp.W = p->W;
  pp.y = p->y;
  pp.C = p->C;
  pp.x_scale = p->x_scale;
  return pp;
}

static int check_problem(const l2o_problem* p) {
  if (!p) return fail(L2O_ERR_ARG, "problem is NULL");
  if (p->B_local <= 0 || p->B_global < p->B_local || p->D <= 0)
    return fail(L2O_ERR_ARG, "bad problem sizes B_local=%d B_global=%d D=%d", p->B_local, p->B_global, p->D);
  switch (p->kind) {
    case L2O_PROB_SIMPLE: return L2O_OK;
    case L2O_PROB_QUADRATIC:
    case L2O_PROB_LASSO:
      if (!p->W || !p->y || p->M <= 0) return fail(L2O_ERR_ARG, "problem needs W, y and M > 0");
      if (p->kind == L2O_PROB_QUADRATIC && p->M != p->D) return fail(L2O_ERR_ARG, "quadratic needs M == D");
      return L2O_OK;
    case L2O_PROB_RASTRIGIN:
    case L2O_PROB_SQUARE_COS:
      if (!p->W || !p->y || !p->C || p->M != p->D)
        return fail(L2O_ERR_ARG, "rastrigin / square_cos need W, y, C and M == D");
      return L2O_OK;
    default: return fail(L2O_ERR_UNSUPPORTED, "problem kind %d has no HIP kernel", p->kind);
  }
}

// float-float split of base^k computed in double
static void pow_ff(double base, int k, float* hi, float* lo) {
  const double p = std::pow((double)(float)base, (double)k);
  *hi = (float)p;
  *lo = (float)(p - (double)*hi);
}

struct UnrollGeom { int CH, nw; size_t lds; };
static bool unroll_geom(const l2o_problem* p, UnrollGeom* g) {
  const int D = p->D, M = p->M;
  g->nw = tiles_per_problem(D);
  if (g->nw > 8 || M > 16 * g->nw) return false;     // rows are covered by the nw waves, 16 each
  const int need = g->nw;                            // 16-float chunks per matrix row
  g->CH = need <= 1 ? 1 : (need <= 2 ? 2 : (need <= 4 ? 4 : 8));
  const int SQ = 16 * g->CH, S = SQ + 16;
  g->lds = sizeof(float) * (2 * (size_t)SQ * S + 3 * (size_t)SQ + 8);
  return g->lds <= 160 * 1024;
}

// ---------------------------------------------------------------------------
// options: A/B switches between kernels that compute the same thing.  The library keeps NO option state: every call
// carries its switches in the caller's own structs -- l2o_net_cfg.options (L2O_OPTW(option, value) words OR-ed
// together; 0 = every default), l2o_problem.flags (L2O_PROB_FG_TWO_PASS) and l2o_mlp.flags (L2O_MLP_GENERIC).  An entry
// point copies the word into a call-scoped thread-local (OptScope) that the launch helpers below read through opt().
// ---------------------------------------------------------------------------
static const int64_t kOptDefault[L2O_OPT_COUNT_] = {
    /* L2O_OPT_PAIR */ 1, /* L2O_OPT_PAIR_PLAIN_STORES */ 1, /* L2O_OPT_UNROLL_CU */ 1,
    /* L2O_OPT_FG_TWO_PASS */ 0, /* L2O_OPT_MLP_GENERIC */ 0, /* L2O_OPT_BWD_BLOCKS */ 0,
    /* L2O_OPT_BWD_KERNEL */ 0, /* L2O_OPT_MLP_UNROLL */ 1, /* L2O_OPT_MLP_XCD_WAVES */ 0, /* L2O_OPT_EXACT_GATES */ 0,
    /* L2O_OPT_WPACK_NO_CLEAR */ 0, /* L2O_OPT_MLP_HIER */ 1, /* L2O_OPT_ONE_LDS */ 1};
static thread_local uint64_t t_optw = 0;
struct OptScope {
  uint64_t saved;
  explicit OptScope(uint64_t w) : saved(t_optw) { t_optw = w; }
  ~OptScope() { t_optw = saved; }
};
static inline uint64_t cfg_optw(const l2o_net_cfg* cfg) { return cfg ? cfg->options : 0; }
static inline int64_t opt(int o) {
  if (o == L2O_OPT_BWD_BLOCKS) return (int64_t)((t_optw >> 48) & 0xffffu);     // a count: its own 16-bit field
  const unsigned nib = (unsigned)(t_optw >> (4 * L2O_OPT_FIELD_(o))) & 0xfu;
  return (nib & 8u) ? (int64_t)(nib & 7u) : kOptDefault[o];
}

// CUs of the device the call runs on (the stream's device; the current device for the null stream).
// Immutable hardware facts cached per device ordinal -- not launch state.
static int device_cu_count(hipStream_t s) {
  constexpr int kMaxDev = 64;
  static std::atomic<int> cache[kMaxDev];
  int dev = -1;
  if (s == nullptr || hipStreamGetDevice(s, &dev) != hipSuccess || dev < 0) {
    if (hipGetDevice(&dev) != hipSuccess) return 0;
  }
  if (dev < kMaxDev) {
    const int c = cache[dev].load(std::memory_order_relaxed);
    if (c > 0) return c;
  }
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  if (dev < kMaxDev) cache[dev].store(n, std::memory_order_relaxed);
  return n;
}

static int device_ordinal(hipStream_t s) {
  int dev = -1;
  if (s == nullptr || hipStreamGetDevice(s, &dev) != hipSuccess || dev < 0) {
    if (hipGetDevice(&dev) != hipSuccess) return -1;
  }
  return dev;
}
// measured co-resident one-per-CU workgroups per device (l2o_coresident_workgroups); 0 = never probed
static std::atomic<int> g_measured_cus[64];

// CUs on which workgroups of a launch on stream `s` can be resident AT THE SAME TIME -- what the kernels with
// inter-workgroup exchanges (two-CU unroll, l2o_mlp_unroll) size their grids against:
//   the device's CU count (reflects ROC_GLOBAL_CU_MASK and the partition mode), the stream's own CU mask
//   (hipExtStreamCreateWithCUMask; 0.3 us per query), and the MEASURED capacity where the caller ran the probe
//   (restrictions below HIP, e.g. HSA_CU_MASK: the device still reports every CU).
static int coresident_cus(hipStream_t s) {
  int n = device_cu_count(s);
  uint32_t mask[16] = {0};
  if (hipExtStreamGetCUMask(s, 16, mask) == hipSuccess) {
    int pop = 0;
    for (uint32_t w : mask) pop += __builtin_popcount(w);
    if (pop > 0 && pop < n) n = pop;
  } else {
    (void)hipGetLastError();
  }
  const int dev = device_ordinal(s);
  if (dev >= 0 && dev < 64) {
    const int m = g_measured_cus[dev].load(std::memory_order_relaxed);
    if (m > 0 && m < n) n = m;
  }
  return n;
}

// The probe: one workgroup per CU (100 KB of LDS each), every workgroup counts itself, waits a FIXED ~30 us (the
// dispatch of a grid this size takes a few) and records how many had arrived by then; the minimum over the workgroups
// is the number that were resident together (a second wave of workgroups sees the full count, the first wave its own
// size).  Bounded: no workgroup waits for another.
__global__ __launch_bounds__(256) void k_coresident_probe(unsigned* ctr, unsigned* min_seen) {
  extern __shared__ float probe_lds[];
  if (threadIdx.x == 0) {
    probe_lds[0] = 1.0f;
    atomicAdd(ctr, 1u);
    for (int i = 0; i < 8; ++i) __builtin_amdgcn_s_sleep(127);        // 8 x 127 x 64 clocks ~ 28 us
    const unsigned v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    atomicMin(min_seen, v);
  }
}

// Split every problem over two workgroups when that still leaves all of them co-resident.
// Problems per launch of the two-CU form, 0 = not this form.  Every problem of a launch must be co-resident with its
// partner: at most #CU / 2 per launch.  A larger batch shard runs as consecutive launches of equal chunks (a multiple
// of the 8-problem launch groups) -- normal-matrix kernel only; a chunk launch with every tile on its own SIMD beats
// one round of the one-CU form (config 4, 1024 problems on one GPU: 5.3 -> 6.0 G coordinate-steps/s).
static int pair_chunk(const l2o_problem* p, const UnrollGeom& g, hipStream_t s) {
  if (!opt(L2O_OPT_PAIR) || g.CH < 2) return 0;
  const int cap = coresident_cus(s) / 2;
  if (p->B_local <= cap) return p->B_local;
  if (cap < 8) return 0;
  const int n = (p->B_local + cap - 1) / cap;               // launches
  const int chunk = (((p->B_local + n - 1) / n) + 7) & ~7;  // balanced, whole launch groups
  return chunk <= cap ? chunk : (cap & ~7);
}
struct PairLayout { size_t xbuf_off, xbuf_bytes, fxh_off, total; int npg; size_t lds; };
static PairLayout pair_layout(const l2o_problem* p, const UnrollGeom& g, int T) {
  PairLayout L;
  const int SQ = 16 * g.CH;
  L.npg = SQ;                         // granules per (half, parity): one per residual row
  L.xbuf_off = sizeof(PairWs);
  L.xbuf_bytes = (size_t)p->B_local * 2 * 2 * L.npg * sizeof(unsigned long long);
  L.fxh_off = L.xbuf_off + L.xbuf_bytes;
  L.total = L.fxh_off + sizeof(float) * (size_t)(T + 1) * g.CH * p->B_local;   // one partial per (step, problem, wave)
  L.lds = 0;                          // static LDS only (xs, rs, fpart)
  return L;
}

template <int PRE, int KIND>
static int launch_unroll_ch(const UnrollArgs& a, const UnrollGeom& g, hipStream_t s, const l2o_problem* prob,
                            void* workspace, float* fx, bool* fx_done) {
  const bool hist = a.hist_st != nullptr;
  int chunk = workspace ? pair_chunk(prob, g, s) : 0;
  // Large shards (round 4) -- more problems than the #CU / 2 one launch of the two-CU kernel holds -- run k_unroll_lds: one
  // problem per CU, the gate-GEMM fragments in LDS, two waves of the SAME problem per SIMD, one launch, no cross-CU
  // protocol.  L2O_OPT_ONE_LDS: 0 = consecutive chunk launches of the two-CU kernel instead (rounds 2-3; A/B runs), 1 = the
  // default, 2 = k_unroll_lds for every shard (A/B runs).  5..8 tiles, no exact-gates request.  Also the exchange-free
  // fallback of those shapes when the two-CU form is not available (no workspace, L2O_OPT_PAIR = 0; the host's recovery
  // from a partner timeout): the alternative there is k_unroll's fp32-MFMA form, 11 200 cycles per step against 7 200.
  // (k_unroll_pair2 -- the two-CU kernel with fragments in LDS, two workgroups per CU -- measured the same as k_unroll_lds
  //  in round 4 and was removed in round 5: docs/DESIGN_history_r04.md.)
  {
    const int one_lds = (int)opt(L2O_OPT_ONE_LDS);
#ifdef L2O_LDS_ABL_ANYNW   // (timing ablation: also 1..4 tiles, i.e. ONE wave per SIMD in this kernel; needs M <= 16 nw)
    const bool shape_ok = g.nw >= 1 && a.pp.M <= 16 * g.nw;
#else
    const bool shape_ok = g.CH == 8 && g.nw >= 5;
#endif
    const bool plain = !(opt(L2O_OPT_EXACT_GATES) && !hist);
    if ((one_lds == 2 || (one_lds == 1 && (chunk == 0 || a.pp.B_local > chunk))) && shape_ok && plain) {
      UnrollArgs al = a;
      al.ticks = workspace ? &reinterpret_cast<PairWs*>(workspace)->ticks : nullptr;
      void (*fl)(UnrollArgs) = hist ? k_unroll_lds<PRE, KIND, true> : k_unroll_lds<PRE, KIND, false>;
      const size_t lds = sizeof(float) * ((size_t)LstmCoreLds<PRE>::kFragWords + 2 * 128 + 8);
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fl), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(fl, dim3(a.pp.B_local), dim3(64 * g.nw), lds, s, al);
      HIP_TRY(hipGetLastError());
      note_form(L2O_FORM_UNROLL_LDS);
      return L2O_OK;
    }
  }
  if (chunk > 0) {
    const PairLayout L = pair_layout(prob, g, a.T);
    UnrollPairArgs pa;
    pa.u = a;
    pa.ws = reinterpret_cast<PairWs*>(workspace);
    pa.xbuf = reinterpret_cast<unsigned long long*>(static_cast<char*>(workspace) + L.xbuf_off);
    pa.fx_half = reinterpret_cast<float*>(static_cast<char*>(workspace) + L.fxh_off);
    // (the per-launch salt of the granule tags is the device-side sequence word ws->seq, which
    //  k_combine_halves advances after every launch: no host-side launch state)
    pa.use_salt = a.T + 1 < 0xffff ? 1u : 0u;
    pa.plain_stores = opt(L2O_OPT_PAIR_PLAIN_STORES) ? 1u : 0u;
    pa.b0 = 0;
    pa.nb = a.pp.B_local;
    // (no memset here: the workspace starts zeroed -- l2o_unroll_workspace_init -- and the epilogue kernel of every
    //  launch leaves the granule area zeroed for the next one)
    // grid: groups of 16 blocks = 8 problems x 2 halves (partners are b and b + 8)
    const int B = a.pp.B_local;
    const bool one_launch = chunk >= B;
    const bool exact = opt(L2O_OPT_EXACT_GATES) != 0 && !hist;   // (the recording unroll keeps the bf16x3 core)
    {
      void (*fn)(UnrollPairArgs) = nullptr;
      const size_t dyn_lds = L.lds;
      switch (g.CH) {
        case 2: fn = hist ? k_unroll_pair<PRE, KIND, 2, true> : (exact ? k_unroll_pair<PRE, KIND, 2, false, true> : k_unroll_pair<PRE, KIND, 2, false>); break;
        case 4: fn = hist ? k_unroll_pair<PRE, KIND, 4, true> : (exact ? k_unroll_pair<PRE, KIND, 4, false, true> : k_unroll_pair<PRE, KIND, 4, false>); break;
        default: fn = hist ? k_unroll_pair<PRE, KIND, 8, true> : (exact ? k_unroll_pair<PRE, KIND, 8, false, true> : k_unroll_pair<PRE, KIND, 8, false>); break;
      }
      for (int b0 = 0; b0 < B; b0 += chunk) {
        pa.b0 = b0;
        pa.nb = B - b0 < chunk ? B - b0 : chunk;
        hipLaunchKernelGGL(fn, dim3((pa.nb + 7) / 8 * 16), dim3(64 * (g.CH / 2)), dyn_lds, s, pa);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_combine_halves, dim3(a.T + 1), dim3(256), 0, s, pa.fx_half, a.fx_part, pa.nb, g.CH,
                           a.pp.inv_bg, one_launch ? fx : nullptr, pa.xbuf, (long)pa.nb * 2 * 2 * L.npg, pa.ws, b0, B);
        HIP_TRY(hipGetLastError());
      }
    }
    note_form(L2O_FORM_UNROLL_PAIR, (B + chunk - 1) / chunk);
    if (fx_done) *fx_done = fx != nullptr && one_launch;
    return L2O_OK;
  }
  void (*fn)(UnrollArgs) = nullptr;
  const bool exact1 = opt(L2O_OPT_EXACT_GATES) != 0 && !hist;
  switch (g.CH) {
    case 1: fn = hist ? k_unroll<PRE, KIND, 1, true> : (exact1 ? k_unroll<PRE, KIND, 1, false, true> : k_unroll<PRE, KIND, 1, false>); break;
    case 2: fn = hist ? k_unroll<PRE, KIND, 2, true> : (exact1 ? k_unroll<PRE, KIND, 2, false, true> : k_unroll<PRE, KIND, 2, false>); break;
    case 4: fn = hist ? k_unroll<PRE, KIND, 4, true> : (exact1 ? k_unroll<PRE, KIND, 4, false, true> : k_unroll<PRE, KIND, 4, false>); break;
    default: fn = hist ? k_unroll<PRE, KIND, 8, true> : k_unroll<PRE, KIND, 8, false>; break;
  }
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)g.lds));
  hipLaunchKernelGGL(fn, dim3(a.pp.B_local), dim3(64 * g.nw), g.lds, s, a);
  HIP_TRY(hipGetLastError());
  note_form(L2O_FORM_UNROLL);
  return L2O_OK;
}

template <int PRE>
static int launch_unroll_kind(const UnrollArgs& a, const UnrollGeom& g, int kind, hipStream_t s,
                              const l2o_problem* prob, void* workspace, float* fx, bool* fx_done) {
  switch (kind) {
    case L2O_PROB_QUADRATIC: return launch_unroll_ch<PRE, L2O_PROB_QUADRATIC>(a, g, s, prob, workspace, fx, fx_done);
    case L2O_PROB_LASSO: return launch_unroll_ch<PRE, L2O_PROB_LASSO>(a, g, s, prob, workspace, fx, fx_done);
    case L2O_PROB_RASTRIGIN: return launch_unroll_ch<PRE, L2O_PROB_RASTRIGIN>(a, g, s, prob, workspace, fx, fx_done);
    case L2O_PROB_SQUARE_COS: return launch_unroll_ch<PRE, L2O_PROB_SQUARE_COS>(a, g, s, prob, workspace, fx, fx_done);
    default: return fail(L2O_ERR_UNSUPPORTED, "no fused kernel for problem kind %d", kind);
  }
}

// The streaming form (csrc/l2o_unroll_cu.h) takes the sizes the LDS-resident kernels cannot: one
// workgroup per problem, matrix streamed once per step, LSTM state in LDS (+ registers).
static bool unroll_cu_eligible(const l2o_problem* p) {
  if (!opt(L2O_OPT_UNROLL_CU)) return false;
  if (p->D < 4 || p->D > 512 || (p->D & 3) || p->M <= 0) return false;   // (D <= 128: only when the rows do not fit the LDS forms)
  return unroll_cu_layout(p->D).lds + sizeof(float) * bx::kBiasWords <= 160 * 1024;   // (+ the static bias table)
}
// L2O_OPT_UNROLL_CU: 0 step-granular path, 1* the eight-wave form (k_unroll_cu8: two waves per SIMD, fragments in LDS, LSTM
// state in registers) for RNNProp's plain unroll and the four-wave form (k_unroll_cu) otherwise, 2 k_unroll_cu always,
// 3 k_unroll_cu8 always, 4 k_unroll_cu8 with three register tiles + one LDS slot per wave (A/B runs)
template <int PRE, int KR>
static int launch_unroll_cu8(const UnrollArgs& a, hipStream_t s) {
  const UnrollCu8Layout L = unroll_cu8_layout(a.pp.D, PRE, KR);
  const bool hist = a.hist_st != nullptr;
  void (*fn)(UnrollArgs) = a.pp.D <= 256 ? (hist ? k_unroll_cu8<PRE, 1, KR, true> : k_unroll_cu8<PRE, 1, KR, false>)
                                         : (hist ? k_unroll_cu8<PRE, 2, KR, true> : k_unroll_cu8<PRE, 2, KR, false>);
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.lds));
  hipLaunchKernelGGL(fn, dim3(a.pp.B_local), dim3(kCu8Threads), L.lds, s, a);
  note_form(L2O_FORM_UNROLL_CU8);
  HIP_TRY(hipGetLastError());
  return L2O_OK;
}
template <int PRE>
static int launch_unroll_cu(const UnrollArgs& a_in, hipStream_t s) {
  const int form = (int)opt(L2O_OPT_UNROLL_CU);
  // Default (1): the eight-wave form (k_unroll_cu8) with the number of register-resident state tiles per wave (KR) that the
  // instantiation holds without spilling at 256 registers (scripts/tu_regs.sh; D = 512): RNNProp's plain unroll 4 (config 3:
  // 3 spilled registers, measured best), the DM nets' plain unroll 3 (round 5: their input-weight rows moved to LDS; 0
  // spills -- with 4: 31-37; measured on DM / Lasso 256 x 512, batch 256: 5.25 G four-wave, 5.26 G KR = 4, 6.24 G KR = 3),
  // every recording unroll 2 (0 / 10 spills) where that LDS image fits (two more 5 KB state slots per wave: D <= 256), the
  // DM nets' recording unroll at larger D 3 (8 spills); RNNProp's recording unroll at D > 256 stays on the four-wave kernel
  // (KR = 3: 52 spills).  2: k_unroll_cu always; 3 / 4 / 5: k_unroll_cu8 always with KR = 4 / 3 / 2 where it fits (A/B runs).
  const bool hist0 = a_in.hist_st != nullptr;
  auto fits = [&](int kr) { return unroll_cu8_layout(a_in.pp.D, PRE, kr).lds + sizeof(float) * bx::kBiasWords <= 160 * 1024; };
  int KR = 0;
  if (form == 1) {
    KR = hist0 ? 2 : (PRE == L2O_PRE_FC_ELU ? 4 : 3);
    if (hist0 && !fits(2) && PRE != L2O_PRE_FC_ELU) KR = 3;
  } else if (form >= 3 && form <= 5) KR = 7 - form;
  if (KR && fits(KR))
    return KR == 4 ? launch_unroll_cu8<PRE, 4>(a_in, s) : (KR == 3 ? launch_unroll_cu8<PRE, 3>(a_in, s) : launch_unroll_cu8<PRE, 2>(a_in, s));
  const UnrollCuLayout L = unroll_cu_layout(a_in.pp.D);
  const UnrollArgs& a = a_in;
  const bool hist = a.hist_st != nullptr;
  void (*fn)(UnrollArgs) = a.pp.D <= 256 ? (hist ? k_unroll_cu<PRE, 1, true> : k_unroll_cu<PRE, 1, false>)
                                         : (hist ? k_unroll_cu<PRE, 2, true> : k_unroll_cu<PRE, 2, false>);
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)L.lds));
  hipLaunchKernelGGL(fn, dim3(a.pp.B_local), dim3(kCuThreads), L.lds, s, a);
  note_form(L2O_FORM_UNROLL_CU);
  HIP_TRY(hipGetLastError());
  return L2O_OK;
}

extern "C" {

int l2o_abi_version(void) { return L2O_ABI_VERSION; }
const char* l2o_last_error(void) { return g_err; }
#ifndef L2O_BUILD_ID
#define L2O_BUILD_ID "unknown"
#endif
const char* l2o_build_id(void) { return L2O_BUILD_ID; }
int l2o_last_unroll_form(void) { return g_last_form; }

size_t l2o_wpack_floats(const l2o_net_cfg* cfg) {
  if (!cfg) return 0;
  if (cfg->n_layers == 0) return 4;
  if (!net_ok_for_mfma(cfg)) return 0;
  return (size_t)wp_rows(cfg->preprocess) * 64 + (size_t)bx::words(cfg->preprocess) + (size_t)bxb::words(cfg->preprocess);
}

// ---- bf16x3 section of wpack (l2o_lstm_bx3.h) --------------------------------
__host__ __device__ static inline uint16_t bf16_rne(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__host__ __device__ static inline double bf16_value(uint16_t h) {
  return (double)__builtin_bit_cast(float, (uint32_t)h << 16);
}
__host__ __device__ static inline void bf16_split3(double v, uint16_t (&out)[3]) {
  for (int s = 0; s < 3; ++s) {
    out[s] = bf16_rne((float)v);
    v -= bf16_value(out[s]);
  }
}

// Lane l's share of the packed weights (the buffer is zero beforehand): the host packer runs it for
// l = 0..63, the device packer (l2o_wpack_device, the meta-training step) with one thread per lane.
// The lanes write disjoint words.
// [t_lo, t_hi): unit slices of the forward sections, [tile_lo, tile_hi): M-tiles of the BPTT section (the device
// packer gives every (lane, slice) and (lane, tile) its own thread: the float64 splits are the cost).
// (ch_lo, ch_hi): the gate-GEMM chunks of slices [t_lo, t_hi) this call packs -- everything of a slice that is not a
// chunk fragment rides with chunk 0; (r_lo, r_hi): the gate types of the BPTT M-tiles [tile_lo, tile_hi).  The host
// packer passes the full ranges; k_wpack one (slice, chunk) or one (tile, gate type) per block.
__host__ __device__ static void wpack_lane(int pre, int l, int t_lo, int t_hi, int tile_lo, int tile_hi,
                                           const float* wg1, const float* bg1, const float* wg2,
                                           const float* bg2, const float* wl, const float* bl, const float* wfc,
                                           const float* bfc, float* out, int ch_lo = 0, int ch_hi = 4, int r_lo = 0,
                                           int r_hi = 4) {
  const bool fc = pre == L2O_PRE_FC_ELU;
  const int P = fc ? kH : (pre == L2O_PRE_LOGSIGN ? 2 : 1);
  const int G = 4 * kH;
  auto col = [](int t, int rho) { return (rho & 3) * kH + 4 * t + (rho >> 2); };
  const int rho = l & 15, kq = l >> 4;   // A-fragment view of the lane
  const int q = l >> 4;                  // C/D + B view of the lane
  for (int t = t_lo; t < t_hi && ch_lo == 0; ++t) {
    const int cA = col(t, rho);
    // layer 1
    if (fc) {
      for (int kk = 0; kk < 10; ++kk) {
        const int row = kk < 5 ? 4 * kk + kq : kH + 4 * (kk - 5) + kq;   // fc features, then h1
        out[(wp_row_a1(pre) + kk * kNT + t) * 64 + l] = wg1[row * G + cA];
      }
    } else {
      for (int kk = 0; kk < 5; ++kk)
        out[(wp_row_a1(pre) + kk * kNT + t) * 64 + l] = wg1[(P + 4 * kk + kq) * G + cA];
      float v = 0.0f;
      if (kq == 0) v = wg1[0 * G + cA];
      else if (kq == 1) v = P == 2 ? wg1[1 * G + cA] : 0.0f;
      else if (kq == 2) v = bg1[cA];
      out[(wp_row_a1(pre) + 5 * kNT + t) * 64 + l] = v;
    }
    // layer 2: kk 0..4 <- h1 (rows 0..19), kk 5..9 <- h2 (rows 20..39)
    for (int kk = 0; kk < 10; ++kk) {
      const int row = kk < 5 ? 4 * kk + kq : kH + 4 * (kk - 5) + kq;
      out[(wp_row_a2(pre) + kk * kNT + t) * 64 + l] = wg2[row * G + cA];
    }
    for (int r = 0; r < 4; ++r) {
      const int cD = col(t, 4 * q + r);
      out[(wp_row_b1(pre) + t * 4 + r) * 64 + l] = fc ? bg1[cD] : 0.0f;
      out[(wp_row_b2(pre) + t * 4 + r) * 64 + l] = bg2[cD];
    }
    out[(wp_row_wl(pre) + t) * 64 + l] = wl[4 * t + q];
    if (fc) {
      out[(wp_row_fc(pre) + t) * 64 + l] = wfc[0 * kH + 4 * t + q];
      out[(wp_row_fc(pre) + kNT + t) * 64 + l] = wfc[1 * kH + 4 * t + q];
      out[(wp_row_fc(pre) + 2 * kNT + t) * 64 + l] = bfc[4 * t + q];
    }
  }
  if (t_lo == 0 && t_hi > 0 && ch_lo == 0) out[wp_row_bl(pre) * 64 + l] = bl[0];
  // ---- bf16x3 fragments: gate rows pre-scaled to exp2 arguments, split from float64 ----
  {
    uint32_t* ow = reinterpret_cast<uint32_t*>(out);
    constexpr double kL2E = 1.4426950408889634074;
    auto gscale = [&](int r) { return r == 1 ? 2.0 * kL2E : -kL2E; };      // rows i, j, f, o
    for (int t = t_lo; t < t_hi; ++t) {
      const int cA = col(t, rho), r = rho & 3;
      for (int ch = ch_lo; ch < ch_hi && ch < bx::nchunks(pre); ++ch) {
        uint16_t sl[5][3], bs[3] = {0, 0, 0};
        for (int i = 0; i < 5; ++i) {
          const int u = 4 * i + kq;
          double v;
          if (ch == bx::kChL1H) v = wg1[(P + u) * G + cA];
          else if (ch == bx::kChL2A) v = wg2[u * G + cA];
          else if (ch == bx::kChL2B) v = wg2[(kH + u) * G + cA];
          else v = wg1[u * G + cA];                              // kChL1X: the fc features
          bf16_split3(v * gscale(r), sl[i]);
        }
        // (the bias slots of the fragments stay ZERO since round 3: the bias is the accumulator init, bx::bias_off)
        // 6-product form: K-slots 0..4 = the units, slot 7 = the bias (q == 0), one fragment per split level
        for (int sp = 0; sp < 3; ++sp)
          for (int rg = 0; rg < 4; ++rg) {
            auto slot = [&](int i) -> uint32_t { return i < 5 ? sl[i][sp] : (i == 7 && kq == 0 ? bs[sp] : 0); };
            ow[bx::frag_off(pre, false, ch, t, sp) + l * 4 + rg] = slot(2 * rg) | (slot(2 * rg + 1) << 16);
          }
        // packed K-slots (l2o_lstm_bx3.h, slot_desc): the weight level that multiplies the slot's activation level
        {
          for (int j = 0; j < bx::kPack; ++j)
            for (int rg = 0; rg < 4; ++rg) {
              uint32_t word = 0;
              for (int h = 0; h < 2; ++h) {
                const bx::SlotDesc d = bx::slot_desc(j, rg, h);
                uint16_t v16 = 0;
                if (d.unit < 5) v16 = sl[d.unit][d.w];
                else if (bx::bias_level(kq, h) >= 0) v16 = bs[bx::bias_level(kq, h)];
                word |= (uint32_t)v16 << (16 * h);
              }
              ow[bx::frag_off(pre, true, ch, t, j) + l * 4 + rg] = word;
            }
        }
      }
      if (ch_lo != 0) continue;                                 // (the rest of the slice rides with chunk 0)
      if ((l & 15) == 0)                                        // one lane per lane group writes its 4 gate rows
        for (int rr = 0; rr < 4; ++rr) {
          const int cD = col(t, 4 * q + rr);
          const double fb = rr == 2 ? 1.0 : 0.0;                  // forget_bias
          out[bx::bias_off(pre) + ((0 * kNT + t) * 4 + q) * 4 + rr] = (float)(((double)bg1[cD] + fb) * gscale(rr));
          out[bx::bias_off(pre) + ((1 * kNT + t) * 4 + q) * 4 + rr] = (float)(((double)bg2[cD] + fb) * gscale(rr));
        }
      if (!fc)
        for (int rr = 0; rr < 4; ++rr) {
          const int cD = col(t, 4 * q + rr);
          out[bx::win_off(pre) + t * 256 + l * 4 + rr] = (float)((double)wg1[0 * G + cD] * gscale(rr));
          out[bx::win_off(pre) + (kNT + t) * 256 + l * 4 + rr] =
              P == 2 ? (float)((double)wg1[1 * G + cD] * gscale(rr)) : 0.0f;
        }
    }
  }
  // ---- bf16x3 fragments of the transposed products of the BPTT step (l2o_bwd_mfma.h): W itself,
  //      unscaled; M-tile rows = input rows of W, K-chunk r = gate type r (Sonnet column block) ----
  {
    uint32_t* ow = reinterpret_cast<uint32_t*>(out) + bxb::base(pre);
    for (int tile = tile_lo; tile < tile_hi; ++tile) {
      const bool l2 = tile < bxb::tiles2();
      const int m = l2 ? tile : tile - bxb::tiles2();
      const float* W = l2 ? wg2 : wg1;
      const int first = l2 ? 0 : (fc ? 0 : -1), second = l2 ? kH : (fc ? kH : P);
      const int row = bxb::src_row(m, rho, first, second);
      if (row < 0) continue;
      for (int r = r_lo; r < r_hi; ++r) {
        uint16_t sl[8][3];
        for (int i = 0; i < 8; ++i)
          for (int sp = 0; sp < 3; ++sp) sl[i][sp] = 0;
        for (int i = 0; i < 5; ++i) bf16_split3((double)W[row * G + r * kH + 4 * i + kq], sl[i]);
        for (int sp = 0; sp < 3; ++sp)
          for (int j = 0; j < 4; ++j)
            ow[bxb::frag_rel(tile, r, sp) + l * 4 + j] = (uint32_t)sl[2 * j][sp] | ((uint32_t)sl[2 * j + 1][sp] << 16);
      }
    }
  }
}

__global__ void k_wpack(int pre, const float* wg1, const float* bg1, const float* wg2, const float* bg2, const float* wl,
                        const float* bl, const float* wfc, const float* bfc, float* out) {
  // one block per (slice, chunk), then one per (BPTT M-tile, gate type): 35 / 44 blocks of one wave (10 / 11 blocks
  // before: 12.5 us on the critical path between the meta-step and the next unroll)
  const int b = blockIdx.x, nch = bx::nchunks(pre);
  if (b < kNT * nch) {
    const int t = b / nch, ch = b - t * nch;
    wpack_lane(pre, threadIdx.x, t, t + 1, 0, 0, wg1, bg1, wg2, bg2, wl, bl, wfc, bfc, out, ch, ch + 1, 0, 0);
  } else {
    const int e = b - kNT * nch, tile = e >> 2, r = e & 3;
    wpack_lane(pre, threadIdx.x, 0, 0, tile, tile + 1, wg1, bg1, wg2, bg2, wl, bl, wfc, bfc, out, 0, 0, r, r + 1);
  }
}

// tf.train.AdamOptimizer._apply_dense (TF 1.x) on one flat fp32 vector, every operation rounded separately
// like the NumPy expression it replaces (no contraction): bit-equal to the host update
// guard != nullptr: the status word of the unroll whose gradients these are (the head of its workspace); a non-zero
// status (a partner timeout: the recorded history is garbage) leaves w, m, v untouched
// map != nullptr: the gradient of weight i is g[map[i]] (map[i] < 0: zero) -- the weight-gradient blocks are read
// straight out of the contraction's [KA][KB] result (l2o_adam_step_gather)
__global__ void k_adam(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v, const float* __restrict__ g,
                       long n, float lr_t, float b1, float omb1, float b2, float omb2, float eps,
                       const unsigned* __restrict__ guard, const int* __restrict__ map) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (guard && __builtin_nontemporal_load(guard) != 0u) return;
  float gi;
  if (map) { const int j = map[i]; gi = j >= 0 ? g[j] : 0.0f; }
  else gi = g[i];
  const float mi = __fadd_rn(__fmul_rn(b1, m[i]), __fmul_rn(omb1, gi));
  const float vi = __fadd_rn(__fmul_rn(b2, v[i]), __fmul_rn(__fmul_rn(omb2, gi), gi));
  m[i] = mi; v[i] = vi;
  w[i] = __fsub_rn(w[i], __fdiv_rn(__fmul_rn(lr_t, mi), __fadd_rn(__fsqrt_rn(vi), eps)));
}

int l2o_wpack_host(const l2o_net_cfg* cfg, const float* wg1, const float* bg1, const float* wg2,
                   const float* bg2, const float* wl, const float* bl, const float* wfc, const float* bfc,
                   float* out) {
  if (!cfg || !out || !wl || !bl) return fail(L2O_ERR_ARG, "l2o_wpack_host: NULL argument");
  if (cfg->n_layers == 0) {
    // Linear(P -> 1) on the preprocessed gradient, P = 1 (identity) or 2 (LogAndSign)
    if (cfg->kind != L2O_NET_CW || cfg->preprocess == L2O_PRE_FC_ELU)
      return fail(L2O_ERR_UNSUPPORTED, "layers=() is implemented for CoordinateWiseDeepLSTM only");
    out[0] = wl[0];
    out[1] = cfg->preprocess == L2O_PRE_LOGSIGN ? wl[1] : 0.0f;
    out[2] = bl[0];
    out[3] = 0.0f;
    return L2O_OK;
  }
  if (!net_ok_for_mfma(cfg))
    return fail(L2O_ERR_UNSUPPORTED, "MFMA kernels implement layers=(20,20) (got n_layers=%d hidden=%d kind=%d pre=%d)",
                cfg->n_layers, cfg->hidden, cfg->kind, cfg->preprocess);
  if (!wg1 || !bg1 || !wg2 || !bg2) return fail(L2O_ERR_ARG, "l2o_wpack_host: NULL LSTM weights");
  const int pre = cfg->preprocess;
  const bool fc = pre == L2O_PRE_FC_ELU;
  if (fc && (!wfc || !bfc)) return fail(L2O_ERR_ARG, "l2o_wpack_host: fc preprocess needs input_projection");
  std::memset(out, 0, sizeof(float) * l2o_wpack_floats(cfg));
  for (int l = 0; l < 64; ++l) wpack_lane(pre, l, 0, kNT, 0, bxb::ntiles(pre), wg1, bg1, wg2, bg2, wl, bl, wfc, bfc, out);
  return L2O_OK;
}

int l2o_wpack_device(const l2o_net_cfg* cfg, const l2o_net_weights* w, float* wpack, void* stream) {
  OptScope opt_scope(cfg_optw(cfg));
  if (!cfg || !w || !wpack) return fail(L2O_ERR_ARG, "l2o_wpack_device: NULL argument");
  if (!net_ok_for_mfma(cfg) || cfg->n_layers == 0)
    return fail(L2O_ERR_UNSUPPORTED, "l2o_wpack_device: layers=(20,20) nets only");
  const bool fc = cfg->preprocess == L2O_PRE_FC_ELU;
  if (!w->w_gates1 || !w->b_gates1 || !w->w_gates2 || !w->b_gates2 || !w->w_lin || !w->b_lin ||
      (fc && (!w->w_fc || !w->b_fc)))
    return fail(L2O_ERR_ARG, "l2o_wpack_device: NULL weight pointer");
  hipStream_t s = (hipStream_t)stream;
  if (!opt(L2O_OPT_WPACK_NO_CLEAR)) HIP_TRY(hipMemsetAsync(wpack, 0, sizeof(float) * l2o_wpack_floats(cfg), s));
  hipLaunchKernelGGL(k_wpack, dim3(kNT * bx::nchunks(cfg->preprocess) + 4 * bxb::ntiles(cfg->preprocess)), dim3(64), 0, s, (int)cfg->preprocess, w->w_gates1, w->b_gates1, w->w_gates2,
                     w->b_gates2, w->w_lin, w->b_lin, w->w_fc, w->b_fc, wpack);
  HIP_TRY(hipGetLastError());
  return L2O_OK;
}

static int adam_launch(float* w, float* m, float* v, const float* g, int64_t n, float lr_t, double beta1, double beta2,
                       double epsilon, const unsigned* guard, void* stream, const int* map = nullptr) {
  if (!w || !m || !v || !g || n < 0) return fail(L2O_ERR_ARG, "l2o_adam_step: bad argument");
  if (n == 0) return L2O_OK;
  hipLaunchKernelGGL(k_adam, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, m, v, g, (long)n,
                     lr_t, (float)beta1, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)epsilon, guard, map);
  HIP_TRY(hipGetLastError());
  return L2O_OK;
}
int l2o_adam_step(float* w, float* m, float* v, const float* g, int64_t n, float lr_t, double beta1, double beta2,
                  double epsilon, void* stream) {
  return adam_launch(w, m, v, g, n, lr_t, beta1, beta2, epsilon, nullptr, stream);
}
int l2o_adam_step_gather(float* w, float* m, float* v, const float* G, const int32_t* map, int64_t n, float lr_t,
                         double beta1, double beta2, double epsilon, const void* unroll_workspace, void* stream) {
  if (!map) return fail(L2O_ERR_ARG, "l2o_adam_step_gather: NULL map");
  return adam_launch(w, m, v, G, n, lr_t, beta1, beta2, epsilon, static_cast<const unsigned*>(unroll_workspace), stream, map);
}
int l2o_adam_step_guarded(float* w, float* m, float* v, const float* g, int64_t n, float lr_t, double beta1,
                          double beta2, double epsilon, const void* unroll_workspace, void* stream) {
  // (the status word is the first member of the workspace header: l2o_unroll_status reads the same 4 bytes)
  return adam_launch(w, m, v, g, n, lr_t, beta1, beta2, epsilon, static_cast<const unsigned*>(unroll_workspace), stream);
}

size_t l2o_state_floats(int64_t B, int64_t D) {
  if (B <= 0 || D <= 0) return 0;
  return (size_t)B * tiles_per_problem(D) * kStateFloatsPerTile;
}

static int state_repack(float* h1, float* c1, float* h2, float* c2, float* st, int64_t B, int64_t D,
                        void* stream, int to_packed) {
  if (!h1 || !c1 || !h2 || !c2 || !st || B <= 0 || D <= 0) return fail(L2O_ERR_ARG, "state repack: bad argument");
  const int tpp = tiles_per_problem(D);
  const size_t total = (size_t)B * tpp * 64 * 20;
  hipLaunchKernelGGL(k_state_repack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h1,
                     c1, h2, c2, st, (int)B, (int)D, tpp, to_packed);
  HIP_TRY(hipGetLastError());
  return L2O_OK;
}
int l2o_state_pack(const float* h1, const float* c1, const float* h2, const float* c2, float* st, int64_t B,
                   int64_t D, void* stream) {
  return state_repack(const_cast<float*>(h1), const_cast<float*>(c1), const_cast<float*>(h2),
                      const_cast<float*>(c2), st, B, D, stream, 1);
}
int l2o_state_unpack(const float* st, float* h1, float* c1, float* h2, float* c2, int64_t B, int64_t D,
                     void* stream) {
  return state_repack(h1, c1, h2, c2, const_cast<float*>(st), B, D, stream, 0);
}

static int launch_problem_fg(const ProbParams& pp, const float* x, float* f_part, float* g, void* stream) {
  const size_t lds = sizeof(float) * ((size_t)((pp.D + 3) & ~3) + ((pp.M + 3) & ~3) + 4 * kFgThreads + kFgWaves);
  if (lds > 160 * 1024) return fail(L2O_ERR_UNSUPPORTED, "problem too large for k_problem_fg (D=%d M=%d)", pp.D, pp.M);
  const bool vec = pp.kind != L2O_PROB_SIMPLE && (pp.D & 3) == 0 && ((uintptr_t)pp.W & 15) == 0;
  if (vec && pp.D >= 64 && pp.D <= 2048 && ((uintptr_t)x & 15) == 0 && (!pp.x_scale || ((uintptr_t)pp.x_scale & 15) == 0) &&
      !pp.w_shared && !opt(L2O_OPT_FG_TWO_PASS)) {         // single pass over the matrix (a shared, L2-resident
                                                            // matrix is faster in two passes: 20 vs 28 us for config 3)
    const size_t lds1 = sizeof(float) * ((size_t)kFgWaves * pp.D + kFgWaves);
    void (*f1)(ProbParams, const float*, float*, float*) =
        pp.D <= 256 ? k_problem_fg1<1> : pp.D <= 512 ? k_problem_fg1<2> : pp.D <= 1024 ? k_problem_fg1<4> : k_problem_fg1<8>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(f1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
    hipLaunchKernelGGL(f1, dim3(pp.B_local), dim3(kFgThreads), lds1, (hipStream_t)stream, pp, x, f_part, g);
    HIP_TRY(hipGetLastError());
    return L2O_OK;
  }
  void (*fn)(ProbParams, const float*, float*, float*) = vec ? k_problem_fg<true> : k_problem_fg<false>;
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds));
  hipLaunchKernelGGL(fn, dim3(pp.B_local), dim3(kFgThreads), lds, (hipStream_t)stream, pp, x, f_part, g);
  HIP_TRY(hipGetLastError());
  return L2O_OK;
}

int l2o_problem_fg(const l2o_problem* prob, const float* x, float* f_part, float* g, void* stream) {
  OptScope opt_scope((prob && (prob->flags & L2O_PROB_FG_TWO_PASS)) ? L2O_OPTW(L2O_OPT_FG_TWO_PASS, 1) : 0);
  int rc = check_problem(prob);
  if (rc) return rc;
  if (!x || !f_part) return fail(L2O_ERR_ARG, "l2o_problem_fg: NULL x / f_part");
  return launch_problem_fg(make_prob_params(prob), x, f_part, g, stream);
}

// out += s * inv_bg * f''(x s) * s * u for the separable cosine term (rastrigin / square_cos)
__global__ void k_hvp_sep(ProbParams pp, const float* __restrict__ x, const float* __restrict__ u, float* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float sc = pp.x_scale ? pp.x_scale[i] : 1.0f;
  const float xs = x[i] * sc;
  out[i] += sc * pp.inv_bg * (pp.twopi * pp.twopi * pp.alpha * pp.C[i] * l2o::cos_f(pp.twopi * xs)) * sc * u[i];
}

int l2o_problem_hvp(const l2o_problem* prob, const float* x, const float* u, float* out, float* scratch, void* stream) {
  OptScope opt_scope((prob && (prob->flags & L2O_PROB_FG_TWO_PASS)) ? L2O_OPTW(L2O_OPT_FG_TWO_PASS, 1) : 0);
  int rc = check_problem(prob);
  if (rc) return rc;
  if (!x || !u || !out || !scratch) return fail(L2O_ERR_ARG, "l2o_problem_hvp: NULL argument");
  ProbParams pp = make_prob_params(prob);
  pp.hvp = 1;
  rc = launch_problem_fg(pp, u, scratch, out, stream);     // cg s inv_bg W^T (W (s u))  (simple: 2 s^2 u)
  if (rc) return rc;
  if (pp.kind == L2O_PROB_RASTRIGIN || pp.kind == L2O_PROB_SQUARE_COS) {
    const size_t n = (size_t)pp.B_local * pp.D;
    hipLaunchKernelGGL(k_hvp_sep, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pp, x, u, out, n);
    HIP_TRY(hipGetLastError());
  }
  return L2O_OK;
}

int l2o_mlp_fg(const l2o_mlp* mlp, const int32_t* indices, const float* w1, const float* b1, const float* w2,
               const float* b2, float* loss, float* gw1, float* gb1, float* gw2, float* gb2, float* scratch,
               void* stream) {
  OptScope opt_scope((mlp && (mlp->flags & L2O_MLP_GENERIC)) ? L2O_OPTW(L2O_OPT_MLP_GENERIC, 1) : 0);
  if (!mlp || !indices || !w1 || !b1 || !w2 || !b2 || !loss || !mlp->images || !mlp->labels)
    return fail(L2O_ERR_ARG, "l2o_mlp_fg: NULL argument");
  const bool want_g = gw1 || gb1 || gw2 || gb2;
  if (want_g && !(gw1 && gb1 && gw2 && gb2)) return fail(L2O_ERR_ARG, "l2o_mlp_fg: pass all four gradients or none");
  if (mlp->n_hidden < 1 || mlp->n_hidden > kMlpMaxH || mlp->n_out < 1 || mlp->n_out > kMlpMaxO ||
      mlp->batch < 1 || mlp->batch > 256 || mlp->n_in < 1 || mlp->n_in > kMlpTPS * kMlpKPT)
    return fail(L2O_ERR_UNSUPPORTED, "l2o_mlp_fg: sizes n_in=%d hidden=%d out=%d batch=%d not implemented",
                mlp->n_in, mlp->n_hidden, mlp->n_out, mlp->batch);
  if (!scratch) return fail(L2O_ERR_ARG, "l2o_mlp_fg: NULL scratch (l2o_mlp_scratch_floats)");
  MlpParams p;
  p.n_in = mlp->n_in; p.H = mlp->n_hidden; p.O = mlp->n_out; p.batch = mlp->batch; p.act = mlp->activation;
  p.images = mlp->images; p.labels = mlp->labels; p.idx = indices;
  p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.loss = loss;
  p.gw1 = gw1; p.gb1 = gb1; p.gw2 = gw2; p.gb2 = gb2;
  p.scratch = scratch;
  hipStream_t s = (hipStream_t)stream;
  const int HP = p.H == 20 ? 20 : kMlpMaxH;             // padded hidden width of the kernels' hot loops
  const int HS = HP | 1;
  const size_t lds_f = sizeof(float) * ((HP == 20 ? 0 : (size_t)p.n_in * HS) + (size_t)kMlpSPB * kMlpTPS * HP + (size_t)kMlpSPB * p.H +
                                        (size_t)kMlpSPB * p.O + (size_t)p.H * p.O + p.H + p.O);
  size_t lds_b = sizeof(float) * ((size_t)p.batch * HP + (size_t)p.batch + 4 * (size_t)kMlpKPB * HP);
  const size_t lds_small = sizeof(float) * (size_t)p.batch * (2 * p.H + p.O + 1);   // the small-tensor workgroup's copy of scratch
  if (lds_small > lds_b) lds_b = lds_small;
  if (lds_f > 160 * 1024 || lds_b > 160 * 1024)
    return fail(L2O_ERR_UNSUPPORTED, "l2o_mlp_fg: needs %zu / %zu bytes of LDS", lds_f, lds_b);
  if (p.H == kMlp20 && !opt(L2O_OPT_MLP_GENERIC)) {       // the reference's width: barrier-free forward, scalar-operand backward
    hipLaunchKernelGGL(k_mlp_fwd20, dim3(p.batch), dim3(64), 0, s, p);
    HIP_TRY(hipGetLastError());
    const int nkb20 = (p.n_in + 63) / 64;
    size_t lds20 = sizeof(float) * (size_t)kMlpBwdWaves * 64 * (kMlp20 + 1);
    if (lds_small > lds20) lds20 = lds_small;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp_bwd20), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds20));
    hipLaunchKernelGGL(k_mlp_bwd20, dim3(nkb20 + 1), dim3(64 * kMlpBwdWaves), lds20, s, p);
    HIP_TRY(hipGetLastError());
    return L2O_OK;
  }
  void (*ffwd)(MlpParams) = HP == 20 ? k_mlp_fwd<20> : k_mlp_fwd<kMlpMaxH>;
  void (*fbwd)(MlpParams) = HP == 20 ? k_mlp_bwd<20> : k_mlp_bwd<kMlpMaxH>;
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(ffwd), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds_f));
  hipLaunchKernelGGL(ffwd, dim3((p.batch + kMlpSPB - 1) / kMlpSPB), dim3(256), lds_f, s, p);
  HIP_TRY(hipGetLastError());
  const int nkb = (p.n_in + kMlpKPB - 1) / kMlpKPB;
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fbwd), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds_b));
  hipLaunchKernelGGL(fbwd, dim3(nkb + 1), dim3(256), lds_b, s, p);
  HIP_TRY(hipGetLastError());
  return L2O_OK;
}

size_t l2o_mlp_scratch_floats(const l2o_mlp* mlp) {
  if (!mlp || mlp->batch < 1) return 0;
  return (size_t)mlp->batch * (2 * (size_t)mlp->n_hidden + mlp->n_out + 1);
}

// ---- G = A^T B, the weight-gradient contraction of the meta-gradient (csrc/l2o_atb.h) ----------------------
static int atb_groups(int64_t R, int wgs_per_cu, hipStream_t s) {
  const int64_t nblk = (R + kAtbRows - 1) / kAtbRows;
  int g = wgs_per_cu * device_cu_count(s);                // persistent workgroups (k_atb: 72 KB of LDS each, k_atb_bx3: 52 | 58)
  if (g <= 0 || g > kAtbMaxGroups) g = kAtbMaxGroups;
  return (int)(nblk < g ? nblk : g);
}
size_t l2o_atb_workspace_bytes(int64_t R, int32_t KA, int32_t KB) {
  if (R <= 0 || KA <= 0 || KB <= 0) return 0;
  return sizeof(float) * (size_t)kAtbMaxGroups * KA * KB;
}
// mask 0: the dense product (fp32 matrix pipe, exact products).  mask 1 | 2: the weight-gradient blocks -- on the
// bf16 pipe (k_atb_bx3) unless the call asks for exact gates (L2O_OPT_EXACT_GATES: the fp32 pipe, bit-equal to the
// dense product on those blocks)
// compact_rows > 0 (mask 1 | 2, bf16 pipe only): A is the COMPACT operand of l2o_cwlstm_bwd_unroll_compact -- T + 1 blocks of
// compact_rows rows of KA - 40 floats -- and R = T * compact_rows rows of B (csrc/l2o_atb.h: atb_compact_off)
static int atb_launch(const float* A, const float* B, int64_t R, int KA, int KB, int mask, float* out, void* workspace,
                      hipStream_t s, int64_t compact_rows = 0, int P = 0) {
  float* part = static_cast<float*>(workspace);
  const int MT = (KA + 15) / 16, NT = (KB + 15) / 16;
  void (*fn)(const float*, const float*, long, int, int, float*) = nullptr;
  void (*fx)(const float*, const float*, long, int, int, float*, int, unsigned, int) = nullptr;
  int wgs = L2O_ATB_WGS_PER_CU;
  const bool bx3 = mask != 0 && !opt(L2O_OPT_EXACT_GATES);
  if (compact_rows > 0 && !bx3) return fail(L2O_ERR_UNSUPPORTED, "the compact A operand is read by the bf16x3 contraction only");
  if (mask == 1) {
    if (bx3) { fx = k_atb_bx3<6, 11, 1>; wgs = atb_bx3_wgs_per_cu<6, 11>(); } else fn = k_atb<6, 11, 1>;
  } else if (mask == 2) {
    if (bx3) { fx = k_atb_bx3<7, 12, 2>; wgs = atb_bx3_wgs_per_cu<7, 12>(); } else fn = k_atb<7, 12, 2>;
  } else if (MT <= 1 && NT <= 1) fn = k_atb<1, 1>;
  else if (MT <= 6 && NT <= 11) fn = k_atb<6, 11>;
  else fn = k_atb<7, 12>;
  const int groups = atb_groups(R, wgs, s);
  if (fx) {
    const int KAs = compact_rows > 0 ? KA - 2 * kH : KA;
    const uint64_t slice = compact_rows > 0 ? (uint64_t)compact_rows * KAs * sizeof(float) : 0;
    if (slice > 0xf0000000ull) return fail(L2O_ERR_UNSUPPORTED, "l2o_cwlstm_wgrad_compact: %lld rows per step is too many", (long long)compact_rows);
    hipLaunchKernelGGL(fx, dim3(groups), dim3(256), 0, s, A, B, (long)R, KA, KB, part, KAs, (unsigned)slice, P);
  } else {
    hipLaunchKernelGGL(fn, dim3(groups), dim3(256), 0, s, A, B, (long)R, KA, KB, part);
  }
  HIP_TRY(hipGetLastError());
  const int n = KA * KB;
  hipLaunchKernelGGL(k_atb_reduce, dim3((n + 31) / 32), dim3(256), 0, s, part, groups, n, out, mask,
                     mask == 2 ? 12 : 11, KB);
  HIP_TRY(hipGetLastError());
  return L2O_OK;
}

int l2o_atb(const float* A, const float* B, int64_t R, int32_t KA, int32_t KB, float* out, void* workspace, void* stream) {
  if (!A || !B || !out || !workspace || R <= 0 || KA <= 0 || KB <= 0) return fail(L2O_ERR_ARG, "l2o_atb: bad argument");
  if (KA > 112 || KB > 192) return fail(L2O_ERR_UNSUPPORTED, "l2o_atb: KA <= 112 and KB <= 192 (got %d x %d)", KA, KB);
  return atb_launch(A, B, R, KA, KB, 0, out, workspace, (hipStream_t)stream);
}

int32_t l2o_cwlstm_wgrad_dims(const l2o_net_cfg* cfg, int32_t* KA, int32_t* KB) {
  if (!cfg || !net_ok_for_mfma(cfg) || cfg->n_layers == 0) return fail(L2O_ERR_UNSUPPORTED, "l2o_cwlstm_wgrad: layers=(20,20) nets only");
  const bool fc = cfg->preprocess == L2O_PRE_FC_ELU;
  const int P = fc ? kH : (cfg->preprocess == L2O_PRE_LOGSIGN ? 2 : 1);
  if (KA) *KA = P + kH + 3 * kH + (fc ? 2 : 0) + 1;         // act1 | act2 | h2 | feats | 1   (BwdTileGeom::KA)
  if (KB) *KB = 8 * kH + 1 + (fc ? kH : 0);                 // dz1 | dz2 | dd | du             (BwdTileGeom::KB)
  return L2O_OK;
}

int l2o_cwlstm_wgrad(const l2o_net_cfg* cfg, const float* A, const float* Bm, int64_t R, float* G, void* workspace,
                     void* stream) {
  OptScope opt_scope(cfg_optw(cfg));
  int32_t KA = 0, KB = 0;
  const int rc = l2o_cwlstm_wgrad_dims(cfg, &KA, &KB);
  if (rc) return rc;
  if (!A || !Bm || !G || !workspace || R <= 0) return fail(L2O_ERR_ARG, "l2o_cwlstm_wgrad: bad argument");
  return atb_launch(A, Bm, R, KA, KB, cfg->preprocess == L2O_PRE_FC_ELU ? 2 : 1, G, workspace, (hipStream_t)stream);
}

int l2o_cwlstm_wgrad_compact(const l2o_net_cfg* cfg, const float* Ac, const float* Bm, int32_t T, int64_t rows, float* G,
                             void* workspace, void* stream) {
  OptScope opt_scope(cfg_optw(cfg));
  int32_t KA = 0, KB = 0;
  const int rc = l2o_cwlstm_wgrad_dims(cfg, &KA, &KB);
  if (rc) return rc;
  if (!Ac || !Bm || !G || !workspace || T <= 0 || rows <= 0) return fail(L2O_ERR_ARG, "l2o_cwlstm_wgrad_compact: bad argument");
  const bool fc = cfg->preprocess == L2O_PRE_FC_ELU;
  const int P = fc ? kH : (cfg->preprocess == L2O_PRE_LOGSIGN ? 2 : 1);
  return atb_launch(Ac, Bm, (int64_t)T * rows, KA, KB, fc ? 2 : 1, G, workspace, (hipStream_t)stream, rows, P);
}

// ---- generic-`layers` optimizer step (csrc/l2o_generic.h) --------------------------------------------------
static int check_gen_net(const l2o_net_cfg* cfg, const l2o_gen_net* net) {
  if (!cfg || !net) return fail(L2O_ERR_ARG, "generic net: NULL argument");
  if (net->n_layers < 1 || net->n_layers > kGenMaxL)
    return fail(L2O_ERR_UNSUPPORTED, "generic net: 1..%d LSTM layers (got %d)", kGenMaxL, net->n_layers);
  for (int l = 0; l < net->n_layers; ++l) {
    if (net->hidden[l] < 1 || net->hidden[l] > kGenMaxH)
      return fail(L2O_ERR_UNSUPPORTED, "generic net: hidden sizes 1..%d (layer %d has %d)", kGenMaxH, l, net->hidden[l]);
    if (!net->w_gates[l] || !net->b_gates[l]) return fail(L2O_ERR_ARG, "generic net: NULL weights of layer %d", l);
  }
  if (!net->w_lin || !net->b_lin) return fail(L2O_ERR_ARG, "generic net: NULL output Linear");
  const int want = cfg->preprocess == L2O_PRE_FC_ELU ? net->in_dim : (cfg->preprocess == L2O_PRE_LOGSIGN ? 2 : 1);
  if (net->in_dim != want || net->in_dim < 1 || net->in_dim > kGenMaxH)
    return fail(L2O_ERR_ARG, "generic net: in_dim %d does not match the preprocessing (%d)", net->in_dim, want);
  if (cfg->preprocess == L2O_PRE_FC_ELU && (!net->w_fc || !net->b_fc)) return fail(L2O_ERR_ARG, "generic net: fc weights");
  return L2O_OK;
}

size_t l2o_gen_state_floats(const l2o_gen_net* net, int64_t N) {
  if (!net || N <= 0 || net->n_layers < 1 || net->n_layers > kGenMaxL) return 0;
  size_t n = 0;
  for (int l = 0; l < net->n_layers; ++l) n += 2 * (size_t)N * (size_t)net->hidden[l];
  return n;
}

int l2o_cwlstm_step_generic(const l2o_net_cfg* cfg, const l2o_gen_net* net, const float* g, const float* m_tilde, float* m,
                            float* v, double pow1, double pow2, float* state, float* x, int64_t N, void* stream) {
  OptScope opt_scope(cfg_optw(cfg));
  int rc = check_gen_net(cfg, net);
  if (rc) return rc;
  if (!g || !state || !x || N <= 0) return fail(L2O_ERR_ARG, "l2o_cwlstm_step_generic: bad argument");
  const bool fc = cfg->preprocess == L2O_PRE_FC_ELU;
  if (fc && net->direct_inputs && !m_tilde) return fail(L2O_ERR_ARG, "l2o_cwlstm_step_generic: direct inputs need m_tilde");
  if (fc && !net->direct_inputs && (!m || !v)) return fail(L2O_ERR_ARG, "l2o_cwlstm_step_generic: RNNProp needs m and v");
  GenParams p;
  std::memset(&p, 0, sizeof(p));
  p.n_layers = net->n_layers;
  for (int l = 0; l < net->n_layers; ++l) { p.H[l] = net->hidden[l]; p.wg[l] = net->w_gates[l]; p.bg[l] = net->b_gates[l]; }
  p.in_dim = net->in_dim; p.pre = cfg->preprocess; p.direct = net->direct_inputs; p.tanh_output = cfg->tanh_output;
  p.scale = (float)cfg->scale;
  p.k_inv = cfg->logsign_k != 0.0 ? (float)(1.0 / cfg->logsign_k) : 0.0f;
  p.exp_k = (float)std::exp(cfg->logsign_k);
  p.beta1 = (float)cfg->beta1; p.beta2 = (float)cfg->beta2;
  p.omb1 = (float)(1.0 - cfg->beta1); p.omb2 = (float)(1.0 - cfg->beta2);
  p.om1 = (float)(1.0 - pow1); p.om2 = (float)(1.0 - pow2);
  p.wl = net->w_lin; p.bl = net->b_lin; p.wfc = net->w_fc; p.bfc = net->b_fc;
  p.g = g; p.m_in = m_tilde; p.m = m; p.v = v; p.state = state; p.x = x; p.N = (long)N;
  hipLaunchKernelGGL(k_cwlstm_generic, dim3((unsigned)((N + kGenThreads - 1) / kGenThreads)), dim3(kGenThreads), 0,
                     (hipStream_t)stream, p);
  HIP_TRY(hipGetLastError());
  return L2O_OK;
}

int l2o_cwlstm_bwd_step_generic(const l2o_net_cfg* cfg, const l2o_gen_net* net, const l2o_gen_bwd_io* io, double pow1,
                                double pow2, int64_t N, void* stream) {
  OptScope opt_scope(cfg_optw(cfg));
  int rc = check_gen_net(cfg, net);
  if (rc) return rc;
  if (!io || N <= 0 || !io->g || !io->st_prev || !io->dx_next || !io->carry_in || !io->carry_out || !io->tc ||
      !io->h_last || !io->dd)
    return fail(L2O_ERR_ARG, "l2o_cwlstm_bwd_step_generic: bad argument");
  if (net->direct_inputs) return fail(L2O_ERR_UNSUPPORTED, "l2o_cwlstm_bwd_step_generic: direct inputs have no moments to chain through");
  const bool fc = cfg->preprocess == L2O_PRE_FC_ELU;
  if (fc && (!io->m || !io->v || !io->feats || !io->du))
    return fail(L2O_ERR_ARG, "l2o_cwlstm_bwd_step_generic: RNNProp needs m, v, feats and du");
  if (fc && io->dg) return fail(L2O_ERR_UNSUPPORTED, "l2o_cwlstm_bwd_step_generic: the input adjoint of RNNProp is formed by the host from du");
  GenBwdParams p;
  std::memset(&p, 0, sizeof(p));
  p.n_layers = net->n_layers;
  for (int l = 0; l < net->n_layers; ++l) {
    if (!io->act[l] || !io->dz[l]) return fail(L2O_ERR_ARG, "l2o_cwlstm_bwd_step_generic: NULL act / dz of layer %d", l);
    p.H[l] = net->hidden[l]; p.wg[l] = net->w_gates[l]; p.bg[l] = net->b_gates[l];
    p.act[l] = io->act[l]; p.dz[l] = io->dz[l];
  }
  p.in_dim = net->in_dim; p.pre = cfg->preprocess; p.tanh_output = cfg->tanh_output;
  p.scale = (float)cfg->scale;
  p.k_inv = cfg->logsign_k != 0.0 ? (float)(1.0 / cfg->logsign_k) : 0.0f;
  p.exp_k = (float)std::exp(cfg->logsign_k);
  p.om1 = (float)(1.0 - pow1); p.om2 = (float)(1.0 - pow2);
  p.wl = net->w_lin; p.bl = net->b_lin; p.wfc = net->w_fc; p.bfc = net->b_fc;
  p.g = io->g; p.m = io->m; p.v = io->v; p.st_prev = io->st_prev; p.dx_next = io->dx_next;
  p.carry_in = io->carry_in; p.carry_out = io->carry_out; p.tc = io->tc; p.h_last = io->h_last; p.dd = io->dd;
  p.feats = io->feats; p.du = io->du; p.dg = io->dg; p.N = (long)N;
  hipLaunchKernelGGL(k_cwlstm_generic_bwd, dim3((unsigned)((N + kGenThreads - 1) / kGenThreads)), dim3(kGenThreads), 0,
                     (hipStream_t)stream, p);
  HIP_TRY(hipGetLastError());
  return L2O_OK;
}

// ---- the fused persistent unroll of the MLP optimizee (csrc/l2o_mlp_unroll.h) ---------------------------
struct MlpUnrollLayout { int n[4], tile_begin[5], nwg, nw1, R, hier_R1; bool fast; size_t NO, NSM, p_off, s_off, sm_off, hs_off, x_off, s1_off, total; };
static bool mlp_unroll_layout(const l2o_mlp* mlp, MlpUnrollLayout* L) {
  if (!mlp) return false;
  const int H = mlp->n_hidden, O = mlp->n_out;
  if (H < kMuMinH || H > kMuMaxH || O < 1 || O > kMuMaxO || mlp->batch < 1 || mlp->batch > kMuMaxBatch || mlp->n_in < 1)
    return false;
  L->n[0] = mlp->n_in * H; L->n[1] = H; L->n[2] = H * O; L->n[3] = O;
  if (L->n[0] % 64) return false;                        // the w1 coordinates fill whole workgroups
  L->tile_begin[0] = 0;
  for (int v = 0; v < 4; ++v) L->tile_begin[v + 1] = L->tile_begin[v] + tiles_per_problem(L->n[v]);
  L->nwg = (L->tile_begin[4] + 3) / 4;
  L->nw1 = L->n[0] / 64;
  L->NO = (size_t)mlp->batch * H;
  L->NSM = (size_t)H + (size_t)H * O + O;
  L->R = (int)((L->NO + L->nwg - 1) / L->nwg);
  L->R = (L->R + 1) & ~1;                                // even: the fast path moves granules in pairs
  // one 64-byte line per (source, reducer): R = 8 measured best (profiles/archive_r01_r03/r02l: R = 6 / 8 / 12 / 16 -> 1.32 / 1.35 /
  // 1.28 / 1.24 G on config 5)
  if (L->R < 8 && L->NO >= 8 * 32) L->R = 8;
  if (L->R > kMuMaxR) return false;
  L->fast = H == 20 && O == 10 && mlp->batch == 64;
  L->p_off = sizeof(MlpWs);
  // XCD-hierarchical all-reduce (fast path; l2o_mlp_unroll.h): group g = wg % 8, every group needs enough w1 owners to
  // reduce all NO outputs in slices of R1 (even) <= kMuHierR1Max
  L->hier_R1 = 0;
  if (L->fast && opt(L2O_OPT_MLP_HIER) && L->nwg <= 256 && L->nw1 >= 2 * kMuHierG && (L->nwg + kMuHierG - 1) / kMuHierG <= kMuHierM) {
    const int min_cnt = L->nw1 / kMuHierG;                 // the smallest group
    int r1 = (int)((L->NO + min_cnt - 1) / min_cnt);
    r1 = (r1 + 1) & ~1;
    if (r1 <= kMuHierR1Max) L->hier_R1 = r1;
  }
  // P: generic [nw1][NO]; fast path: one inbox per reducing workgroup, [nwg][nw1][R]; hierarchical: [8][M][M][R1]
  const size_t pg = (size_t)L->nw1 * L->NO, pf = (size_t)L->nwg * L->nw1 * L->R;
  const size_t ph = L->hier_R1 ? (size_t)kMuHierG * kMuHierM * kMuHierM * L->hier_R1 : 0;
  size_t pmax = pg > pf ? pg : pf;
  if (ph > pmax) pmax = ph;
  L->s_off = L->p_off + sizeof(unsigned long long) * pmax;
  L->sm_off = L->s_off + sizeof(unsigned long long) * 2 * L->NO;
  L->hs_off = L->sm_off + sizeof(unsigned long long) * 2 * ((L->NSM + 1) & ~(size_t)1);
  L->x_off = L->hs_off + sizeof(unsigned long long) * 256;
  L->s1_off = L->x_off + sizeof(unsigned long long) * 2 * kMuHierG * kMuHierM * kMuHierR1Max;
  L->total = L->s1_off + sizeof(unsigned long long) * 2 * kMuHierG * kMuHierS;
  return true;
}

int l2o_mlp_unroll_supported(const l2o_net_cfg* cfg, const l2o_mlp* mlp, void* stream) {
  OptScope opt_scope(cfg_optw(cfg));
  MlpUnrollLayout L;
  if (!cfg || !net_ok_for_mfma(cfg) || !opt(L2O_OPT_MLP_UNROLL) || !mlp_unroll_layout(mlp, &L)) return 0;
  if (L.nwg > coresident_cus((hipStream_t)stream)) return 0;     // one workgroup per CU, all co-resident
  return L.fast ? 2 : 1;                                           // 2: the reference's shape (static loop bounds, paired granules)
}

size_t l2o_mlp_unroll_workspace_bytes(const l2o_mlp* mlp) {
  MlpUnrollLayout L;
  return mlp_unroll_layout(mlp, &L) ? L.total : 0;
}

static int mlp_unroll_launch(const l2o_net_cfg* cfg, const float* wpack, const l2o_mlp* mlp, const int32_t* indices,
                             float* const* x, float* const* st, float* const* m, float* const* v,
                             const float* const* x_scale, int32_t T, int32_t step0, float* fx, const l2o_mlp_hist* hist,
                             void* workspace, void* stream) {
  OptScope opt_scope(cfg_optw(cfg));
  if (!cfg || !wpack || !mlp || !indices || !x || !st || !fx || !workspace || T < 0 || !mlp->images || !mlp->labels)
    return fail(L2O_ERR_ARG, "l2o_mlp_unroll: bad argument");
  hipStream_t s = (hipStream_t)stream;
  MlpUnrollLayout L;
  if (!net_ok_for_mfma(cfg) || !mlp_unroll_layout(mlp, &L) || L.nwg > coresident_cus(s))
    return fail(L2O_ERR_UNSUPPORTED, "l2o_mlp_unroll: no fused kernel for n_in=%d hidden=%d out=%d batch=%d net(layers=%d)",
                mlp->n_in, mlp->n_hidden, mlp->n_out, mlp->batch, cfg->n_layers);
  const bool rn = cfg->preprocess == L2O_PRE_FC_ELU;
  MlpUnrollArgs a;
  std::memset(&a, 0, sizeof(a));
  a.np = make_net_params(cfg, wpack);
  a.n_in = mlp->n_in; a.H = mlp->n_hidden; a.O = mlp->n_out; a.batch = mlp->batch; a.act = mlp->activation;
  a.images = mlp->images; a.labels = mlp->labels; a.idx = indices;
  for (int k = 0; k < 4; ++k) {
    if (!x[k] || !st[k] || (rn && (!m || !v || !m[k] || !v[k]))) return fail(L2O_ERR_ARG, "l2o_mlp_unroll: NULL buffer of variable %d", k);
    a.x[k] = x[k]; a.st[k] = st[k]; a.m[k] = rn ? m[k] : nullptr; a.v[k] = rn ? v[k] : nullptr;
    a.xscale[k] = x_scale ? x_scale[k] : nullptr;
    a.n[k] = L.n[k];
  }
  for (int k = 0; k < 5; ++k) a.tile_begin[k] = L.tile_begin[k];
  a.T = T;
  pow_ff(cfg->beta1, step0, &a.p1_hi, &a.p1_lo);
  pow_ff(cfg->beta2, step0, &a.p2_hi, &a.p2_lo);
  a.fx = fx;
  char* wsb = static_cast<char*>(workspace);
  a.ws = reinterpret_cast<MlpWs*>(wsb);
  a.P = reinterpret_cast<unsigned long long*>(wsb + L.p_off);
  a.S = reinterpret_cast<unsigned long long*>(wsb + L.s_off);
  a.Sm = reinterpret_cast<unsigned long long*>(wsb + L.sm_off);
  a.nwg = L.nwg; a.nw1 = L.nw1; a.R = L.R;
  a.hier_R1 = L.hier_R1;
  a.HS = reinterpret_cast<unsigned long long*>(wsb + L.hs_off);
  a.X = reinterpret_cast<unsigned long long*>(wsb + L.x_off);
  a.S1 = reinterpret_cast<unsigned long long*>(wsb + L.s1_off);
  a.use_salt = T + 1 < 0xffff ? 1u : 0u;
  if (hist) {
    for (int k = 0; k < 4; ++k) {
      if (!hist->st[k] || !hist->g[k] || (rn && (!hist->m[k] || !hist->v[k])))
        return fail(L2O_ERR_ARG, "l2o_mlp_unroll_record: NULL history buffer of variable %d", k);
      a.hist_st[k] = hist->st[k]; a.hist_g[k] = hist->g[k];
      a.hist_m[k] = rn ? hist->m[k] : nullptr; a.hist_v[k] = rn ? hist->v[k] : nullptr;
    }
  }
  HIP_TRY(hipMemsetAsync(wsb + L.p_off, 0, L.total - L.p_off, s));   // the granules only: the header survives
  const dim3 grid(L.nwg), block(256);
  void (*fn)(MlpUnrollArgs) = nullptr;
  if (hist) {
    switch (cfg->preprocess) {
      case L2O_PRE_IDENTITY: fn = L.fast ? k_mlp_unroll<L2O_PRE_IDENTITY, true, true> : k_mlp_unroll<L2O_PRE_IDENTITY, false, true>; break;
      case L2O_PRE_LOGSIGN: fn = L.fast ? k_mlp_unroll<L2O_PRE_LOGSIGN, true, true> : k_mlp_unroll<L2O_PRE_LOGSIGN, false, true>; break;
      default: fn = L.fast ? k_mlp_unroll<L2O_PRE_FC_ELU, true, true> : k_mlp_unroll<L2O_PRE_FC_ELU, false, true>;
    }
  } else {
    switch (cfg->preprocess) {
      case L2O_PRE_IDENTITY: fn = L.fast ? k_mlp_unroll<L2O_PRE_IDENTITY, true> : k_mlp_unroll<L2O_PRE_IDENTITY, false>; break;
      case L2O_PRE_LOGSIGN: fn = L.fast ? k_mlp_unroll<L2O_PRE_LOGSIGN, true> : k_mlp_unroll<L2O_PRE_LOGSIGN, false>; break;
      default: fn = L.fast ? k_mlp_unroll<L2O_PRE_FC_ELU, true> : k_mlp_unroll<L2O_PRE_FC_ELU, false>;
    }
  }
  hipLaunchKernelGGL(fn, grid, block, 0, s, a);
  HIP_TRY(hipGetLastError());
  note_form(!L.fast ? L2O_FORM_MLP_UNROLL_GENERIC : (a.hier_R1 ? L2O_FORM_MLP_UNROLL_HIER : L2O_FORM_MLP_UNROLL));
  return L2O_OK;
}

int l2o_mlp_unroll(const l2o_net_cfg* cfg, const float* wpack, const l2o_mlp* mlp, const int32_t* indices,
                   float* const* x, float* const* st, float* const* m, float* const* v, const float* const* x_scale,
                   int32_t T, int32_t step0, float* fx, void* workspace, void* stream) {
  return mlp_unroll_launch(cfg, wpack, mlp, indices, x, st, m, v, x_scale, T, step0, fx, nullptr, workspace, stream);
}
int l2o_mlp_unroll_record(const l2o_net_cfg* cfg, const float* wpack, const l2o_mlp* mlp, const int32_t* indices,
                          float* const* x, float* const* st, float* const* m, float* const* v,
                          const float* const* x_scale, int32_t T, int32_t step0, float* fx, const l2o_mlp_hist* hist,
                          void* workspace, void* stream) {
  if (!hist) return fail(L2O_ERR_ARG, "l2o_mlp_unroll_record: NULL hist");
  return mlp_unroll_launch(cfg, wpack, mlp, indices, x, st, m, v, x_scale, T, step0, fx, hist, workspace, stream);
}

// ---- problems.mnist with several hidden layers (csrc/l2o_mlp_deep.h) ---------------------------------------------------
static bool mlp_deep_ok(const l2o_mlp_deep* m) {
  if (!m || m->n_hidden_layers < 1 || m->n_hidden_layers > kMdMaxHidden || m->n_in < 1 || m->n_in > kMdMaxIn) return false;
  if (m->n_out < 1 || m->n_out > kMdMaxOut || m->batch < 1 || m->batch > 4096) return false;
  for (int l = 0; l < m->n_hidden_layers; ++l)
    if (m->hidden[l] < 1 || m->hidden[l] > kMdMaxWidth) return false;
  return true;
}
size_t l2o_mlp_deep_scratch_floats(const l2o_mlp_deep* m) {
  if (!mlp_deep_ok(m)) return 0;
  return (size_t)m->batch * kMdMaxWidth * (2 * m->n_hidden_layers + 1) + m->batch;
}
int l2o_mlp_deep_fg(const l2o_mlp_deep* m, const int32_t* indices, const float* const* w, float* loss, float* const* g,
                    float* scratch, void* stream) {
  if (!mlp_deep_ok(m))
    return fail(L2O_ERR_UNSUPPORTED, "l2o_mlp_deep_fg: 1-3 hidden layers of <= 32 units, n_in <= 1024, n_out <= 16, batch <= 4096");
  if (!indices || !w || !loss || !scratch || !m->images || !m->labels) return fail(L2O_ERR_ARG, "l2o_mlp_deep_fg: NULL argument");
  MlpDeepArgs a;
  std::memset(&a, 0, sizeof(a));
  a.n_in = m->n_in; a.n_out = m->n_out; a.batch = m->batch; a.act = m->activation; a.nh = m->n_hidden_layers;
  for (int l = 0; l < a.nh; ++l) a.width[l] = m->hidden[l];
  a.width[a.nh] = m->n_out;
  a.images = m->images; a.labels = m->labels; a.idx = indices;
  long ncoord = 0;
  for (int l = 0; l <= a.nh; ++l) {
    if (!w[2 * l] || !w[2 * l + 1] || (g && (!g[2 * l] || !g[2 * l + 1]))) return fail(L2O_ERR_ARG, "l2o_mlp_deep_fg: NULL buffer of layer %d", l);
    a.w[l] = w[2 * l]; a.b[l] = w[2 * l + 1];
    if (g) { a.gw[l] = g[2 * l]; a.gb[l] = g[2 * l + 1]; }
    ncoord += (long)(l == 0 ? a.n_in : a.width[l - 1]) * a.width[l] + a.width[l];
  }
  a.acts = scratch;
  a.deltas = a.acts + (size_t)a.nh * a.batch * kMdMaxWidth;
  a.loss_s = a.deltas + (size_t)(a.nh + 1) * a.batch * kMdMaxWidth;
  a.loss = loss;
  a.want_grad = g ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_mlp_deep_sample, dim3(a.batch), dim3(kMdThreads), 0, s, a);
  const long nthreads = g ? ncoord : 1;
  hipLaunchKernelGGL(k_mlp_deep_grad, dim3((unsigned)((nthreads + kMdThreads - 1) / kMdThreads)), dim3(kMdThreads), 0, s, a);
  HIP_TRY(hipGetLastError());
  return L2O_OK;
}

#ifndef L2O_MLP_XCD_DEFAULT_FOUR
#define L2O_MLP_XCD_DEFAULT_FOUR 0     // (set from the A/B measurement: profiles/r06_c5_xcd_forms_ab.txt)
#endif
// ---- one optimizee instance per XCD (csrc/l2o_mlp_xcd.h) ------------------------------------------------------------
struct MlpXcdLayout { int n[4], tile_begin[5], nw1; size_t team_off, inst_off, p_bytes, s_bytes, sm_bytes, inst_bytes, total; };
static bool mlp_xcd_layout(const l2o_mlp* mlp, int n_inst, MlpXcdLayout* L) {
  if (!mlp || n_inst < 1 || n_inst > kMxMaxInst) return false;
  if (mlp->n_hidden != kMxH || mlp->n_out != kMxO || mlp->batch != kMxB || mlp->n_in < 1) return false;
  L->n[0] = mlp->n_in * kMxH; L->n[1] = kMxH; L->n[2] = kMxH * kMxO; L->n[3] = kMxO;
  L->tile_begin[0] = 0;
  for (int v = 0; v < 4; ++v) L->tile_begin[v + 1] = L->tile_begin[v] + tiles_per_problem(L->n[v]);
  // members 0 .. 30: the w1 tiles (32 each); member 31: b1 | w2 | b2
  if (L->tile_begin[1] > (kMxMembers - 1) * kMxSlots || L->tile_begin[4] - L->tile_begin[1] > kMxSlots) return false;
  L->nw1 = (L->n[0] + kMxCoords - 1) / kMxCoords;
  L->team_off = sizeof(MlpWs);
  L->inst_off = L->team_off + 64;
  L->p_bytes = sizeof(unsigned long long) * kMxMembers * kMxMembers * kMxR;
  L->s_bytes = sizeof(unsigned long long) * 2 * kMxNO;
  L->sm_bytes = sizeof(unsigned long long) * 2 * kMxNSMp;
  L->inst_bytes = (L->p_bytes + L->s_bytes + L->sm_bytes + 255) & ~(size_t)255;
  L->total = L->inst_off + (size_t)n_inst * L->inst_bytes;
  return true;
}

int l2o_mlp_unroll_multi_supported(const l2o_net_cfg* cfg, const l2o_mlp* mlp, int32_t n_inst, void* stream) {
  OptScope opt_scope(cfg_optw(cfg));
  MlpXcdLayout L;
  if (!cfg || !net_ok_for_mfma(cfg) || !opt(L2O_OPT_MLP_UNROLL) || !mlp_xcd_layout(mlp, n_inst, &L)) return 0;
  // every XCD's 32 CUs must be able to host one workgroup each at the same time
  return coresident_cus((hipStream_t)stream) >= kMxMaxInst * kMxMembers ? 1 : 0;
}

size_t l2o_mlp_unroll_multi_workspace_bytes(const l2o_mlp* mlp, int32_t n_inst) {
  MlpXcdLayout L;
  return mlp_xcd_layout(mlp, n_inst, &L) ? L.total : 0;
}

int l2o_mlp_unroll_multi(const l2o_net_cfg* cfg, const float* wpack, const l2o_mlp* mlp, const l2o_mlp_instance* inst,
                         int32_t n_inst, int32_t T, int32_t step0, void* workspace, void* stream) {
  OptScope opt_scope(cfg_optw(cfg));
  if (!cfg || !wpack || !mlp || !inst || !workspace || T < 0 || !mlp->images || !mlp->labels)
    return fail(L2O_ERR_ARG, "l2o_mlp_unroll_multi: bad argument");
  hipStream_t s = (hipStream_t)stream;
  MlpXcdLayout L;
  if (!net_ok_for_mfma(cfg) || !mlp_xcd_layout(mlp, n_inst, &L) || coresident_cus(s) < kMxMaxInst * kMxMembers)
    return fail(L2O_ERR_UNSUPPORTED, "l2o_mlp_unroll_multi: no one-XCD kernel for n_in=%d hidden=%d out=%d batch=%d x %d instances "
                "(needs the reference's shape and all 8 x 32 CUs)", mlp->n_in, mlp->n_hidden, mlp->n_out, mlp->batch, (int)n_inst);
  const bool rn = cfg->preprocess == L2O_PRE_FC_ELU;
  MlpXcdArgs a;
  std::memset(&a, 0, sizeof(a));
  a.np = make_net_params(cfg, wpack);
  a.n_in = mlp->n_in; a.act = mlp->activation; a.T = T; a.ninst = n_inst;
  a.images = mlp->images; a.labels = mlp->labels;
  for (int k = 0; k < 4; ++k) a.n[k] = L.n[k];
  for (int k = 0; k < 5; ++k) a.tile_begin[k] = L.tile_begin[k];
  a.nw1 = L.nw1;
  pow_ff(cfg->beta1, step0, &a.p1_hi, &a.p1_lo);
  pow_ff(cfg->beta2, step0, &a.p2_hi, &a.p2_lo);
  char* wsb = static_cast<char*>(workspace);
  a.ws = reinterpret_cast<MlpWs*>(wsb);
  a.team = reinterpret_cast<unsigned*>(wsb + L.team_off);
  for (int j = 0; j < n_inst; ++j) {
    const l2o_mlp_instance& in = inst[j];
    MxInst& o = a.inst[j];
    if (!in.indices || !in.fx) return fail(L2O_ERR_ARG, "l2o_mlp_unroll_multi: NULL indices / fx of instance %d", j);
    o.idx = in.indices; o.fx = in.fx;
    for (int k = 0; k < 4; ++k) {
      if (!in.x[k] || !in.st[k] || (rn && (!in.m[k] || !in.v[k])))
        return fail(L2O_ERR_ARG, "l2o_mlp_unroll_multi: NULL buffer of variable %d of instance %d", k, j);
      o.x[k] = in.x[k]; o.st[k] = in.st[k]; o.m[k] = rn ? in.m[k] : nullptr; o.v[k] = rn ? in.v[k] : nullptr;
      o.xscale[k] = in.x_scale[k];
    }
    char* ib = wsb + L.inst_off + (size_t)j * L.inst_bytes;
    o.P = reinterpret_cast<unsigned long long*>(ib);
    o.S = reinterpret_cast<unsigned long long*>(ib + L.p_bytes);
    o.Sm = reinterpret_cast<unsigned long long*>(ib + L.p_bytes + L.s_bytes);
  }
  HIP_TRY(hipMemsetAsync(wsb + L.team_off, 0, L.total - L.team_off, s));   // team counters + granules: the header survives
  // which form: four waves per member stepping tile PAIRS (a lone wave per SIMD with two independent chains; RNNProp: no
  // spills at 438 registers) or eight waves stepping single tiles (two waves per SIMD; the DM nets' pair form spills)
  const int wopt = (int)opt(L2O_OPT_MLP_XCD_WAVES);
  const bool four = wopt == 2 || (wopt == 0 && rn && L2O_MLP_XCD_DEFAULT_FOUR);
  void (*fn)(MlpXcdArgs) = nullptr;
  size_t lds = 0;
  switch (cfg->preprocess) {
    case L2O_PRE_IDENTITY: fn = four ? k_mlp_xcd<L2O_PRE_IDENTITY, 4> : k_mlp_xcd<L2O_PRE_IDENTITY, 8>; lds = mlp_xcd_lds_bytes<L2O_PRE_IDENTITY>(); break;
    case L2O_PRE_LOGSIGN: fn = four ? k_mlp_xcd<L2O_PRE_LOGSIGN, 4> : k_mlp_xcd<L2O_PRE_LOGSIGN, 8>; lds = mlp_xcd_lds_bytes<L2O_PRE_LOGSIGN>(); break;
    default: fn = four ? k_mlp_xcd<L2O_PRE_FC_ELU, 4> : k_mlp_xcd<L2O_PRE_FC_ELU, 8>; lds = mlp_xcd_lds_bytes<L2O_PRE_FC_ELU>();
  }
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // one workgroup per CU of the whole chip: the 32 that land on XCD j < n_inst form instance j's team, the others exit
  hipLaunchKernelGGL(fn, dim3(kMxMaxInst * kMxMembers), dim3(four ? kMxThreads4 : kMxThreads), lds, s, a);
  HIP_TRY(hipGetLastError());
  note_form(L2O_FORM_MLP_XCD);
  return L2O_OK;
}

int l2o_cwlstm_step_multi(const l2o_net_cfg* cfg, const float* wpack, const l2o_step_seg* segs, int32_t nseg,
                          double pow1, double pow2, void* stream) {
  OptScope opt_scope(cfg_optw(cfg));
  if (!cfg || !wpack || !segs || nseg < 1) return fail(L2O_ERR_ARG, "l2o_cwlstm_step: bad argument");
  if (nseg > kMaxStepSegs) return fail(L2O_ERR_UNSUPPORTED, "l2o_cwlstm_step_multi: at most %d segments", kMaxStepSegs);
  for (int i = 0; i < nseg; ++i)
    if (!segs[i].g || !segs[i].x || segs[i].B <= 0 || segs[i].D <= 0)
      return fail(L2O_ERR_ARG, "l2o_cwlstm_step: bad argument (segment %d)", i);
  const NetParams np = make_net_params(cfg, wpack);
  hipStream_t s = (hipStream_t)stream;
  if (cfg->n_layers == 0) {
    if (cfg->kind != L2O_NET_CW || cfg->preprocess == L2O_PRE_FC_ELU)
      return fail(L2O_ERR_UNSUPPORTED, "layers=() is implemented for CoordinateWiseDeepLSTM only");
    for (int i = 0; i < nseg; ++i) {
      const size_t n = (size_t)segs[i].B * segs[i].D;
      const dim3 grid((unsigned)((n + 255) / 256));
      if (cfg->preprocess == L2O_PRE_LOGSIGN)
        hipLaunchKernelGGL(k_linear_step<L2O_PRE_LOGSIGN>, grid, dim3(256), 0, s, np, segs[i].g, segs[i].x, n);
      else
        hipLaunchKernelGGL(k_linear_step<L2O_PRE_IDENTITY>, grid, dim3(256), 0, s, np, segs[i].g, segs[i].x, n);
    }
    HIP_TRY(hipGetLastError());
    return L2O_OK;
  }
  if (!net_ok_for_mfma(cfg)) return fail(L2O_ERR_UNSUPPORTED, "l2o_cwlstm_step: only layers=(20,20) / () nets");
  StepSegs sg;
  std::memset(&sg, 0, sizeof(sg));
  sg.n = nseg;
  int64_t ntiles = 0;
  for (int i = 0; i < nseg; ++i) {
    if (!segs[i].st) return fail(L2O_ERR_ARG, "l2o_cwlstm_step: NULL state");
    if (cfg->preprocess == L2O_PRE_FC_ELU && (!segs[i].m || !segs[i].v))
      return fail(L2O_ERR_ARG, "l2o_cwlstm_step: RNNProp needs m and v");
    sg.tpp[i] = tiles_per_problem(segs[i].D);
    sg.D[i] = (int)segs[i].D;
    ntiles += segs[i].B * sg.tpp[i];
    if (ntiles > INT32_MAX) return fail(L2O_ERR_UNSUPPORTED, "l2o_cwlstm_step: too many coordinates");
    sg.tile_end[i] = (int)ntiles;
    sg.g[i] = segs[i].g; sg.m[i] = segs[i].m; sg.v[i] = segs[i].v; sg.st[i] = segs[i].st; sg.x[i] = segs[i].x;
    sg.st_out[i] = segs[i].st_out ? segs[i].st_out : segs[i].st;
    sg.m_out[i] = segs[i].m_out ? segs[i].m_out : segs[i].m;
    sg.v_out[i] = segs[i].v_out ? segs[i].v_out : segs[i].v;
  }
  int blocks = (int)((ntiles + 3) / 4);
  if (blocks > 256) blocks = 256;            // one 4-wave block per CU (one wave per SIMD), grid-stride beyond
  const dim3 grid(blocks), block(256);
  const float om1 = (float)(1.0 - pow1), om2 = (float)(1.0 - pow2);
  switch (cfg->preprocess) {
    case L2O_PRE_IDENTITY: hipLaunchKernelGGL(k_cwlstm_step<L2O_PRE_IDENTITY>, grid, block, 0, s, np, sg, om1, om2); break;
    case L2O_PRE_LOGSIGN: hipLaunchKernelGGL(k_cwlstm_step<L2O_PRE_LOGSIGN>, grid, block, 0, s, np, sg, om1, om2); break;
    default: hipLaunchKernelGGL(k_cwlstm_step<L2O_PRE_FC_ELU>, grid, block, 0, s, np, sg, om1, om2);
  }
  HIP_TRY(hipGetLastError());
  return L2O_OK;
}

int l2o_cwlstm_step(const l2o_net_cfg* cfg, const float* wpack, const float* g, float* m, float* v, double pow1,
                    double pow2, float* st, float* x, int64_t B, int64_t D, void* stream) {
  OptScope opt_scope(cfg_optw(cfg));
  l2o_step_seg seg;
  std::memset(&seg, 0, sizeof(seg));
  seg.g = g; seg.m = m; seg.v = v; seg.st = st; seg.x = x; seg.B = B; seg.D = D;
  return l2o_cwlstm_step_multi(cfg, wpack, &seg, 1, pow1, pow2, stream);
}

static int launch_bwd_tile(const BwdParams& p, int pre, hipStream_t s) {
  size_t nblk = ((size_t)p.tile_end[p.nseg - 1] + 3) / 4;
  size_t cap = (size_t)device_cu_count(s);              // persistent: one workgroup per CU walks the tile groups
  if (opt(L2O_OPT_BWD_BLOCKS) > 0) cap = (size_t)opt(L2O_OPT_BWD_BLOCKS);
  if (nblk > cap && p.T <= 1) nblk = cap;                  // (a T-step launch keeps one workgroup per four tiles: each lives T steps)
  const dim3 grid((unsigned)nblk), block(256);
  void (*fn)(BwdParams) = nullptr;
  size_t lds = 0;
  if (p.wpack && opt(L2O_OPT_BWD_KERNEL) == 0) {                // the matrix-core form needs the packed weights
    switch (pre) {
      case L2O_PRE_IDENTITY: fn = k_cwlstm_bwd_mfma<L2O_PRE_IDENTITY>; lds = BwdMfmaGeom<L2O_PRE_IDENTITY>::kLdsFloats; break;
      case L2O_PRE_LOGSIGN: fn = k_cwlstm_bwd_mfma<L2O_PRE_LOGSIGN>; lds = BwdMfmaGeom<L2O_PRE_LOGSIGN>::kLdsFloats; break;
      default: fn = k_cwlstm_bwd_mfma<L2O_PRE_FC_ELU>; lds = BwdMfmaGeom<L2O_PRE_FC_ELU>::kLdsFloats;
    }
  } else {
    switch (pre) {
      case L2O_PRE_IDENTITY: fn = k_cwlstm_bwd_tile<L2O_PRE_IDENTITY>; lds = BwdTileGeom<L2O_PRE_IDENTITY>::kLdsFloats; break;
      case L2O_PRE_LOGSIGN: fn = k_cwlstm_bwd_tile<L2O_PRE_LOGSIGN>; lds = BwdTileGeom<L2O_PRE_LOGSIGN>::kLdsFloats; break;
      default: fn = k_cwlstm_bwd_tile<L2O_PRE_FC_ELU>; lds = BwdTileGeom<L2O_PRE_FC_ELU>::kLdsFloats;
    }
  }
  lds *= sizeof(float);
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(fn, grid, block, lds, s, p);
  HIP_TRY(hipGetLastError());
  return L2O_OK;
}

static void fill_bwd_net(BwdParams& p, const l2o_net_cfg* cfg, const l2o_net_weights* w, double pow1, double pow2) {
  p.pre = cfg->preprocess; p.tanh_output = cfg->tanh_output; p.n_layers = cfg->n_layers;
  p.scale = (float)cfg->scale;
  p.k_inv = cfg->logsign_k != 0.0 ? (float)(1.0 / cfg->logsign_k) : 0.0f;
  p.exp_k = (float)std::exp(cfg->logsign_k);
  p.beta1 = (float)cfg->beta1; p.beta2 = (float)cfg->beta2;
  p.om1 = (float)(1.0 - pow1); p.om2 = (float)(1.0 - pow2);
  p.wg1 = w->w_gates1; p.bg1 = w->b_gates1; p.wg2 = w->w_gates2; p.bg2 = w->b_gates2;
  p.wl = w->w_lin; p.bl = w->b_lin; p.wfc = w->w_fc; p.bfc = w->b_fc;
  p.wpack = w->wpack;
}

int l2o_cwlstm_bwd_multi(const l2o_net_cfg* cfg, const l2o_net_weights* w, const l2o_bwd_seg* segs, int32_t nseg,
                         const float* carry_in, float* carry_out, float* A, float* Bm, double pow1, double pow2,
                         void* stream) {
  OptScope opt_scope(cfg_optw(cfg));
  if (!cfg || !w || !segs || nseg < 1 || !carry_in || !carry_out || !A || !Bm)
    return fail(L2O_ERR_ARG, "l2o_cwlstm_bwd_multi: bad argument");
  if (nseg > 8) return fail(L2O_ERR_UNSUPPORTED, "l2o_cwlstm_bwd_multi: at most 8 panels");
  if (!net_ok_for_mfma(cfg) || cfg->n_layers == 0)
    return fail(L2O_ERR_UNSUPPORTED, "l2o_cwlstm_bwd_multi: only layers=(20,20) nets");
  if (!w->w_gates1 || !w->b_gates1 || !w->w_gates2 || !w->b_gates2 || !w->w_lin || !w->b_lin)
    return fail(L2O_ERR_ARG, "l2o_cwlstm_bwd_multi: NULL weights");
  const int pre = cfg->preprocess;
  if (pre == L2O_PRE_FC_ELU && (!w->w_fc || !w->b_fc)) return fail(L2O_ERR_ARG, "l2o_cwlstm_bwd_multi: fc weights");
  if (((uintptr_t)A & 15) || ((uintptr_t)Bm & 15)) return fail(L2O_ERR_ARG, "l2o_cwlstm_bwd_multi: A / Bm must be 16-byte aligned");
  BwdParams p;
  std::memset(&p, 0, sizeof(p));
  fill_bwd_net(p, cfg, w, pow1, pow2);
  p.carry_in = carry_in; p.carry_out = carry_out; p.act1 = A; p.dz1 = Bm;
  p.nseg = nseg;
  long tiles = 0;
  for (int i = 0; i < nseg; ++i) {
    const l2o_bwd_seg& sgm = segs[i];
    if (!sgm.g || !sgm.st_prev || !sgm.dx_next || sgm.B <= 0 || sgm.D <= 0 ||
        (pre == L2O_PRE_FC_ELU && (!sgm.m || !sgm.v)))
      return fail(L2O_ERR_ARG, "l2o_cwlstm_bwd_multi: bad panel %d", i);
    if (sgm.D % kTile != 0 && sgm.B != 1)
      return fail(L2O_ERR_UNSUPPORTED, "l2o_cwlstm_bwd_multi: panel %d needs D %% 16 == 0 or B == 1", i);
    const long n = (long)(sgm.B * sgm.D);
    tiles += (n + kTile - 1) / kTile;
    if (tiles > INT32_MAX / kTile) return fail(L2O_ERR_UNSUPPORTED, "l2o_cwlstm_bwd_multi: too many coordinates");
    p.tile_end[i] = (int)tiles;
    p.seg_n[i] = n;
    p.seg_d[i] = n; p.seg_tpp[i] = (int)((n + kTile - 1) / kTile);   // (tile-aligned or one row: the flat numbering)
    p.seg_g[i] = sgm.g; p.seg_m[i] = sgm.m; p.seg_v[i] = sgm.v; p.seg_st[i] = sgm.st_prev; p.seg_dx[i] = sgm.dx_next;
  }
  p.rows_total = tiles * kTile;
  return launch_bwd_tile(p, pre, (hipStream_t)stream);
}

static int bwd_unroll_impl(const l2o_net_cfg* cfg, const l2o_net_weights* w, const l2o_bwd_unroll_seg* segs,
                           int32_t nseg, const float* const* table, int32_t T, int64_t step0, const float* carry_in,
                           float* carry_out, float* A, float* Bm, void* stream, int compact);
int l2o_cwlstm_bwd_unroll(const l2o_net_cfg* cfg, const l2o_net_weights* w, const l2o_bwd_unroll_seg* segs,
                          int32_t nseg, const float* const* table, int32_t T, int64_t step0, const float* carry_in,
                          float* carry_out, float* A, float* Bm, void* stream) {
  return bwd_unroll_impl(cfg, w, segs, nseg, table, T, step0, carry_in, carry_out, A, Bm, stream, 0);
}
int l2o_cwlstm_bwd_unroll_compact(const l2o_net_cfg* cfg, const l2o_net_weights* w, const l2o_bwd_unroll_seg* segs,
                                  int32_t nseg, const float* const* table, int32_t T, int64_t step0, const float* carry_in,
                                  float* carry_out, float* Ac, float* Bm, void* stream) {
  return bwd_unroll_impl(cfg, w, segs, nseg, table, T, step0, carry_in, carry_out, Ac, Bm, stream, 1);
}
static int bwd_unroll_impl(const l2o_net_cfg* cfg, const l2o_net_weights* w, const l2o_bwd_unroll_seg* segs,
                           int32_t nseg, const float* const* table, int32_t T, int64_t step0, const float* carry_in,
                           float* carry_out, float* A, float* Bm, void* stream, int compact) {
  OptScope opt_scope(cfg_optw(cfg));
  if (!cfg || !w || !segs || nseg < 1 || !table || T < 1 || step0 < 0 || !A || !Bm)
    return fail(L2O_ERR_ARG, "l2o_cwlstm_bwd_unroll: bad argument");
  if (nseg > 8) return fail(L2O_ERR_UNSUPPORTED, "l2o_cwlstm_bwd_unroll: at most 8 panels");
  if (!net_ok_for_mfma(cfg) || cfg->n_layers == 0)
    return fail(L2O_ERR_UNSUPPORTED, "l2o_cwlstm_bwd_unroll: only layers=(20,20) nets");
  if (!w->wpack) return fail(L2O_ERR_ARG, "l2o_cwlstm_bwd_unroll: needs l2o_net_weights.wpack");
  if (((uintptr_t)A & 15) || ((uintptr_t)Bm & 15)) return fail(L2O_ERR_ARG, "l2o_cwlstm_bwd_unroll: A / Bm must be 16-byte aligned");
  const int pre = cfg->preprocess;
  BwdParams p;
  std::memset(&p, 0, sizeof(p));
  fill_bwd_net(p, cfg, w, 0.0, 0.0);
  p.carry_in = carry_in; p.carry_out = carry_out; p.act1 = A; p.dz1 = Bm;
  p.nseg = nseg; p.T = T; p.table = table;
  if (compact && opt(L2O_OPT_BWD_KERNEL) != 0)
    return fail(L2O_ERR_UNSUPPORTED, "l2o_cwlstm_bwd_unroll_compact: the matrix-core BPTT kernel only (L2O_OPT_BWD_KERNEL = 0)");
  p.compact_a = compact;
  p.pw1_last = std::pow((double)p.beta1, (double)(step0 + T - 1));
  p.pw2_last = std::pow((double)p.beta2, (double)(step0 + T - 1));
  long tiles = 0;
  for (int i = 0; i < nseg; ++i) {
    const l2o_bwd_unroll_seg& sgm = segs[i];
    if (sgm.B <= 0 || sgm.D <= 0) return fail(L2O_ERR_ARG, "l2o_cwlstm_bwd_unroll: bad panel %d", i);
    const long n = (long)(sgm.B * sgm.D);
    const long tpp = (long)tiles_per_problem(sgm.D);        // per-problem tiles, the packed-state layout of the forward
    tiles += (long)sgm.B * tpp;
    if (tiles > INT32_MAX / kTile) return fail(L2O_ERR_UNSUPPORTED, "l2o_cwlstm_bwd_unroll: too many coordinates");
    p.tile_end[i] = (int)tiles;
    p.seg_n[i] = n;
    p.seg_d[i] = (long)sgm.D;
    p.seg_tpp[i] = (int)tpp;
    p.seg_gfinal[i] = sgm.g_final;
  }
  p.rows_total = tiles * kTile;
  return launch_bwd_tile(p, pre, (hipStream_t)stream);
}

int l2o_cwlstm_bwd_step(const l2o_net_cfg* cfg, const l2o_net_weights* w, const l2o_bwd_io* io, double pow1,
                        double pow2, int64_t B, int64_t D, void* stream) {
  OptScope opt_scope(cfg_optw(cfg));
  if (!cfg || !w || !io || B <= 0 || D <= 0 || !io->g || !io->dx_next || !io->act1 || !io->dd || !w->w_lin ||
      !w->b_lin)
    return fail(L2O_ERR_ARG, "l2o_cwlstm_bwd_step: bad argument");
  BwdParams p;
  std::memset(&p, 0, sizeof(p));
  p.B = (int)B; p.D = (int)D; p.tpp = tiles_per_problem(D);
  p.pre = cfg->preprocess; p.tanh_output = cfg->tanh_output; p.n_layers = cfg->n_layers;
  p.scale = (float)cfg->scale;
  p.k_inv = cfg->logsign_k != 0.0 ? (float)(1.0 / cfg->logsign_k) : 0.0f;
  p.exp_k = (float)std::exp(cfg->logsign_k);
  p.beta1 = (float)cfg->beta1; p.beta2 = (float)cfg->beta2;
  p.om1 = (float)(1.0 - pow1); p.om2 = (float)(1.0 - pow2);
  p.wg1 = w->w_gates1; p.bg1 = w->b_gates1; p.wg2 = w->w_gates2; p.bg2 = w->b_gates2;
  p.wl = w->w_lin; p.bl = w->b_lin; p.wfc = w->w_fc; p.bfc = w->b_fc;
  p.wpack = w->wpack;
  p.g = io->g; p.m = io->m; p.v = io->v; p.st_prev = io->st_prev; p.dx_next = io->dx_next;
  p.carry_in = io->carry_in; p.carry_out = io->carry_out; p.act1 = io->act1; p.dz1 = io->dz1;
  p.act2 = io->act2; p.dz2 = io->dz2; p.h2o = io->h2; p.dd = io->dd; p.feats = io->feats; p.du = io->du;
  p.dg = io->dg;
  if (io->dg && cfg->preprocess == L2O_PRE_FC_ELU)
    return fail(L2O_ERR_UNSUPPORTED, "l2o_cwlstm_bwd_step: the input adjoint (dg) is implemented for the DM nets (identity / LogAndSign)");
  {
    const int pre_ = cfg->preprocess;
    const long P_ = cfg->n_layers == 0 ? 2 : (pre_ == L2O_PRE_FC_ELU ? kH : (pre_ == L2O_PRE_LOGSIGN ? 2 : 1));
    const long K1_ = cfg->n_layers == 0 ? 2 : P_ + kH;
    const long as = (long)io->a_stride, bs = (long)io->b_stride;
    p.s_act1 = as ? as : K1_; p.s_act2 = as ? as : 2 * kH; p.s_h2 = as ? as : kH; p.s_feats = as ? as : 2;
    p.s_dz1 = bs ? bs : 4 * kH; p.s_dz2 = bs ? bs : 4 * kH; p.s_dd = bs ? bs : 1; p.s_du = bs ? bs : kH;
  }
  hipStream_t s = (hipStream_t)stream;
  const size_t N = (size_t)B * D;
  if (cfg->n_layers == 0) {
    if (cfg->kind != L2O_NET_CW || cfg->preprocess == L2O_PRE_FC_ELU)
      return fail(L2O_ERR_UNSUPPORTED, "layers=() is implemented for CoordinateWiseDeepLSTM only");
    const dim3 grid((unsigned)((N + 255) / 256));
    if (cfg->preprocess == L2O_PRE_LOGSIGN) hipLaunchKernelGGL(k_linear_bwd_step<L2O_PRE_LOGSIGN>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(k_linear_bwd_step<L2O_PRE_IDENTITY>, grid, dim3(256), 0, s, p);
    HIP_TRY(hipGetLastError());
    return L2O_OK;
  }
  if (!net_ok_for_mfma(cfg)) return fail(L2O_ERR_UNSUPPORTED, "l2o_cwlstm_bwd_step: only layers=(20,20) / () nets");
  if (!w->w_gates1 || !w->b_gates1 || !w->w_gates2 || !w->b_gates2 || !io->st_prev || !io->carry_in ||
      !io->carry_out || !io->dz1 || !io->act2 || !io->dz2 || !io->h2)
    return fail(L2O_ERR_ARG, "l2o_cwlstm_bwd_step: NULL LSTM buffer");
  const int pre = cfg->preprocess;
  // fast path: tile-aligned panel + the interleaved A / Bm layout -> the quad / LDS-tile kernel
  {
    const int Pq = pre == L2O_PRE_FC_ELU ? kH : (pre == L2O_PRE_LOGSIGN ? 2 : 1);
    const int K1q = Pq + kH;
    const long KA = K1q + 3 * kH + (pre == L2O_PRE_FC_ELU ? 2 : 0) + 1, KB = 8 * kH + 1 + (pre == L2O_PRE_FC_ELU ? kH : 0);
    const bool layout = io->a_stride == KA && io->b_stride == KB && io->act2 == io->act1 + K1q &&
                        io->h2 == io->act1 + K1q + 2 * kH && io->dz2 == io->dz1 + 4 * kH && io->dd == io->dz1 + 8 * kH &&
                        (pre != L2O_PRE_FC_ELU || (io->feats == io->act1 + K1q + 3 * kH && io->du == io->dz1 + 8 * kH + 1 &&
                                                   io->m && io->v && w->w_fc && w->b_fc));
    if (layout && (D % kTile == 0 || B == 1) && ((uintptr_t)io->act1 & 15) == 0 && ((uintptr_t)io->dz1 & 15) == 0 &&
        opt(L2O_OPT_BWD_KERNEL) != 2 && !io->dg) {           // (the input adjoint is emitted by the generic kernel)
      p.nseg = 1;
      p.tile_end[0] = (int)((N + kTile - 1) / kTile);
      p.seg_n[0] = (long)N;
      p.seg_d[0] = (long)N; p.seg_tpp[0] = p.tile_end[0];
      p.seg_g[0] = io->g; p.seg_m[0] = io->m; p.seg_v[0] = io->v; p.seg_st[0] = io->st_prev; p.seg_dx[0] = io->dx_next;
      p.rows_total = (long)N;
      return launch_bwd_tile(p, pre, s);
      return L2O_OK;
    }
  }
  const size_t lds = 0;                                   // static LDS only (the per-thread input column)
  const dim3 grid((unsigned)((N + 63) / 64)), block(64);
  switch (pre) {
    case L2O_PRE_IDENTITY: hipLaunchKernelGGL(k_cwlstm_bwd_step<L2O_PRE_IDENTITY>, grid, block, lds, s, p); break;
    case L2O_PRE_LOGSIGN: hipLaunchKernelGGL(k_cwlstm_bwd_step<L2O_PRE_LOGSIGN>, grid, block, lds, s, p); break;
    default:
      if (!io->m || !io->v || !io->feats || !io->du || !w->w_fc || !w->b_fc)
        return fail(L2O_ERR_ARG, "l2o_cwlstm_bwd_step: RNNProp needs m, v, feats, du and the fc weights");
      hipLaunchKernelGGL(k_cwlstm_bwd_step<L2O_PRE_FC_ELU>, grid, block, lds, s, p);
  }
  HIP_TRY(hipGetLastError());
  return L2O_OK;
}

int l2o_unroll_supported(const l2o_net_cfg* cfg, const l2o_problem* prob) {
  OptScope opt_scope(cfg_optw(cfg));
  if (!cfg || !prob || !net_ok_for_mfma(cfg)) return 0;
  if (prob->kind != L2O_PROB_QUADRATIC && prob->kind != L2O_PROB_LASSO && prob->kind != L2O_PROB_RASTRIGIN &&
      prob->kind != L2O_PROB_SQUARE_COS)
    return 0;
  if (prob->D <= 0 || prob->M <= 0) return 0;
  UnrollGeom g;
  return (unroll_geom(prob, &g) || unroll_cu_eligible(prob)) ? 1 : 0;
}

int l2o_unroll_record_supported(const l2o_net_cfg* cfg, const l2o_problem* prob) {
  OptScope opt_scope(cfg_optw(cfg));
  return l2o_unroll_supported(cfg, prob);                   // every fused form records (ABI v6: the streaming form too)
}

size_t l2o_unroll_workspace_bytes(const l2o_net_cfg* cfg, const l2o_problem* prob, int32_t T) {
  OptScope opt_scope(cfg_optw(cfg));
  if (!l2o_unroll_supported(cfg, prob) || T < 0) return 0;
  UnrollGeom g;
  if (!unroll_geom(prob, &g) || g.CH < 2) return 0;       // (the streaming form needs no workspace)
  return pair_layout(prob, g, T).total;
}

int l2o_unroll_status(const void* workspace_header_host) {
  if (!workspace_header_host) return L2O_OK;
  const unsigned st = *static_cast<const unsigned*>(workspace_header_host);
  if (st == 0) return L2O_OK;
  if (st == 2)
    return fail(L2O_ERR_TIMEOUT, "l2o_mlp_unroll: a workgroup's all-reduce inputs never arrived (status 2): the persistent "
                             "launch was not fully co-resident (a shared / masked device?); the iterate and LSTM state "
                             "of that launch are invalid.  Run l2o_coresident_workgroups once so that the library "
                             "sizes against what is really available, or switch the fused form off "
                             "(L2O_OPT_MLP_UNROLL = 0 / L2O_NO_MLP_UNROLL=1)");
  return fail(L2O_ERR_TIMEOUT, "l2o_unroll: partner workgroup timed out (status %u): the two halves of a problem were not "
                           "co-resident (a shared / masked device?); the iterate and LSTM state of that launch are "
                           "invalid.  Run l2o_coresident_workgroups once so that the library sizes against what is "
                           "really available, or use one CU per problem (L2O_OPT_PAIR = 0 / L2O_NO_PAIR=1)", st);
}

static int unroll_impl(const l2o_net_cfg* cfg, const float* wpack, const l2o_problem* prob, float* x, float* st,
                       float* m, float* v, int32_t T, int32_t step0, float* fx_part, void* workspace,
                       const l2o_unroll_hist* hist, float* fx, void* stream, const float* x0 = nullptr, int flags = 0);

int l2o_unroll(const l2o_net_cfg* cfg, const float* wpack, const l2o_problem* prob, float* x, float* st, float* m,
               float* v, int32_t T, int32_t step0, float* fx_part, void* workspace, void* stream) {
  OptScope opt_scope(cfg_optw(cfg));
  return unroll_impl(cfg, wpack, prob, x, st, m, v, T, step0, fx_part, workspace, nullptr, nullptr, stream);
}

int l2o_unroll_record(const l2o_net_cfg* cfg, const float* wpack, const l2o_problem* prob, float* x, float* st,
                      float* m, float* v, int32_t T, int32_t step0, float* fx_part, void* workspace,
                      const l2o_unroll_hist* hist, void* stream) {
  OptScope opt_scope(cfg_optw(cfg));
  return unroll_impl(cfg, wpack, prob, x, st, m, v, T, step0, fx_part, workspace, hist, nullptr, stream);
}

int l2o_unroll_reduce(const l2o_net_cfg* cfg, const float* wpack, const l2o_problem* prob, const float* x0, float* x,
                      float* st, float* m, float* v, int32_t T, int32_t step0, int32_t flags, float* fx_part, float* fx,
                      void* workspace, const l2o_unroll_hist* hist, void* stream) {
  OptScope opt_scope(cfg_optw(cfg));
  if (!fx) return fail(L2O_ERR_ARG, "l2o_unroll_reduce: NULL fx");
  if (flags & ~L2O_UNROLL_ZERO_STATE)
    return fail(L2O_ERR_ARG, "l2o_unroll_reduce: unknown flags %d", flags);
  return unroll_impl(cfg, wpack, prob, x, st, m, v, T, step0, fx_part, workspace, hist, fx, stream, x0, flags);
}

int32_t l2o_coresident_workgroups(void* scratch, void* stream) {
  if (!scratch) return fail(L2O_ERR_ARG, "l2o_coresident_workgroups: NULL scratch");
  hipStream_t s = (hipStream_t)stream;
  const int dev = device_ordinal(s);
  if (dev >= 0 && dev < 64) {
    const int m = g_measured_cus[dev].load(std::memory_order_relaxed);
    if (m > 0) return m;
  }
  const int n = device_cu_count(s);
  if (n <= 0) return fail(L2O_ERR_HIP, "l2o_coresident_workgroups: no device");
  unsigned* d = static_cast<unsigned*>(scratch);
  constexpr int kLds = 100 * 1024;                           // more than half a CU's LDS: one workgroup per CU
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_coresident_probe), hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
  // A pass can only UNDER-count (a slow dispatch lets the earliest workgroup sample before the others arrived), never
  // over-count (at most `capacity` workgroups exist until the first one leaves): the maximum of a few passes.
  int got = 0;
  for (int pass = 0; pass < 3; ++pass) {
    const unsigned init[2] = {0u, 0xffffffffu};
    HIP_TRY(hipMemcpyAsync(d, init, sizeof(init), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_coresident_probe, dim3(n), dim3(256), kLds, s, d, d + 1);
    HIP_TRY(hipGetLastError());
    unsigned out[2] = {0u, 0u};
    HIP_TRY(hipMemcpyAsync(out, d, sizeof(out), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const int seen = (int)out[1];
    if (seen > got && seen <= n) got = seen;
    if (got == n) break;
  }
  if (got <= 0) got = n;
  if (dev >= 0 && dev < 64) g_measured_cus[dev].store(got, std::memory_order_relaxed);
  return got;
}

int l2o_unroll_workspace_init(void* workspace, size_t bytes, void* stream) {
  if (!workspace && bytes) return fail(L2O_ERR_ARG, "l2o_unroll_workspace_init: NULL workspace");
  if (bytes) HIP_TRY(hipMemsetAsync(workspace, 0, bytes, (hipStream_t)stream));
  return L2O_OK;
}

int64_t l2o_unroll_workspace_layout(const l2o_net_cfg* cfg, const l2o_problem* prob) {
  OptScope opt_scope(cfg_optw(cfg));
  UnrollGeom g;
  if (!cfg || !prob || !l2o_unroll_supported(cfg, prob) || !unroll_geom(prob, &g) || g.CH < 2) return 0;
  return ((int64_t)prob->B_local << 8) | g.CH;            // changes whenever the granule area's layout does
}

static int unroll_impl(const l2o_net_cfg* cfg, const float* wpack, const l2o_problem* prob, float* x, float* st,
                       float* m, float* v, int32_t T, int32_t step0, float* fx_part, void* workspace,
                       const l2o_unroll_hist* hist, float* fx, void* stream, const float* x0, int flags) {
  int rc = check_problem(prob);
  if (rc) return rc;
  if (!cfg || !wpack || !x || !st || !fx_part || T < 0) return fail(L2O_ERR_ARG, "l2o_unroll: bad argument");
  if (hist && (!hist->st || !hist->g || !hist->g_final ||
               (cfg->preprocess == L2O_PRE_FC_ELU && (!hist->m || !hist->v))))
    return fail(L2O_ERR_ARG, "l2o_unroll_record: incomplete history buffers");
  if (!l2o_unroll_supported(cfg, prob))
    return fail(L2O_ERR_UNSUPPORTED, "l2o_unroll: no fused kernel for kind=%d D=%d M=%d net(kind=%d,layers=%d)",
                prob->kind, prob->D, prob->M, cfg->kind, cfg->n_layers);
  UnrollGeom g;
  const bool lds_form = unroll_geom(prob, &g);
  UnrollArgs a;
  a.np = make_net_params(cfg, wpack);
  a.pp = make_prob_params(prob);
  a.x = x; a.st = st; a.m = m; a.v = v; a.fx_part = fx_part;
  a.x_in = x0; a.zero_state = (flags & L2O_UNROLL_ZERO_STATE) ? 1 : 0;
  a.ticks = nullptr;
  a.T = T;
  a.hist_st = hist ? hist->st : nullptr;
  a.hist_g = hist ? hist->g : nullptr;
  a.hist_m = hist ? hist->m : nullptr;
  a.hist_v = hist ? hist->v : nullptr;
  a.hist_gfinal = hist ? hist->g_final : nullptr;
  pow_ff(cfg->beta1, step0, &a.p1_hi, &a.p1_lo);
  pow_ff(cfg->beta2, step0, &a.p2_hi, &a.p2_lo);
  hipStream_t s = (hipStream_t)stream;
  bool fx_done = false;
  rc = L2O_OK;
  if (cfg->preprocess == L2O_PRE_FC_ELU && (!m || !v)) return fail(L2O_ERR_ARG, "l2o_unroll: RNNProp needs m and v");
  if (!lds_form) {
    switch (cfg->preprocess) {
      case L2O_PRE_IDENTITY: rc = launch_unroll_cu<L2O_PRE_IDENTITY>(a, s); break;
      case L2O_PRE_LOGSIGN: rc = launch_unroll_cu<L2O_PRE_LOGSIGN>(a, s); break;
      default: rc = launch_unroll_cu<L2O_PRE_FC_ELU>(a, s);
    }
  } else {
    switch (cfg->preprocess) {
      case L2O_PRE_IDENTITY: rc = launch_unroll_kind<L2O_PRE_IDENTITY>(a, g, prob->kind, s, prob, workspace, fx, &fx_done); break;
      case L2O_PRE_LOGSIGN: rc = launch_unroll_kind<L2O_PRE_LOGSIGN>(a, g, prob->kind, s, prob, workspace, fx, &fx_done); break;
      default: rc = launch_unroll_kind<L2O_PRE_FC_ELU>(a, g, prob->kind, s, prob, workspace, fx, &fx_done);
    }
  }
  if (rc) return rc;
  if (fx && !fx_done)                                       // forms without the fused epilogue: the separate reduction
    return l2o_reduce_fx(fx_part, T + 1, prob->B_local, prob->B_global, fx, stream);
  return L2O_OK;
}

int l2o_reduce_fx(const float* fx_part, int32_t T1, int32_t B_local, int32_t B_global, float* fx, void* stream) {
  if (!fx_part || !fx || T1 <= 0 || B_local <= 0 || B_global < B_local)
    return fail(L2O_ERR_ARG, "l2o_reduce_fx: bad argument");
  hipLaunchKernelGGL(k_reduce_fx, dim3(T1), dim3(64), 0, (hipStream_t)stream, fx_part, (int)T1, (int)B_local,
                     1.0f / (float)B_global, fx);
  HIP_TRY(hipGetLastError());
  return L2O_OK;
}

// ---- small vector passes of the meta-gradient (csrc/l2o_vecops.h; ABI v11) ---------------------------------
int l2o_suffix_sums(const float* const* g, const float* g_final, float* out, int64_t n, int32_t T, void* stream) {
  if (!g || !g_final || !out || n <= 0 || T < 0) return fail(L2O_ERR_ARG, "l2o_suffix_sums: bad argument");
  if (T == 0) return L2O_OK;
  hipLaunchKernelGGL(k_suffix_sums, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, g_final, out,
                     (long)n, (int)T);
  HIP_TRY(hipGetLastError());
  return L2O_OK;
}
size_t l2o_colsum_scratch_floats(int64_t batch, int32_t cols) {
  if (batch <= 0 || cols <= 0) return 0;
  return (size_t)batch * kColsumSplit * (size_t)cols;
}
int l2o_colsum(const float* A, int64_t batch, int64_t rows, int32_t cols, float* out, int32_t accumulate, float* scratch,
               void* stream) {
  if (!A || !out || !scratch || batch <= 0 || rows <= 0 || cols <= 0 || batch > 65535)
    return fail(L2O_ERR_ARG, "l2o_colsum: bad argument");
  const unsigned gx = (unsigned)((cols + 255) / 256);
  hipLaunchKernelGGL(k_colsum_part, dim3(gx, kColsumSplit, (unsigned)batch), dim3(256), 0, (hipStream_t)stream, A, (long)rows,
                     (int)cols, scratch);
  hipLaunchKernelGGL(k_colsum_final, dim3(gx, (unsigned)batch), dim3(256), 0, (hipStream_t)stream, scratch, (int)cols, out,
                     (int)accumulate);
  HIP_TRY(hipGetLastError());
  return L2O_OK;
}
int l2o_lincomb(float* out, const float* a, float ca, const float* b, float cb, const float* c, float cc, int64_t n,
                void* stream) {
  if (!out || !a || n <= 0) return fail(L2O_ERR_ARG, "l2o_lincomb: bad argument");
  hipLaunchKernelGGL(k_lincomb, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, a, ca, b, cb, c, cc,
                     (long)n);
  HIP_TRY(hipGetLastError());
  return L2O_OK;
}
int l2o_rnnprop_input_adjoint(const float* Bm, int64_t ldb, int32_t du_col, int32_t H, const float* w_fc, const float* g,
                              const float* m, const float* v, double pow1, double pow2, double beta1, double beta2,
                              float* dm, float* dv, float* dg, int64_t n, void* stream) {
  if (!Bm || !w_fc || !g || !m || !v || !dm || !dv || !dg || n <= 0 || H <= 0 || du_col < 0 || ldb < du_col + H)
    return fail(L2O_ERR_ARG, "l2o_rnnprop_input_adjoint: bad argument");
  RnnpropAdjArgs a;
  a.Bm = Bm; a.ldb = (long)ldb; a.du_col = du_col; a.H = H; a.w_fc = w_fc; a.g = g; a.m = m; a.v = v;
  a.om1 = (float)(1.0 - pow1); a.om2 = (float)(1.0 - pow2);
  a.b1 = (float)beta1; a.b2 = (float)beta2;
  a.omb1 = 1.0f - a.b1; a.omb2 = 1.0f - a.b2;               // (fp32, like the forward's 1 - beta)
  a.dm = dm; a.dv = dv; a.dg = dg; a.n = (long)n;
  hipLaunchKernelGGL(k_rnnprop_input_adjoint, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  HIP_TRY(hipGetLastError());
  return L2O_OK;
}

}  // extern "C"

#endif  // L2O_TU_ILP
