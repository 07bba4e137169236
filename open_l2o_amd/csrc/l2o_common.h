// l2o_common.h -- shared device helpers for the gfx950 (CDNA4) L2O kernels.
//
// Work decomposition used by every LSTM kernel in this library
// ------------------------------------------------------------
// A *tile* is 16 optimizee coordinates handled by ONE 64-lane wavefront.
// Lane l = (c, q) with c = l & 15 (coordinate inside the tile) and q = l >> 4.
// Lane (c, q) owns the five hidden units u = 4*t + q (t = 0..4) of coordinate c
// in both LSTM layers: h1[t], c1[t], h2[t], c2[t] live in its registers.
//
// The gate pre-activations  z[80] = [in, h_prev] @ w_gates + b_gates  of the 16
// coordinates are computed TRANSPOSED on the matrix cores with
// v_mfma_f32_16x16x4_f32 (exact fp32, == an fmaf chain):
//     D[rho][c] += A[rho][k] * B[k][c]
//   A = a 16-row slice of w_gates^T (weights; one VGPR per (k-step, slice))
//   B = the activations            (one VGPR per k-step: lane (c,q) supplies k = 4*kk + q)
//   D = 16 gate rows x 16 coordinates; lane (c,q) register r holds row rho = 4*q + r.
// The gate rows of slice t are permuted so that rho = 4*q + r  <->  gate r (i,j,f,o)
// of hidden unit 4*t + q: every lane receives exactly the four gates of the units it
// owns, and the K ordering k = 4*kk + q <-> unit 4*kk + q means the B operand of the
// next step / next layer is the h value the lane already holds.  No transpose, no
// LDS round trip, no cross-lane traffic in the recurrent loop; the only shuffles are
// the two butterfly adds of the 20 -> 1 output Linear.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/l2o_abi.h"

// wave-uniform reads through the scalar unit: a float pointer in the constant address space
typedef const __attribute__((address_space(4))) float* l2o_cfp;

namespace l2o {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kH = 20;        // hidden units per LSTM layer implemented by the MFMA kernels
constexpr int kNT = 5;        // unit slices (kH / 4)
constexpr int kTile = 16;     // coordinates per wave tile
constexpr int kStateFloatsPerTile = 4 * kH * kTile;   // h1,c1,h2,c2 -> 1280 floats

// ---- wpack row offsets (one "row" = 64 floats, lane-major) -----------------
__host__ __device__ constexpr int wp_ks1(int pre) { return pre == L2O_PRE_FC_ELU ? 10 : 6; }
__host__ __device__ constexpr int wp_row_a1(int) { return 0; }
__host__ __device__ constexpr int wp_row_b1(int pre) { return wp_ks1(pre) * kNT; }
__host__ __device__ constexpr int wp_row_a2(int pre) { return wp_row_b1(pre) + 4 * kNT; }
__host__ __device__ constexpr int wp_row_b2(int pre) { return wp_row_a2(pre) + 10 * kNT; }
__host__ __device__ constexpr int wp_row_wl(int pre) { return wp_row_b2(pre) + 4 * kNT; }
__host__ __device__ constexpr int wp_row_bl(int pre) { return wp_row_wl(pre) + kNT; }
__host__ __device__ constexpr int wp_row_fc(int pre) { return wp_row_bl(pre) + 1; }
__host__ __device__ constexpr int wp_rows(int pre) { return wp_row_fc(pre) + 3 * kNT; }

// ---- fast, accurate-enough transcendental forms ---------------------------
// v_exp_f32 / v_rcp_f32 / v_log_f32 / v_sqrt_f32 are 1-ulp instructions; the forms
// below keep the ABSOLUTE error of sigmoid/tanh at the 1e-7 level (fp32 rounding of
// an O(1) quantity), which is what the recurrent state needs.
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// (L2O_ABLATE_* macros exist only for the timing ablations of scripts/ablate.sh; they
// produce wrong numbers on purpose and are never defined in the shipped library.)
__device__ __forceinline__ float sigmoidf_(float x) {
#ifdef L2O_ABLATE_TRANS
  return x * 0.25f + 0.5f;
#endif
  // 1 / (1 + e^-x) ; e^-x = 2^(-x*log2e)
  return fast_rcp(1.0f + fast_exp2(x * -1.4426950408889634f));
}
// sigmoid(x + 1)  (snt.LSTM forget_bias = 1.0) with the +1 folded into the FMA
__device__ __forceinline__ float sigmoid_p1f_(float x) {
#ifdef L2O_ABLATE_TRANS
  return x * 0.25f + 0.75f;
#endif
  return fast_rcp(1.0f + fast_exp2(__builtin_fmaf(x, -1.4426950408889634f, -1.4426950408889634f)));
}
__device__ __forceinline__ float tanhf_(float x) {
#ifdef L2O_ABLATE_TRANS
  return x * 0.5f;
#endif
  // tanh|x| = (1 - e)/(1 + e), e = e^(-2|x|) in (0, 1]: no overflow, abs err ~1e-7
  float e = fast_exp2(__builtin_fabsf(x) * -2.8853900817779268f);
  float t = (1.0f - e) * fast_rcp(1.0f + e);
  return __builtin_copysignf(t, x);
}
__device__ __forceinline__ float eluf_(float x) {
  // tf.nn.elu: x > 0 ? x : e^x - 1
  float e = fast_exp2(__builtin_fminf(x, 0.0f) * 1.4426950408889634f) - 1.0f;
  return x > 0.0f ? x : e;
}

// ---- cross-lane sums without LDS traffic ------------------------------------
// DPP row operations (quad_perm / row_half_mirror / row_mirror) inside a 16-lane row and
// the gfx950 v_permlane16_swap / v_permlane32_swap across rows: all VALU-rate, no
// ds_bpermute latency on the per-step critical path.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
// v[l] + v[l ^ 16]: swap odd rows of a copy with even rows of the other copy
__device__ __forceinline__ float xor16_add(float v) {
  const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// v[l] + v[l ^ 32]
__device__ __forceinline__ float xor32_add(float v) {
  const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// sum over the 4 lanes of a quad (lanes 4k..4k+3); result in all 4
__device__ __forceinline__ float quad_sum(float v) {
  v = dpp_add<0xB1>(v);   // quad_perm:[1,0,3,2]
  v = dpp_add<0x4E>(v);   // quad_perm:[2,3,0,1]
  return v;
}
// sum over a 16-lane row; result in all 16
__device__ __forceinline__ float row_sum16(float v) {
  v = quad_sum(v);
  v = dpp_add<0x141>(v);  // row_half_mirror
  v = dpp_add<0x140>(v);  // row_mirror
  return v;
}
__device__ __forceinline__ float wave_sum64(float v) {
  return xor32_add(xor16_add(row_sum16(v)));
}
// sum over the four q lanes that share a coordinate c (lanes c, c+16, c+32, c+48)
__device__ __forceinline__ float quad_q_sum(float v) {
  return xor32_add(xor16_add(v));
}

// Workgroup barrier for data exchanged through LDS only: wait for this wave's LDS operations, not for its
// global loads (__syncthreads() drains vmcnt as well, which would serialise the row groups requested
// ahead of the barrier behind it).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- weights in registers --------------------------------------------------
template <int PRE>
struct NetW {
  static constexpr int KS1 = wp_ks1(PRE);
  float a1[KS1][kNT];   // layer-1 w_gates^T fragments
  f32x4 b1[kNT];        // layer-1 bias as accumulator init (fc/RNNProp only; DM rides in a k-slot)
  float a2[10][kNT];    // layer-2 fragments: kk 0..4 <- h1 (new), kk 5..9 <- h2 (prev)
  f32x4 b2[kNT];
  float wl[kNT];        // output Linear
  float bl;
  float fcw0[kNT], fcw1[kNT], fcb[kNT];   // RNNProp input projection (2 -> 20)
};

template <int PRE>
__device__ __forceinline__ void load_netw(NetW<PRE>& w, const float* __restrict__ wp, int lane) {
  const float* p = wp + lane;
#pragma unroll
  for (int kk = 0; kk < NetW<PRE>::KS1; ++kk)
#pragma unroll
    for (int t = 0; t < kNT; ++t) w.a1[kk][t] = p[(wp_row_a1(PRE) + kk * kNT + t) * 64];
#pragma unroll
  for (int t = 0; t < kNT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (PRE == L2O_PRE_FC_ELU) w.b1[t][r] = p[(wp_row_b1(PRE) + t * 4 + r) * 64];
      else w.b1[t][r] = 0.0f;
      w.b2[t][r] = p[(wp_row_b2(PRE) + t * 4 + r) * 64];
    }
#pragma unroll
  for (int kk = 0; kk < 10; ++kk)
#pragma unroll
    for (int t = 0; t < kNT; ++t) w.a2[kk][t] = p[(wp_row_a2(PRE) + kk * kNT + t) * 64];
#pragma unroll
  for (int t = 0; t < kNT; ++t) {
    w.wl[t] = p[(wp_row_wl(PRE) + t) * 64];
    if (PRE == L2O_PRE_FC_ELU) {
      w.fcw0[t] = p[(wp_row_fc(PRE) + t) * 64];
      w.fcw1[t] = p[(wp_row_fc(PRE) + kNT + t) * 64];
      w.fcb[t] = p[(wp_row_fc(PRE) + 2 * kNT + t) * 64];
    }
  }
  w.bl = p[wp_row_bl(PRE) * 64];
}

struct TileState {
  float h1[kNT], c1[kNT], h2[kNT], c2[kNT];
};

// packed HBM layout of one tile's state: [5][64 lanes][4] floats; element e = 4*j + w of a
// lane is array a = e / 5 (h1, c1, h2, c2) slice t = e % 5.
__device__ __forceinline__ void load_tile_state(TileState& s, const float* __restrict__ st_tile, int lane) {
  float e[20];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const float4 v = *reinterpret_cast<const float4*>(st_tile + (j * 64 + lane) * 4);
    e[4 * j + 0] = v.x; e[4 * j + 1] = v.y; e[4 * j + 2] = v.z; e[4 * j + 3] = v.w;
  }
#pragma unroll
  for (int t = 0; t < kNT; ++t) { s.h1[t] = e[t]; s.c1[t] = e[5 + t]; s.h2[t] = e[10 + t]; s.c2[t] = e[15 + t]; }
}
__device__ __forceinline__ void store_tile_state(const TileState& s, float* __restrict__ st_tile, int lane) {
  float e[20];
#pragma unroll
  for (int t = 0; t < kNT; ++t) { e[t] = s.h1[t]; e[5 + t] = s.c1[t]; e[10 + t] = s.h2[t]; e[15 + t] = s.c2[t]; }
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    float4 v; v.x = e[4 * j]; v.y = e[4 * j + 1]; v.z = e[4 * j + 2]; v.w = e[4 * j + 3];
    *reinterpret_cast<float4*>(st_tile + (j * 64 + lane) * 4) = v;
  }
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
#ifdef L2O_ABLATE_MFMA
  c[0] = __builtin_fmaf(a, b, c[0]);
  return c;
#endif
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- optional phase clock (scripts/phase_profile.sh builds with -DL2O_PROFILE_PHASES; the
// shipped library compiles this to nothing) -------------------------------------------
struct PhaseClock {
#ifdef L2O_PROFILE_PHASES
  long long prev;
  long long acc[12];
  __device__ __forceinline__ void start() {
    for (int i = 0; i < 12; ++i) acc[i] = 0;
    prev = __builtin_readcyclecounter();
  }
  // a mark is a full fence in the profiling build: nothing is scheduled across it and all
  // outstanding memory operations are drained, so that each phase pays for its own work
  __device__ __forceinline__ void mark(int i) {
    __builtin_amdgcn_sched_barrier(0);
#ifdef L2O_PROFILE_LIGHT
    // (light marks: outstanding vector-memory operations -- the fire-and-forget publish stores, polls in flight --
    //  are NOT drained, so a phase shows what the wave actually waits for)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
    const long long n = __builtin_readcyclecounter();
    __builtin_amdgcn_sched_barrier(0);
    acc[i] += n - prev;
    prev = n;
  }
  // wait for the matrix pipe: touch every accumulator
  template <class A>
  __device__ __forceinline__ void drain(A& a) {
#pragma unroll
    for (int t = 0; t < 5; ++t) asm volatile("" : "+v"(a[t]));
  }
  template <class V>
  __device__ __forceinline__ void drain1(V& v) { asm volatile("" : "+v"(v)); }
  __device__ __forceinline__ void dump(long long* out) {
    for (int i = 0; i < 12; ++i) out[i] = acc[i];
  }
#else
  template <class V>
  __device__ __forceinline__ void drain1(V&) {}
  __device__ __forceinline__ void start() {}
  __device__ __forceinline__ void mark(int) {}
  template <class A>
  __device__ __forceinline__ void drain(A&) {}
  __device__ __forceinline__ void dump(long long*) {}
#endif
};

// ---- the optimizer network, split so that the matrix work that depends only on the
// PREVIOUS step's state can be issued early (interleaved with the optimizee GEMV) ----
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// slots LO..HI-1 of the 25 layer-2 MFMAs fed by the previous h2 (slot i: k-step i/5, slice i%5).
// The first k-step takes the bias fragment as its C operand (accumulator init for free).
template <int PRE, int LO, int HI>
__device__ __forceinline__ void lstm_issue_l2_prev(const NetW<PRE>& w, const TileState& s, f32x4 (&acc2)[kNT]) {
  static_for<LO, HI>([&](auto ic) {
    constexpr int i = decltype(ic)::value, kk = i / kNT, t = i % kNT;
    if constexpr (kk == 0) acc2[t] = mfma16(w.a2[5][t], s.h2[0], w.b2[t]);
    else acc2[t] = mfma16(w.a2[5 + kk][t], s.h2[kk], acc2[t]);
  });
}
// slots LO..HI-1 of the 25 layer-1 MFMAs fed by the previous h1; first k-step: C = bias
// fragment (fc / RNNProp) or the inline constant 0 (DM: the bias rides in a K slot).
template <int PRE, int LO, int HI>
__device__ __forceinline__ void lstm_issue_l1_prev(const NetW<PRE>& w, const TileState& s, f32x4 (&acc1)[kNT]) {
  static_for<LO, HI>([&](auto ic) {
    constexpr int i = decltype(ic)::value, kk = i / kNT, t = i % kNT;
    constexpr int ka = (PRE == L2O_PRE_FC_ELU) ? 5 + kk : kk;
    if constexpr (kk == 0) {
      if constexpr (PRE == L2O_PRE_FC_ELU) acc1[t] = mfma16(w.a1[ka][t], s.h1[0], w.b1[t]);
      else acc1[t] = mfma16(w.a1[ka][t], s.h1[0], f32x4{0.f, 0.f, 0.f, 0.f});
    } else {
      acc1[t] = mfma16(w.a1[ka][t], s.h1[kk], acc1[t]);
    }
  });
}

// LSTM nonlinearity of the 5 unit slices a lane owns (snt.LSTM: gates i, j, f, o;
// forget_bias 1):   c' = sigmoid(f + 1) * c + sigmoid(i) * tanh(j) ;  h' = tanh(c') * sigmoid(o)
// written with e_x = 2^(-x log2e) so that sigmoid(x) = 1/(1+e_x), tanh|x| = (1-E_x)/(1+E_x),
// E_x = e^(-2|x|) <= 1, and the two quotients of each product share ONE v_rcp_f32:
//   sigmoid(i) tanh(j) = sgn(j) (1-E_j) / ((1+e_i)(1+E_j))
// (5 v_exp + 3 v_rcp per unit instead of 5 + 5; no overflow: e_i = inf -> rcp(inf) = 0).
// The code is STAGE-major over the 5 independent units: MFMA and VALU instructions of one
// SIMD do not execute concurrently on gfx950 (profiles/archive_r01_r03/r01_b_microbench_*), so what matters
// for the VALU part is instruction-level parallelism -- five independent dependency chains
// side by side hide the ~8-cycle transcendental / ~4-cycle VALU latencies in one wave.
__device__ __forceinline__ void lstm_gates5(const f32x4 (&acc)[kNT], float (&c)[kNT], float (&h)[kNT]) {
#ifdef L2O_ABLATE_TRANS
#pragma unroll
  for (int t = 0; t < kNT; ++t) {
    c[t] = sigmoid_p1f_(acc[t][2]) * c[t] + sigmoidf_(acc[t][0]) * tanhf_(acc[t][1]);
    h[t] = tanhf_(c[t]) * sigmoidf_(acc[t][3]);
  }
#else
  constexpr float kL2E = 1.4426950408889634f;
  float e_i[kNT], E_j[kNT], e_f[kNT], e_o[kNT], ij[kNT], rf[kNT], cn[kNT], E_c[kNT], ro[kNT];
#pragma unroll
  for (int t = 0; t < kNT; ++t) e_i[t] = fast_exp2(acc[t][0] * -kL2E);
#pragma unroll
  for (int t = 0; t < kNT; ++t) E_j[t] = fast_exp2(__builtin_fabsf(acc[t][1]) * (-2.0f * kL2E));
#pragma unroll
  for (int t = 0; t < kNT; ++t) e_f[t] = fast_exp2(__builtin_fmaf(acc[t][2], -kL2E, -kL2E));
#pragma unroll
  for (int t = 0; t < kNT; ++t) e_o[t] = fast_exp2(acc[t][3] * -kL2E);
#pragma unroll
  for (int t = 0; t < kNT; ++t) ij[t] = fast_rcp((1.0f + e_i[t]) * (1.0f + E_j[t]));
#pragma unroll
  for (int t = 0; t < kNT; ++t) rf[t] = fast_rcp(1.0f + e_f[t]);
#pragma unroll
  for (int t = 0; t < kNT; ++t)
    cn[t] = __builtin_fmaf(rf[t], c[t], __builtin_copysignf((1.0f - E_j[t]) * ij[t], acc[t][1]));
#pragma unroll
  for (int t = 0; t < kNT; ++t) E_c[t] = fast_exp2(__builtin_fabsf(cn[t]) * (-2.0f * kL2E));
#pragma unroll
  for (int t = 0; t < kNT; ++t) ro[t] = fast_rcp((1.0f + E_c[t]) * (1.0f + e_o[t]));
#pragma unroll
  for (int t = 0; t < kNT; ++t) {
    c[t] = cn[t];
    h[t] = __builtin_copysignf((1.0f - E_c[t]) * ro[t], cn[t]);
  }
#endif
}

// sin / cos of an fp32 angle at libm accuracy in ~14 VALU instructions each (hipcc's sinf / cosf: ~125, most of it a
// Payne-Hanek reduction for arguments the optimizees never produce): n = rint(a 2/pi), three-constant Cody-Waite
// reduction to [-pi/4, pi/4] (exact for |n| < 2^13), the cephes single-precision minimax polynomials, quadrant fix-up.
// Max abs error 9.2e-8 for |a| <= 8192 (libm's fp32 sinf: 7e-8; measured over 2M samples per decade); beyond that the
// libm call (a wave-level branch nobody takes on a converging trajectory).  The rastrigin / square_cos terms
// (DM/problems.py:206-211, 983-991) evaluate cos and sin of the SAME fp32 product 2 pi x as the reference does.
struct SinCos { float s, c; };
__device__ __forceinline__ SinCos sincos_f(float a) {
  SinCos o;
  if (__builtin_expect(__builtin_fabsf(a) > 8192.0f, 0)) {
    o.s = sinf(a); o.c = cosf(a);
    return o;
  }
  const float n = __builtin_rintf(a * 0.6366197723675814f);
  float r = __builtin_fmaf(-n, 1.5703125f, a);
  r = __builtin_fmaf(-n, 4.837512969970703125e-4f, r);
  r = __builtin_fmaf(-n, 7.54978995489188216e-8f, r);
  const float z = r * r;
  const float ps = __builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
  const float sn = __builtin_fmaf(ps * z, r, r);
  const float pc = __builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
  const float cs = __builtin_fmaf(pc * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
  const int q = (int)n;
  const float a0 = (q & 1) ? cs : sn, b0 = (q & 1) ? sn : cs;          // sin, cos up to sign
  o.s = (q & 2) ? -a0 : a0;
  o.c = ((q + 1) & 2) ? -b0 : b0;
  return o;
}
__device__ __forceinline__ float sin_f(float a) { return sincos_f(a).s; }
__device__ __forceinline__ float cos_f(float a) { return sincos_f(a).c; }

// everything that needs this step's gradient:
//   PRE = IDENTITY : in0 = g
//   PRE = LOGSIGN  : in0 = clamped log, in1 = clamped sign  (computed by the caller)
//   PRE = FC_ELU   : in0 = m~, in1 = g~  (RNNProp)
// acc1 must already hold bias + the h1(t-1) part, acc2 bias + the h2(t-1) part.
// With NEXT = true the layer-1 MFMAs of the NEXT step (fed by the h1 just produced) are
// issued into acc1 (dead by then) right behind the layer-2 ones, so that the caller's next
// step starts with them done.
// Returns the Linear output (before tanh / scale), identical on the four q lanes.
template <int PRE, bool NEXT>
__device__ __forceinline__ float lstm_finish(const NetW<PRE>& w, TileState& s, f32x4 (&acc1)[kNT],
                                             f32x4 (&acc2)[kNT], float in0, float in1, int q, PhaseClock& pc) {
  if (PRE == L2O_PRE_FC_ELU) {
    float fc[kNT];
#pragma unroll
    for (int t = 0; t < kNT; ++t)
      fc[t] = eluf_(__builtin_fmaf(w.fcw1[t], in1, __builtin_fmaf(w.fcw0[t], in0, w.fcb[t])));
#pragma unroll
    for (int kk = 0; kk < kNT; ++kk)
#pragma unroll
      for (int t = 0; t < kNT; ++t) acc1[t] = mfma16(w.a1[kk][t], fc[kk], acc1[t]);
  } else {
    // last k-step: q=0 -> feature 0, q=1 -> feature 1, q=2 -> bias (x 1.0), q=3 -> unused
    const float bv = q == 0 ? in0 : (q == 1 ? in1 : (q == 2 ? 1.0f : 0.0f));
#pragma unroll
    for (int t = 0; t < kNT; ++t) acc1[t] = mfma16(w.a1[5][t], bv, acc1[t]);
  }
  pc.drain(acc1);
  pc.mark(5);
  lstm_gates5(acc1, s.c1, s.h1);
  pc.mark(6);
#pragma unroll
  for (int kk = 0; kk < kNT; ++kk)
#pragma unroll
    for (int t = 0; t < kNT; ++t) acc2[t] = mfma16(w.a2[kk][t], s.h1[kk], acc2[t]);
  if (NEXT) lstm_issue_l1_prev<PRE, 0, 25>(w, s, acc1);
  pc.mark(7);
  pc.drain(acc2);
  pc.mark(10);
  lstm_gates5(acc2, s.c2, s.h2);
  pc.mark(8);
  float d0 = s.h2[0] * w.wl[0], d1 = s.h2[1] * w.wl[1];
  d0 = __builtin_fmaf(s.h2[2], w.wl[2], d0);
  d1 = __builtin_fmaf(s.h2[3], w.wl[3], d1);
  d0 = __builtin_fmaf(s.h2[4], w.wl[4], d0);
  const float d = quad_q_sum(d0 + d1);
  return d + w.bl;
}

// One optimizer-network evaluation for a 16-coordinate tile (step-granular kernel).
template <int PRE>
__device__ __forceinline__ float lstm_tile_step(const NetW<PRE>& w, TileState& s, float in0, float in1, int q) {
  f32x4 acc1[kNT], acc2[kNT];
  lstm_issue_l2_prev<PRE, 0, 25>(w, s, acc2);
  lstm_issue_l1_prev<PRE, 0, 25>(w, s, acc1);
  PhaseClock pc;
  return lstm_finish<PRE, false>(w, s, acc1, acc2, in0, in1, q, pc);
}

// Gradient preprocessing -> the (in0, in1) pair fed to lstm_tile_step.
template <int PRE>
__device__ __forceinline__ void preprocess_grad(float g, float k_inv_ln2, float exp_k, float& in0, float& in1) {
  if (PRE == L2O_PRE_LOGSIGN) {
    // DM/preprocess.py:63-70: max(log(|g|+eps)/k, -1) ; clip(g*e^k, -1, 1)
    const float lg = __builtin_amdgcn_logf(__builtin_fabsf(g) + 1.1920928955078125e-07f);   // log2
    in0 = __builtin_fmaxf(lg * k_inv_ln2, -1.0f);
    in1 = __builtin_fminf(__builtin_fmaxf(g * exp_k, -1.0f), 1.0f);
  } else {
    in0 = g;
    in1 = 0.0f;
  }
}

// RNNProp inputs, DM/meta_rnnprop_train.py:383-388.  om1 = 1 - beta1^k, om2 = 1 - beta2^k.
//  omb1 = fp32(1 - beta1) rounded from the python double like TF rounds the constant.
__device__ __forceinline__ void rnnprop_inputs(float g, float& m, float& v, float beta1, float beta2,
                                               float omb1, float omb2, float om1, float om2,
                                               float& m_tilde, float& g_tilde) {
  m = beta1 * m + omb1 * g;
  v = beta2 * v + omb2 * g * g;
  // the two bias corrections and the common denominator through v_rcp_f32 / v_sqrt_f32 (1 ulp each)
  // instead of four IEEE divisions and an IEEE square root (~50 VALU instructions per step): the
  // carried moments m, v are untouched, only the network inputs move by ~1e-7 relative
  const float m_hat = m * fast_rcp(om1);
  const float v_hat = v * fast_rcp(om2);
  const float inv = fast_rcp(__builtin_amdgcn_sqrtf(v_hat) + 1e-8f);
  m_tilde = m_hat * inv;
  g_tilde = g * inv;
}

}  // namespace l2o
