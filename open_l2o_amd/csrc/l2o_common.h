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

#include "../../include/l2o_abi.h"

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

__device__ __forceinline__ float sigmoidf_(float x) {
  // 1 / (1 + e^-x) ; e^-x = 2^(-x*log2e)
  return fast_rcp(1.0f + fast_exp2(x * -1.4426950408889634f));
}
// sigmoid(x + 1)  (snt.LSTM forget_bias = 1.0) with the +1 folded into the FMA
__device__ __forceinline__ float sigmoid_p1f_(float x) {
  return fast_rcp(1.0f + fast_exp2(__builtin_fmaf(x, -1.4426950408889634f, -1.4426950408889634f)));
}
__device__ __forceinline__ float tanhf_(float x) {
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

__device__ __forceinline__ float wave_sum64(float v) {
  v += __shfl_xor(v, 32);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 1);
  return v;
}
// sum over the four q lanes that share a coordinate c (lanes c, c+16, c+32, c+48)
__device__ __forceinline__ float quad_q_sum(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
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
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// LSTM nonlinearity for the 5 units a lane owns (snt.LSTM: i, j, f, o; forget_bias 1).
__device__ __forceinline__ void lstm_gates(const f32x4 (&acc)[kNT], float (&c)[kNT], float (&h)[kNT]) {
#pragma unroll
  for (int t = 0; t < kNT; ++t) {
    const float gi = acc[t][0], gj = acc[t][1], gf = acc[t][2], go = acc[t][3];
    const float cn = sigmoid_p1f_(gf) * c[t] + sigmoidf_(gi) * tanhf_(gj);
    c[t] = cn;
    h[t] = tanhf_(cn) * sigmoidf_(go);
  }
}

// One optimizer-network evaluation for a 16-coordinate tile.
//   PRE = IDENTITY : in0 = g
//   PRE = LOGSIGN  : in0 = clamped log, in1 = clamped sign  (computed by the caller)
//   PRE = FC_ELU   : in0 = m~, in1 = g~  (RNNProp)
// Returns the Linear output (before tanh / scale), identical on the four q lanes.
template <int PRE>
__device__ __forceinline__ float lstm_tile_step(const NetW<PRE>& w, TileState& s, float in0, float in1, int q) {
  f32x4 acc1[kNT], acc2[kNT];
  // layer-2 contribution of the PREVIOUS h2: independent of layer 1, issue first
#pragma unroll
  for (int t = 0; t < kNT; ++t) acc2[t] = w.b2[t];
#pragma unroll
  for (int kk = 0; kk < kNT; ++kk)
#pragma unroll
    for (int t = 0; t < kNT; ++t) acc2[t] = mfma16(w.a2[5 + kk][t], s.h2[kk], acc2[t]);

  if (PRE == L2O_PRE_FC_ELU) {
    float fc[kNT];
#pragma unroll
    for (int t = 0; t < kNT; ++t)
      fc[t] = eluf_(__builtin_fmaf(w.fcw1[t], in1, __builtin_fmaf(w.fcw0[t], in0, w.fcb[t])));
#pragma unroll
    for (int t = 0; t < kNT; ++t) acc1[t] = w.b1[t];
#pragma unroll
    for (int kk = 0; kk < kNT; ++kk)
#pragma unroll
      for (int t = 0; t < kNT; ++t) acc1[t] = mfma16(w.a1[5 + kk][t], s.h1[kk], acc1[t]);
#pragma unroll
    for (int kk = 0; kk < kNT; ++kk)
#pragma unroll
      for (int t = 0; t < kNT; ++t) acc1[t] = mfma16(w.a1[kk][t], fc[kk], acc1[t]);
  } else {
#pragma unroll
    for (int t = 0; t < kNT; ++t) acc1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < kNT; ++kk)
#pragma unroll
      for (int t = 0; t < kNT; ++t) acc1[t] = mfma16(w.a1[kk][t], s.h1[kk], acc1[t]);
    // last k-step: q=0 -> feature 0, q=1 -> feature 1, q=2 -> bias (x 1.0), q=3 -> unused
    const float bv = q == 0 ? in0 : (q == 1 ? in1 : (q == 2 ? 1.0f : 0.0f));
#pragma unroll
    for (int t = 0; t < kNT; ++t) acc1[t] = mfma16(w.a1[5][t], bv, acc1[t]);
  }
  lstm_gates(acc1, s.c1, s.h1);

#pragma unroll
  for (int kk = 0; kk < kNT; ++kk)
#pragma unroll
    for (int t = 0; t < kNT; ++t) acc2[t] = mfma16(w.a2[kk][t], s.h1[kk], acc2[t]);
  lstm_gates(acc2, s.c2, s.h2);

  float d = 0.0f;
#pragma unroll
  for (int t = 0; t < kNT; ++t) d = __builtin_fmaf(s.h2[t], w.wl[t], d);
  d = quad_q_sum(d);
  return d + w.bl;
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
  const float m_hat = m / om1;
  const float v_hat = v / om2;
  const float den = __builtin_sqrtf(v_hat) + 1e-8f;
  m_tilde = m_hat / den;
  g_tilde = g / den;
}

}  // namespace l2o
