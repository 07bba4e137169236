// l2o_bwd.h -- one step of back-propagation-through-time of the coordinate-wise optimizer
// network (the meta-gradient of MetaOptimizer.meta_minimize, DM/meta.py:398-414, with the
// optimizee gradient treated as a constant: tf.stop_gradient, DM/meta.py:328-329).
// Included by l2o_kernels.hip.
//
// Kernel of the TRAINING path: one thread per coordinate, weights (Sonnet layouts) read as
// scalar loads (wave-uniform addresses -> SGPR operands), the step's forward recomputed from the
// state saved before the step and kept in registers for the backward pass.  It emits, per coordinate,
// the rows the host needs for the weight gradients as plain GEMMs over (steps x coordinates):
//     dW1 = act1^T dz1   db1 = sum dz1        dW2 = act2^T dz2   db2 = sum dz2
//     dw_lin = h2^T dd   db_lin = sum dd      (RNNProp) dW_fc = feats^T du   db_fc = sum du
// and the carries (dh1, dc1, dh2, dc2) that flow to the previous step.
#pragma once

struct BwdParams {
  int B, D, tpp;
  int pre, tanh_output, n_layers;
  float scale, k_inv, exp_k;          // LogAndSign: 1/k, e^k
  float beta1, beta2, om1, om2;       // RNNProp: 1 - beta^k of this step
  const float *wg1, *bg1, *wg2, *bg2, *wl, *bl, *wfc, *bfc;
  const float *g, *m, *v, *st_prev, *dx_next, *carry_in;
  float *carry_out, *act1, *dz1, *act2, *dz2, *h2o, *dd, *feats, *du;
  long s_act1, s_dz1, s_act2, s_dz2, s_h2, s_dd, s_feats, s_du;   // row strides (floats) of the emitted rows
};

// weights are read through the CONSTANT address space: the addresses are wave-uniform, so the
// loads become s_load_dwordx16 and the FMAs take the weight as an SGPR operand (v_fmac v, s, v)
// -- no LDS, no per-lane weight traffic.  (They are never written while the kernel runs.)
typedef const __attribute__((address_space(4))) float* l2o_cfp;

__device__ __forceinline__ float bw_sig(float x) { return l2o::fast_rcp(1.0f + l2o::fast_exp2(x * -1.4426950408889634f)); }
__device__ __forceinline__ float bw_tanh(float x) {
  const float e = l2o::fast_exp2(__builtin_fabsf(x) * -2.8853900817779268f);
  return __builtin_copysignf((1.0f - e) * l2o::fast_rcp(1.0f + e), x);
}

// packed-state address of (array a, unit u) for coordinate (tile, c)
__device__ __forceinline__ size_t st_addr(size_t tile, int c, int a, int u) {
  const int q = u & 3, t = u >> 2, e = a * 5 + t;
  return tile * kStateFloatsPerTile + ((size_t)(e >> 2) * 64 + (q * 16 + c)) * 4 + (e & 3);
}

// One thread per coordinate; everything a coordinate needs lives in its registers (one wave per
// SIMD: up to 512), except the layer input vector, which is indexed by a run-time k and sits in
// a per-thread LDS column.
template <int PRE>
__global__ __launch_bounds__(64) void k_cwlstm_bwd_step(BwdParams p) {
  constexpr int P = PRE == L2O_PRE_FC_ELU ? kH : (PRE == L2O_PRE_LOGSIGN ? 2 : 1);
  constexpr int K1 = P + kH, G = 4 * kH, NT = 64;
  __shared__ float xin[2 * kH * NT];                     // [40][NT] input vector of the current GEMM (k-major)
  const l2o_cfp W1 = (l2o_cfp)p.wg1, b1 = (l2o_cfp)p.bg1, W2 = (l2o_cfp)p.wg2, b2 = (l2o_cfp)p.bg2;
  const l2o_cfp wl = (l2o_cfp)p.wl, wfc = (l2o_cfp)p.wfc, bfc = (l2o_cfp)p.bfc;
  const int tid = threadIdx.x;
  const size_t N = (size_t)p.B * p.D;
  size_t n = (size_t)blockIdx.x * NT + tid;
  const bool valid = n < N;
  if (!valid) n = N - 1;                                  // keep the wave convergent; stores are masked
  const int b = (int)(n / p.D), j = (int)(n - (size_t)b * p.D);
  const size_t tile = (size_t)b * p.tpp + j / kTile;
  const int c = j % kTile;
  float* xi = xin + tid;                                  // element k at xi[k * NT]

  // z[q] = bias[q] + sum_k in[k] W[k][q]   (weights: scalar loads)
  auto gemm = [&](l2o_cfp W, l2o_cfp bias, int KK, float (&z)[G]) {
#pragma unroll
    for (int q = 0; q < G; ++q) z[q] = bias[q];
    for (int k = 0; k < KK; ++k) {
      const float xv = xi[k * NT];
      const l2o_cfp wr = W + k * G;
#pragma unroll
      for (int q = 0; q < G; ++q) z[q] = __builtin_fmaf(xv, wr[q], z[q]);
    }
  };
  // xi[k] = sum_q dz[q] W[k][q]
  auto gemm_t = [&](l2o_cfp W, int KK, const float (&dz)[G]) {
    for (int k = 0; k < KK; ++k) {
      const l2o_cfp wr = W + k * G;
      float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
#pragma unroll
      for (int q = 0; q < G; q += 4) {
        s0 = __builtin_fmaf(dz[q], wr[q], s0);
        s1 = __builtin_fmaf(dz[q + 1], wr[q + 1], s1);
        s2 = __builtin_fmaf(dz[q + 2], wr[q + 2], s2);
        s3 = __builtin_fmaf(dz[q + 3], wr[q + 3], s3);
      }
      xi[k * NT] = (s0 + s1) + (s2 + s3);
    }
  };
  auto activate = [&](float (&z)[G]) {                    // -> (sig i, tanh j, sig(f+1), sig o)
#pragma unroll
    for (int u = 0; u < kH; ++u) {
      z[u] = bw_sig(z[u]);
      z[kH + u] = bw_tanh(z[kH + u]);
      z[2 * kH + u] = bw_sig(z[2 * kH + u] + 1.0f);
      z[3 * kH + u] = bw_sig(z[3 * kH + u]);
    }
  };

  // ---- features -------------------------------------------------------------
  const float gv = p.g[n];
  float pre_fc[PRE == L2O_PRE_FC_ELU ? kH : 1];
  float f0 = 0.0f, f1 = 0.0f;
  if (PRE == L2O_PRE_FC_ELU) {
    const float m_hat = p.m[n] / p.om1, v_hat = p.v[n] / p.om2;
    const float den = sqrtf(v_hat) + 1e-8f;
    f0 = m_hat / den;
    f1 = gv / den;
#pragma unroll
    for (int u = 0; u < kH; ++u) {
      const float zz = f0 * wfc[u] + f1 * wfc[kH + u] + bfc[u];
      pre_fc[u] = zz;
      xi[u * NT] = zz > 0.0f ? zz : expm1f(zz);
    }
  } else if (PRE == L2O_PRE_LOGSIGN) {
    xi[0] = fmaxf(logf(fabsf(gv) + 1.1920928955078125e-07f) * p.k_inv, -1.0f);
    xi[NT] = fminf(fmaxf(gv * p.exp_k, -1.0f), 1.0f);
  } else {
    xi[0] = gv;
  }
  float c1p[kH], c2p[kH];                                 // c1(t-1), c2(t-1)
#pragma unroll
  for (int u = 0; u < kH; ++u) {
    xi[(P + u) * NT] = p.st_prev[st_addr(tile, c, 0, u)]; // h1(t-1)
    c1p[u] = p.st_prev[st_addr(tile, c, 1, u)];
    c2p[u] = p.st_prev[st_addr(tile, c, 3, u)];
  }
  if (valid) {
    float* row = p.act1 + n * p.s_act1;
    for (int k = 0; k < K1; ++k) row[k] = xi[k * NT];
  }

  // ---- layer 1 forward: gates kept in registers for the backward pass ------------------
  float z1[G];
  gemm(W1, b1, K1, z1);
  activate(z1);
  float tc1[kH];
#pragma unroll
  for (int u = 0; u < kH; ++u) {
    const float c1 = z1[2 * kH + u] * c1p[u] + z1[u] * z1[kH + u];
    tc1[u] = bw_tanh(c1);
    xi[u * NT] = tc1[u] * z1[3 * kH + u];                 // h1(t): layer-2 input 0..19
    xi[(kH + u) * NT] = p.st_prev[st_addr(tile, c, 2, u)];   // h2(t-1)
  }
  if (valid) {
    float* row = p.act2 + n * p.s_act2;
    for (int k = 0; k < 2 * kH; ++k) row[k] = xi[k * NT];
  }
  // ---- layer 2 forward + backward ----------------------------------------------------------
  float z2[G];
  gemm(W2, b2, 2 * kH, z2);
  activate(z2);
  const float* cin = p.carry_in;
  {
    float tc2[kH];
    float dlin = p.bl[0];
#pragma unroll
    for (int u = 0; u < kH; ++u) {
      const float c2 = z2[2 * kH + u] * c2p[u] + z2[u] * z2[kH + u];
      tc2[u] = bw_tanh(c2);
      const float h2 = tc2[u] * z2[3 * kH + u];
      if (valid) p.h2o[n * p.s_h2 + u] = h2;
      dlin = __builtin_fmaf(h2, wl[u], dlin);
    }
    float ddv = p.dx_next[n] * p.scale;
    if (p.tanh_output) { const float th = bw_tanh(dlin); ddv *= 1.0f - th * th; }
    if (valid) p.dd[n * p.s_dd] = ddv;
#pragma unroll
    for (int u = 0; u < kH; ++u) {
      const float gi = z2[u], gj = z2[kH + u], gf = z2[2 * kH + u], go = z2[3 * kH + u];
      const float dh2 = ddv * wl[u] + cin[(2 * N + n) * kH + u];
      const float dc2 = cin[(3 * N + n) * kH + u] + dh2 * go * (1.0f - tc2[u] * tc2[u]);
      if (valid) p.carry_out[(3 * N + n) * kH + u] = dc2 * gf;
      z2[u] = dc2 * gj * gi * (1.0f - gi);
      z2[kH + u] = dc2 * gi * (1.0f - gj * gj);
      z2[2 * kH + u] = dc2 * c2p[u] * gf * (1.0f - gf);
      z2[3 * kH + u] = dh2 * tc2[u] * go * (1.0f - go);
    }
  }
  if (valid) {
    float* row = p.dz2 + n * p.s_dz2;
#pragma unroll
    for (int q = 0; q < G; ++q) row[q] = z2[q];
  }
  gemm_t(W2, 2 * kH, z2);                                 // d[h1; h2(t-1)] = dz2 . W2^T
  // ---- layer 1 backward ------------------------------------------------------------------------
#pragma unroll
  for (int u = 0; u < kH; ++u) {
    const float dh1 = xi[u * NT] + cin[(0 * N + n) * kH + u];
    if (valid) p.carry_out[(2 * N + n) * kH + u] = xi[(kH + u) * NT];
    const float gi = z1[u], gj = z1[kH + u], gf = z1[2 * kH + u], go = z1[3 * kH + u];
    const float dc1 = cin[(1 * N + n) * kH + u] + dh1 * go * (1.0f - tc1[u] * tc1[u]);
    if (valid) p.carry_out[(1 * N + n) * kH + u] = dc1 * gf;
    z1[u] = dc1 * gj * gi * (1.0f - gi);
    z1[kH + u] = dc1 * gi * (1.0f - gj * gj);
    z1[2 * kH + u] = dc1 * c1p[u] * gf * (1.0f - gf);
    z1[3 * kH + u] = dh1 * tc1[u] * go * (1.0f - go);
  }
  if (valid) {
    float* row = p.dz1 + n * p.s_dz1;
#pragma unroll
    for (int q = 0; q < G; ++q) row[q] = z1[q];
  }
  gemm_t(W1, K1, z1);                                     // d[inputs; h1(t-1)] = dz1 . W1^T
  if (valid) {
#pragma unroll
    for (int u = 0; u < kH; ++u) p.carry_out[(0 * N + n) * kH + u] = xi[(P + u) * NT];
    if (PRE == L2O_PRE_FC_ELU) {
      p.feats[n * p.s_feats] = f0;
      p.feats[n * p.s_feats + 1] = f1;
#pragma unroll
      for (int u = 0; u < kH; ++u)
        p.du[n * p.s_du + u] = xi[u * NT] * (pre_fc[u] > 0.0f ? 1.0f : expf(pre_fc[u]));
    }
  }
}

// layers == (): delta = scale * (tanh)(w . feats + b): emit feats rows and dd
template <int PRE>
__global__ void k_linear_bwd_step(BwdParams p) {
  const size_t n = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t N = (size_t)p.B * p.D;
  if (n >= N) return;
  const float gv = p.g[n];
  float a0 = gv, a1 = 0.0f;
  if (PRE == L2O_PRE_LOGSIGN) {
    a0 = fmaxf(logf(fabsf(gv) + 1.1920928955078125e-07f) * p.k_inv, -1.0f);
    a1 = fminf(fmaxf(gv * p.exp_k, -1.0f), 1.0f);
  }
  p.act1[n * p.s_act1] = a0;
  p.act1[n * p.s_act1 + 1] = a1;
  float ddv = p.dx_next[n] * p.scale;
  if (p.tanh_output) {
    const float th = tanhf(a0 * p.wl[0] + a1 * (PRE == L2O_PRE_LOGSIGN ? p.wl[1] : 0.0f) + p.bl[0]);
    ddv *= 1.0f - th * th;
  }
  p.dd[n * p.s_dd] = ddv;
}
