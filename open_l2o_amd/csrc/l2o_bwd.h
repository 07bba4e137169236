// l2o_bwd.h -- one step of back-propagation-through-time of the coordinate-wise optimizer
// network (the meta-gradient of MetaOptimizer.meta_minimize, DM/meta.py:398-414, with the
// optimizee gradient treated as a constant: tf.stop_gradient, DM/meta.py:328-329).
// Included by l2o_kernels.hip.
//
// Correctness-first kernel for the TRAINING path (not the benchmarked inner loop): one thread
// per coordinate, weights (Sonnet layouts) staged in LDS and read as broadcasts, the step's
// forward is recomputed from the state saved before the step.  It emits, per coordinate,
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
};

__device__ __forceinline__ float sg(float x) { return 1.0f / (1.0f + expf(-x)); }

// packed-state address of (array a, unit u) for coordinate (tile, c)
__device__ __forceinline__ size_t st_addr(size_t tile, int c, int a, int u) {
  const int q = u & 3, t = u >> 2, e = a * 5 + t;
  return tile * kStateFloatsPerTile + ((size_t)(e >> 2) * 64 + (q * 16 + c)) * 4 + (e & 3);
}

template <int PRE>
__global__ __launch_bounds__(64) void k_cwlstm_bwd_step(BwdParams p) {
  constexpr int P = PRE == L2O_PRE_FC_ELU ? kH : (PRE == L2O_PRE_LOGSIGN ? 2 : 1);
  constexpr int K1 = P + kH, G = 4 * kH, NT = 64;
  extern __shared__ float sm[];
  float* W1 = sm;                 // [K1][80]
  float* b1 = W1 + K1 * G;        // [80]
  float* W2 = b1 + G;             // [40][80]
  float* b2 = W2 + 2 * kH * G;    // [80]
  float* wl = b2 + G;             // [20]
  float* wfc = wl + kH;           // [2][20]
  float* bfc = wfc + 2 * kH;      // [20]
  float* xin = bfc + kH;          // [40][NT]  per-thread input vector of the current layer (k-major)
  const int tid = threadIdx.x;
  for (int i = tid; i < K1 * G; i += NT) W1[i] = p.wg1[i];
  for (int i = tid; i < 2 * kH * G; i += NT) W2[i] = p.wg2[i];
  for (int i = tid; i < G; i += NT) { b1[i] = p.bg1[i]; b2[i] = p.bg2[i]; }
  for (int i = tid; i < kH; i += NT) {
    wl[i] = p.wl[i];
    if (PRE == L2O_PRE_FC_ELU) { wfc[i] = p.wfc[i]; wfc[kH + i] = p.wfc[kH + i]; bfc[i] = p.bfc[i]; }
  }
  __syncthreads();
  const size_t N = (size_t)p.B * p.D;
  size_t n = (size_t)blockIdx.x * NT + tid;
  const bool valid = n < N;
  if (!valid) n = N - 1;                                  // keep the workgroup convergent; stores are masked
  const int b = (int)(n / p.D), j = (int)(n - (size_t)b * p.D);
  const size_t tile = (size_t)b * p.tpp + j / kTile;
  const int c = j % kTile;
  float* xi = xin + tid;                                  // element k at xi[k * NT]

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
#pragma unroll
  for (int u = 0; u < kH; ++u) xi[(P + u) * NT] = p.st_prev[st_addr(tile, c, 0, u)];   // h1_{t-1}
  if (valid) {
    for (int k = 0; k < K1; ++k) p.act1[n * K1 + k] = xi[k * NT];
  }

  float z[G];
  // gates of layer L from the vector in xi[0..KK): z <- (sig i, tanh j, sig(f+1), sig o)
  auto gates = [&](const float* W, const float* bb, int KK) {
#pragma unroll
    for (int q = 0; q < G; ++q) z[q] = bb[q];
    for (int k = 0; k < KK; ++k) {
      const float xv = xi[k * NT];
      const float* wr = W + k * G;
#pragma unroll
      for (int q = 0; q < G; ++q) z[q] = fmaf(xv, wr[q], z[q]);
    }
#pragma unroll
    for (int u = 0; u < kH; ++u) {
      z[u] = sg(z[u]);
      z[kH + u] = tanhf(z[kH + u]);
      z[2 * kH + u] = sg(z[2 * kH + u] + 1.0f);
      z[3 * kH + u] = sg(z[3 * kH + u]);
    }
  };
  // ---- layer 1 forward -> c1, h1 -----------------------------------------------
  float c1[kH], h1[kH];
  gates(W1, b1, K1);
#pragma unroll
  for (int u = 0; u < kH; ++u) {
    c1[u] = z[2 * kH + u] * p.st_prev[st_addr(tile, c, 1, u)] + z[u] * z[kH + u];
    h1[u] = tanhf(c1[u]) * z[3 * kH + u];
  }
  // ---- layer 2 forward ---------------------------------------------------------
#pragma unroll
  for (int u = 0; u < kH; ++u) {
    xi[u * NT] = h1[u];
    xi[(kH + u) * NT] = p.st_prev[st_addr(tile, c, 2, u)];   // h2_{t-1}
  }
  if (valid) {
    for (int k = 0; k < 2 * kH; ++k) p.act2[n * 2 * kH + k] = xi[k * NT];
  }
  gates(W2, b2, 2 * kH);
  float dh1[kH];
  {
    const float* cin = p.carry_in;
    float tc2[kH];
    float dlin = p.bl[0];
#pragma unroll
    for (int u = 0; u < kH; ++u) {
      const float c2 = z[2 * kH + u] * p.st_prev[st_addr(tile, c, 3, u)] + z[u] * z[kH + u];
      tc2[u] = tanhf(c2);
      const float h2 = tc2[u] * z[3 * kH + u];
      if (valid) p.h2o[n * kH + u] = h2;
      dlin = fmaf(h2, wl[u], dlin);
    }
    float ddv = p.dx_next[n] * p.scale;
    if (p.tanh_output) { const float th = tanhf(dlin); ddv *= 1.0f - th * th; }
    if (valid) p.dd[n] = ddv;
    // ---- layer 2 backward -------------------------------------------------------
#pragma unroll
    for (int u = 0; u < kH; ++u) {
      const float gi = z[u], gj = z[kH + u], gf = z[2 * kH + u], go = z[3 * kH + u];
      const float dh2 = ddv * wl[u] + cin[(2 * N + n) * kH + u];
      const float dc2 = cin[(3 * N + n) * kH + u] + dh2 * go * (1.0f - tc2[u] * tc2[u]);
      if (valid) p.carry_out[(3 * N + n) * kH + u] = dc2 * gf;
      z[u] = dc2 * gj * gi * (1.0f - gi);
      z[kH + u] = dc2 * gi * (1.0f - gj * gj);
      z[2 * kH + u] = dc2 * p.st_prev[st_addr(tile, c, 3, u)] * gf * (1.0f - gf);
      z[3 * kH + u] = dh2 * tc2[u] * go * (1.0f - go);
    }
    if (valid) {
#pragma unroll
      for (int q = 0; q < G; ++q) p.dz2[n * G + q] = z[q];
    }
    // d[h1; h2_prev] = dz2 . W2^T
    for (int k = 0; k < 2 * kH; ++k) {
      const float* wr = W2 + k * G;
      float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
      for (int q = 0; q < G; q += 2) { s0 = fmaf(z[q], wr[q], s0); s1 = fmaf(z[q + 1], wr[q + 1], s1); }
      xi[k * NT] = s0 + s1;
    }
#pragma unroll
    for (int u = 0; u < kH; ++u) {
      dh1[u] = xi[u * NT] + cin[(0 * N + n) * kH + u];
      if (valid) p.carry_out[(2 * N + n) * kH + u] = xi[(kH + u) * NT];
    }
  }
  // ---- layer 1 backward (gates recomputed from act1) ---------------------------------
  for (int k = 0; k < K1; ++k) xi[k * NT] = p.act1[n * K1 + k];   // written above by this thread (or row N-1's owner)
  gates(W1, b1, K1);
  {
    const float* cin = p.carry_in;
#pragma unroll
    for (int u = 0; u < kH; ++u) {
      const float gi = z[u], gj = z[kH + u], gf = z[2 * kH + u], go = z[3 * kH + u];
      const float tc1 = tanhf(c1[u]);
      const float dc1 = cin[(1 * N + n) * kH + u] + dh1[u] * go * (1.0f - tc1 * tc1);
      if (valid) p.carry_out[(1 * N + n) * kH + u] = dc1 * gf;
      z[u] = dc1 * gj * gi * (1.0f - gi);
      z[kH + u] = dc1 * gi * (1.0f - gj * gj);
      z[2 * kH + u] = dc1 * p.st_prev[st_addr(tile, c, 1, u)] * gf * (1.0f - gf);
      z[3 * kH + u] = dh1[u] * tc1 * go * (1.0f - go);
    }
    if (valid) {
#pragma unroll
      for (int q = 0; q < G; ++q) p.dz1[n * G + q] = z[q];
    }
    for (int k = 0; k < K1; ++k) {
      const float* wr = W1 + k * G;
      float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
      for (int q = 0; q < G; q += 2) { s0 = fmaf(z[q], wr[q], s0); s1 = fmaf(z[q + 1], wr[q + 1], s1); }
      xi[k * NT] = s0 + s1;
    }
    if (valid) {
#pragma unroll
      for (int u = 0; u < kH; ++u) p.carry_out[(0 * N + n) * kH + u] = xi[(P + u) * NT];
      if (PRE == L2O_PRE_FC_ELU) {
        p.feats[n * 2] = f0;
        p.feats[n * 2 + 1] = f1;
#pragma unroll
        for (int u = 0; u < kH; ++u) p.du[n * kH + u] = xi[u * NT] * (pre_fc[u] > 0.0f ? 1.0f : expf(pre_fc[u]));
      }
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
  p.act1[n * 2] = a0;
  p.act1[n * 2 + 1] = a1;
  float ddv = p.dx_next[n] * p.scale;
  if (p.tanh_output) {
    const float th = tanhf(a0 * p.wl[0] + a1 * (PRE == L2O_PRE_LOGSIGN ? p.wl[1] : 0.0f) + p.bl[0]);
    ddv *= 1.0f - th * th;
  }
  p.dd[n] = ddv;
}
