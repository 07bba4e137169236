// l2o_mlp.h -- the neural optimizee of the reference harness: problems.mnist
// (DM/problems.py:246-288), a [n_in -> n_hidden -> n_out] MLP with mean sparse-softmax
// cross-entropy on a gathered minibatch, forward + gradient in ONE single-workgroup launch
// (8 MFLOP: latency-, not throughput-bound; the LSTM step on its 15 910 coordinates runs
// in k_cwlstm_step).  Included by l2o_kernels.hip.
#pragma once

constexpr int kMlpThreads = 1024;
constexpr int kMlpMaxH = 32;
constexpr int kMlpMaxO = 16;

struct MlpParams {
  int n_in, H, O, batch, act;
  const float* images;
  const int* labels;
  const int* idx;
  const float *w1, *b1, *w2, *b2;
  float* loss;
  float *gw1, *gb1, *gw2, *gb2;
};

__global__ __launch_bounds__(kMlpThreads) void k_mlp_fg(MlpParams p) {
  extern __shared__ float sm[];
  const int n_in = p.n_in, H = p.H, O = p.O, Bn = p.batch;
  const int tid = threadIdx.x;
  const int G = kMlpThreads / Bn;                       // k-slices of the input layer
  float* w1s = sm;                                      // [n_in][H]
  float* P = w1s + n_in * H;                            // [G][Bn][H] partial pre-activations (phase A)
  float* dHs = P;                                       // [Bn][H]   d loss / d pre-activation  } alias P,
  float* dZs = P + Bn * H;                              // [Bn][O]                               } dead after A
  const int psz = G * Bn * H > Bn * (H + O) ? G * Bn * H : Bn * (H + O);
  float* Hs = P + psz;                                  // [Bn][H]   activations
  float* w2s = Hs + Bn * H;                             // [H][O]
  float* lossn = w2s + H * O;                           // [Bn]
  int* rows = reinterpret_cast<int*>(lossn + Bn);       // [Bn]  gathered row index
  for (int i = tid; i < n_in * H; i += kMlpThreads) w1s[i] = p.w1[i];
  for (int i = tid; i < H * O; i += kMlpThreads) w2s[i] = p.w2[i];
  for (int i = tid; i < Bn; i += kMlpThreads) rows[i] = p.idx[i];
  __syncthreads();

  // ---- A: hidden pre-activation, thread = (sample n, k-slice g) ----------------
  {
    const int n = tid % Bn, g = tid / Bn;
    if (g < G) {
      const int ks = (n_in + G - 1) / G, k0 = g * ks, k1 = min(n_in, k0 + ks);
      const float* xrow = p.images + (size_t)rows[n] * n_in;
      float acc[kMlpMaxH];
#pragma unroll
      for (int u = 0; u < kMlpMaxH; ++u) acc[u] = 0.0f;
      for (int k = k0; k < k1; ++k) {
        const float xv = xrow[k];
        const float* wr = w1s + k * H;
#pragma unroll
        for (int u = 0; u < kMlpMaxH; ++u)
          if (u < H) acc[u] = __builtin_fmaf(xv, wr[u], acc[u]);
      }
#pragma unroll
      for (int u = 0; u < kMlpMaxH; ++u)
        if (u < H) P[(g * Bn + n) * H + u] = acc[u];
    }
  }
  __syncthreads();
  for (int e = tid; e < Bn * H; e += kMlpThreads) {
    const int u = e % H;
    float a = p.b1[u];
    for (int g = 0; g < G; ++g) a += P[g * Bn * H + e];
    Hs[e] = p.act == 0 ? 1.0f / (1.0f + expf(-a)) : fmaxf(a, 0.0f);
  }
  __syncthreads();
  // ---- B: logits, softmax cross-entropy, dZ (thread = sample) -----------------
  if (tid < Bn) {
    const int n = tid;
    float z[kMlpMaxO];
    float zmax = -3.0e38f;
#pragma unroll
    for (int o = 0; o < kMlpMaxO; ++o) {
      if (o < O) {
        float a = p.b2[o];
        for (int u = 0; u < H; ++u) a = __builtin_fmaf(Hs[n * H + u], w2s[u * O + o], a);
        z[o] = a;
        zmax = fmaxf(zmax, a);
      }
    }
    float se = 0.0f;
#pragma unroll
    for (int o = 0; o < kMlpMaxO; ++o)
      if (o < O) se += expf(z[o] - zmax);
    const int lab = p.labels[rows[n]];
    const float lse = zmax + logf(se);
    float zl = 0.0f;
    const float inv = 1.0f / (float)Bn;
#pragma unroll
    for (int o = 0; o < kMlpMaxO; ++o) {
      if (o < O) {
        if (o == lab) zl = z[o];
        dZs[n * O + o] = (expf(z[o] - lse) - (o == lab ? 1.0f : 0.0f)) * inv;
      }
    }
    lossn[n] = lse - zl;
  }
  __syncthreads();
  if (tid == 0) {
    float s = 0.0f;
    for (int n = 0; n < Bn; ++n) s += lossn[n];           // fixed order
    p.loss[0] = s / (float)Bn;
  }
  if (p.gw1 == nullptr) return;
  // ---- C: gw2, gb2, dH ----------------------------------------------------------
  for (int e = tid; e < H * O; e += kMlpThreads) {
    const int u = e / O, o = e % O;
    float a = 0.0f;
    for (int n = 0; n < Bn; ++n) a = __builtin_fmaf(Hs[n * H + u], dZs[n * O + o], a);
    p.gw2[e] = a;
  }
  for (int o = tid; o < O; o += kMlpThreads) {
    float a = 0.0f;
    for (int n = 0; n < Bn; ++n) a += dZs[n * O + o];
    p.gb2[o] = a;
  }
  for (int e = tid; e < Bn * H; e += kMlpThreads) {
    const int n = e / H, u = e % H;
    float a = 0.0f;
    for (int o = 0; o < O; ++o) a = __builtin_fmaf(dZs[n * O + o], w2s[u * O + o], a);
    const float h = Hs[e];
    dHs[e] = p.act == 0 ? a * h * (1.0f - h) : (h > 0.0f ? a : 0.0f);
  }
  __syncthreads();
  // ---- D: gw1 (thread = input k: coalesced image rows), gb1 ----------------------
  for (int k = tid; k < n_in; k += kMlpThreads) {
    float acc[kMlpMaxH];
#pragma unroll
    for (int u = 0; u < kMlpMaxH; ++u) acc[u] = 0.0f;
    for (int n = 0; n < Bn; ++n) {
      const float xv = p.images[(size_t)rows[n] * n_in + k];
      const float* dh = dHs + n * H;
#pragma unroll
      for (int u = 0; u < kMlpMaxH; ++u)
        if (u < H) acc[u] = __builtin_fmaf(xv, dh[u], acc[u]);
    }
#pragma unroll
    for (int u = 0; u < kMlpMaxH; ++u)
      if (u < H) p.gw1[k * H + u] = acc[u];
  }
  for (int u = tid; u < H; u += kMlpThreads) {
    float a = 0.0f;
    for (int n = 0; n < Bn; ++n) a += dHs[n * H + u];
    p.gb1[u] = a;
  }
}
