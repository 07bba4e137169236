// l2o_mlp.h -- the neural optimizee of the reference harness: problems.mnist
// (DM/problems.py:246-288), a [n_in -> n_hidden -> n_out] MLP with mean sparse-softmax
// cross-entropy on a gathered minibatch: forward + gradient in two small multi-workgroup
// launches (8 MFLOP per evaluation: latency-, not throughput-bound; the LSTM step on its
// 15 910 coordinates runs in k_cwlstm_step).  Included by l2o_kernels.hip.
//   k_mlp_fwd : 8 samples per workgroup, 32 threads per sample over the 784 inputs
//               -> H, dZ = (softmax - onehot)/batch, dH, per-sample loss (scratch in HBM)
//   k_mlp_bwd : 64 input rows of gw1 per workgroup (coalesced image rows), one extra
//               workgroup for gw2, gb2, gb1 and the fixed-order loss sum
#pragma once

constexpr int kMlpMaxH = 32;
constexpr int kMlpMaxO = 16;
constexpr int kMlpSPB = 8;        // samples per forward workgroup
constexpr int kMlpKPB = 64;       // gw1 rows per backward workgroup

struct MlpParams {
  int n_in, H, O, batch, act;
  const float* images;
  const int* labels;
  const int* idx;
  const float *w1, *b1, *w2, *b2;
  float* loss;
  float *gw1, *gb1, *gw2, *gb2;
  float* scratch;                 // [batch*H] H | [batch*O] dZ | [batch*H] dH | [batch] loss_n
};

__global__ __launch_bounds__(256) void k_mlp_fwd(MlpParams p) {
  extern __shared__ float sm[];
  const int n_in = p.n_in, H = p.H, O = p.O, Bn = p.batch;
  const int HS = H | 1;                                 // odd row stride: conflict-free w1 reads
  float* w1s = sm;                                      // [n_in][HS]
  float* part = w1s + n_in * HS;                        // [SPB][32][H]
  float* hs = part + kMlpSPB * 32 * H;                  // [SPB][H]
  float* zs = hs + kMlpSPB * H;                         // [SPB][O]
  const int tid = threadIdx.x;
  for (int i = tid; i < n_in * H; i += 256) w1s[(i / H) * HS + (i % H)] = p.w1[i];
  __syncthreads();
  const int sl = tid >> 5, l32 = tid & 31;
  const int n = blockIdx.x * kMlpSPB + sl;
  const bool valid = n < Bn;
  const int row = valid ? p.idx[n] : 0;
  {
    const float* xrow = p.images + (size_t)row * n_in;
    float acc[kMlpMaxH];
#pragma unroll
    for (int u = 0; u < kMlpMaxH; ++u) acc[u] = 0.0f;
    for (int k = l32; k < n_in; k += 32) {
      const float xv = xrow[k];
      const float* wr = w1s + k * HS;
#pragma unroll
      for (int u = 0; u < kMlpMaxH; ++u)
        if (u < H) acc[u] = __builtin_fmaf(xv, wr[u], acc[u]);
    }
#pragma unroll
    for (int u = 0; u < kMlpMaxH; ++u)
      if (u < H) part[(sl * 32 + l32) * H + u] = acc[u];
  }
  __syncthreads();
  float* gH = p.scratch;
  float* gdZ = gH + Bn * H;
  float* gdH = gdZ + Bn * O;
  float* gloss = gdH + Bn * H;
  if (tid < kMlpSPB * H) {                              // hidden activation: thread = (sample, unit)
    const int s2 = tid / H, u = tid % H;
    float a = p.b1[u];
    for (int l = 0; l < 32; ++l) a += part[(s2 * 32 + l) * H + u];
    hs[s2 * H + u] = p.act == 0 ? 1.0f / (1.0f + expf(-a)) : fmaxf(a, 0.0f);
  }
  __syncthreads();
  if (tid < kMlpSPB * O) {                              // logits: thread = (sample, class)
    const int s2 = tid / O, o = tid % O;
    float a = p.b2[o];
    for (int u = 0; u < H; ++u) a = __builtin_fmaf(hs[s2 * H + u], p.w2[u * O + o], a);
    zs[s2 * O + o] = a;
  }
  __syncthreads();
  if (tid < kMlpSPB) {                                  // softmax cross-entropy: thread = sample
    const int n2 = blockIdx.x * kMlpSPB + tid;
    if (n2 < Bn) {
      const float* z = zs + tid * O;
      float zmax = z[0];
      for (int o = 1; o < O; ++o) zmax = fmaxf(zmax, z[o]);
      float se = 0.0f;
      for (int o = 0; o < O; ++o) se += expf(z[o] - zmax);
      const float lse = zmax + logf(se);
      const int lab = p.labels[p.idx[n2]];
      const float inv = 1.0f / (float)Bn;
      gloss[n2] = lse - z[lab];                         // before z is overwritten by dZ
      for (int o = 0; o < O; ++o) {
        const float d = (expf(z[o] - lse) - (o == lab ? 1.0f : 0.0f)) * inv;
        zs[tid * O + o] = d;                            // reuse as dZ
        gdZ[n2 * O + o] = d;
      }
    }
  }
  __syncthreads();
  if (tid < kMlpSPB * H) {                              // dH and H to HBM: thread = (sample, unit)
    const int s2 = tid / H, u = tid % H;
    const int n2 = blockIdx.x * kMlpSPB + s2;
    if (n2 < Bn) {
      float a = 0.0f;
      for (int o = 0; o < O; ++o) a = __builtin_fmaf(zs[s2 * O + o], p.w2[u * O + o], a);
      const float h = hs[s2 * H + u];
      gH[n2 * H + u] = h;
      gdH[n2 * H + u] = p.act == 0 ? a * h * (1.0f - h) : (h > 0.0f ? a : 0.0f);
    }
  }
}

__global__ __launch_bounds__(256) void k_mlp_bwd(MlpParams p) {
  extern __shared__ float sm[];
  const int n_in = p.n_in, H = p.H, O = p.O, Bn = p.batch;
  const float* gH = p.scratch;
  const float* gdZ = gH + Bn * H;
  const float* gdH = gdZ + Bn * O;
  const float* gloss = gdH + Bn * H;
  const int tid = threadIdx.x;
  const int nkb = (n_in + kMlpKPB - 1) / kMlpKPB;
  if ((int)blockIdx.x == nkb) {                         // the small tensors + the loss
    if (tid == 0) {
      float s = 0.0f;
      for (int n = 0; n < Bn; ++n) s += gloss[n];       // fixed order
      p.loss[0] = s / (float)Bn;
    }
    if (p.gw1 == nullptr) return;
    for (int e = tid; e < H * O; e += 256) {
      const int u = e / O, o = e % O;
      float a = 0.0f;
      for (int n = 0; n < Bn; ++n) a = __builtin_fmaf(gH[n * H + u], gdZ[n * O + o], a);
      p.gw2[e] = a;
    }
    for (int o = tid; o < O; o += 256) {
      float a = 0.0f;
      for (int n = 0; n < Bn; ++n) a += gdZ[n * O + o];
      p.gb2[o] = a;
    }
    for (int u = tid; u < H; u += 256) {
      float a = 0.0f;
      for (int n = 0; n < Bn; ++n) a += gdH[n * H + u];
      p.gb1[u] = a;
    }
    return;
  }
  if (p.gw1 == nullptr) return;
  float* dhs = sm;                                      // [Bn][H]
  int* rows = reinterpret_cast<int*>(dhs + Bn * H);     // [Bn]
  float* part = reinterpret_cast<float*>(rows + Bn);    // [4][KPB][H]
  for (int i = tid; i < Bn * H; i += 256) dhs[i] = gdH[i];
  for (int i = tid; i < Bn; i += 256) rows[i] = p.idx[i];
  __syncthreads();
  const int kl = tid & (kMlpKPB - 1), np = tid >> 6;    // 64 rows x 4 sample-quarters
  const int k = blockIdx.x * kMlpKPB + kl;
  float acc[kMlpMaxH];
#pragma unroll
  for (int u = 0; u < kMlpMaxH; ++u) acc[u] = 0.0f;
  if (k < n_in) {
    for (int n = np; n < Bn; n += 4) {
      const float xv = p.images[(size_t)rows[n] * n_in + k];
      const float* dh = dhs + n * H;
#pragma unroll
      for (int u = 0; u < kMlpMaxH; ++u)
        if (u < H) acc[u] = __builtin_fmaf(xv, dh[u], acc[u]);
    }
  }
#pragma unroll
  for (int u = 0; u < kMlpMaxH; ++u)
    if (u < H) part[(np * kMlpKPB + kl) * H + u] = acc[u];
  __syncthreads();
  for (int e = tid; e < kMlpKPB * H; e += 256) {
    const int kk = e / H, u = e % H;
    const int kg = blockIdx.x * kMlpKPB + kk;
    if (kg < n_in) {
      const float s = (part[(0 * kMlpKPB + kk) * H + u] + part[(1 * kMlpKPB + kk) * H + u]) +
                      (part[(2 * kMlpKPB + kk) * H + u] + part[(3 * kMlpKPB + kk) * H + u]);
      p.gw1[kg * H + u] = s;
    }
  }
}
