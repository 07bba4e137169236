// l2o_mlp_deep.h -- forward + gradient of problems.mnist with MORE THAN ONE hidden layer (DM/problems.py:254-288 builds
// snt.nets.MLP(list(layers) + [10]); DM/util.py:157-163 "mnist_deeper" = layers (20, 20)): the step-granular evaluation
// l2o_mlp_deep_fg.  Included by l2o_kernels.hip; written for gfx950 only.
//
// Not a fast path (the fused unrolls serve the one-hidden-layer optimizee of BASELINE config 5): two launches per
// evaluation, correctness-first.
//   k_mlp_deep_sample   one workgroup per minibatch sample: gather the image row, the forward through the hidden layers
//                       (the 784-wide first layer split over eight k-slices of the workgroup), softmax cross-entropy, the
//                       backward deltas of every layer; activations / deltas / per-sample loss -> scratch
//   k_mlp_deep_grad     one thread per weight / bias coordinate: the sum over the samples of input x delta, in sample order
//                       (fixed order: bit-reproducible); thread 0 of block 0 also adds the per-sample losses
#pragma once

namespace l2o {

constexpr int kMdMaxHidden = 3;          // hidden layers
constexpr int kMdMaxWidth = 32;          // units per hidden layer
constexpr int kMdMaxOut = 16;
constexpr int kMdMaxIn = 1024;
constexpr int kMdThreads = 256;

struct MlpDeepArgs {
  int n_in, n_out, batch, act, nh;       // nh hidden layers
  int width[kMdMaxHidden + 1];           // units of layer l (the last entry: n_out)
  const float* images;
  const int* labels;
  const int* idx;                        // [batch]
  const float* w[kMdMaxHidden + 1];      // layer l: [n_prev, width[l]]
  const float* b[kMdMaxHidden + 1];
  float* gw[kMdMaxHidden + 1];
  float* gb[kMdMaxHidden + 1];
  float* acts;                           // [nh][batch][32]     hidden activations
  float* deltas;                         // [nh + 1][batch][32] dL/d(pre-activation) of every layer (already / batch)
  float* loss_s;                         // [batch]
  float* loss;                           // [1]
  int want_grad;
};

__global__ __launch_bounds__(kMdThreads) void k_mlp_deep_sample(MlpDeepArgs a) {
  __shared__ float xs[kMdMaxIn];
  __shared__ float part[8][kMdMaxWidth];
  __shared__ float act[kMdMaxHidden + 1][kMdMaxWidth];      // act[l]: output of layer l (the last: logits)
  __shared__ float dl[kMdMaxHidden + 1][kMdMaxWidth];
  const int s = blockIdx.x, tid = threadIdx.x;
  const int row = a.idx[s];
  for (int k = tid; k < a.n_in; k += kMdThreads) xs[k] = a.images[(size_t)row * a.n_in + k];
  __syncthreads();
  // layer 0: thread = (unit h, k-slice p): k = p, p + 8, ... ascending, then the eight slices in order
  {
    const int h = tid & 31, p = tid >> 5, H = a.width[0];
    float acc = 0.0f;
    if (h < H)
      for (int k = p; k < a.n_in; k += 8) acc = __builtin_fmaf(xs[k], a.w[0][(size_t)k * H + h], acc);
    part[p][h] = acc;
    __syncthreads();
    if (tid < H) {
      float z = a.b[0][tid];
      for (int q = 0; q < 8; ++q) z += part[q][tid];
      act[0][tid] = a.nh == 0 ? z : (a.act == 0 ? sigmoidf_(z) : fmaxf(z, 0.0f));
    }
    __syncthreads();
  }
  for (int l = 1; l <= a.nh; ++l) {
    const int Hp = a.width[l - 1], H = a.width[l];
    if (tid < H) {
      float z = a.b[l][tid];
      for (int k = 0; k < Hp; ++k) z = __builtin_fmaf(act[l - 1][k], a.w[l][k * H + tid], z);
      act[l][tid] = l == a.nh ? z : (a.act == 0 ? sigmoidf_(z) : fmaxf(z, 0.0f));
    }
    __syncthreads();
  }
  // softmax cross-entropy of this sample (DM/problems.py:41-46, mean over the minibatch)
  const int O = a.n_out, L = a.nh;
  const float invB = 1.0f / (float)a.batch;
  if (tid == 0) {
    const int lab = a.labels[row];
    float zmax = act[L][0];
    for (int o = 1; o < O; ++o) zmax = fmaxf(zmax, act[L][o]);
    float se = 0.0f;
    for (int o = 0; o < O; ++o) se += expf(act[L][o] - zmax);
    const float lse = zmax + logf(se);
    a.loss_s[s] = lse - act[L][lab];
    for (int o = 0; o < O; ++o) dl[L][o] = (expf(act[L][o] - lse) - (o == lab ? 1.0f : 0.0f)) * invB;
  }
  __syncthreads();
  if (!a.want_grad) return;
  for (int l = L - 1; l >= 0; --l) {                       // delta_l = (W_{l+1} delta_{l+1}) * act'(a_l)
    const int H = a.width[l], Hn = a.width[l + 1];
    if (tid < H) {
      float d = 0.0f;
      for (int o = 0; o < Hn; ++o) d = __builtin_fmaf(dl[l + 1][o], a.w[l + 1][tid * Hn + o], d);
      const float hv = act[l][tid];
      dl[l][tid] = a.act == 0 ? d * hv * (1.0f - hv) : (hv > 0.0f ? d : 0.0f);
    }
    __syncthreads();
  }
  for (int l = 0; l <= L; ++l)
    if (tid < a.width[l]) {
      a.deltas[((size_t)l * a.batch + s) * kMdMaxWidth + tid] = dl[l][tid];
      if (l < L) a.acts[((size_t)l * a.batch + s) * kMdMaxWidth + tid] = act[l][tid];
    }
}

__global__ __launch_bounds__(kMdThreads) void k_mlp_deep_grad(MlpDeepArgs a) {
  long i = (long)blockIdx.x * kMdThreads + threadIdx.x;
  if (i == 0) {                                            // the mean loss: per-sample terms in sample order
    float t = 0.0f;
    for (int s = 0; s < a.batch; ++s) t += a.loss_s[s];
    a.loss[0] = t / (float)a.batch;
  }
  if (!a.want_grad) return;
  for (int l = 0; l <= a.nh; ++l) {
    const int np = l == 0 ? a.n_in : a.width[l - 1], H = a.width[l];
    const long nw = (long)np * H;
    if (i < nw) {
      const int k = (int)(i / H), h = (int)(i - (long)k * H);
      float g = 0.0f;
      for (int s = 0; s < a.batch; ++s) {
        const float in = l == 0 ? a.images[(size_t)a.idx[s] * a.n_in + k] : a.acts[((size_t)(l - 1) * a.batch + s) * kMdMaxWidth + k];
        g = __builtin_fmaf(in, a.deltas[((size_t)l * a.batch + s) * kMdMaxWidth + h], g);
      }
      a.gw[l][i] = g;
      return;
    }
    i -= nw;
    if (i < H) {
      float g = 0.0f;
      for (int s = 0; s < a.batch; ++s) g += a.deltas[((size_t)l * a.batch + s) * kMdMaxWidth + (int)i];
      a.gb[l][i] = g;
      return;
    }
    i -= H;
  }
}

}  // namespace l2o
