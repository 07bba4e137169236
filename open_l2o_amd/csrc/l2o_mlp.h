// l2o_mlp.h -- the neural optimizee of the reference harness: problems.mnist
// (DM/problems.py:246-288), a [n_in -> n_hidden -> n_out] MLP with mean sparse-softmax
// cross-entropy on a gathered minibatch: forward + gradient in two small multi-workgroup
// launches (8 MFLOP per evaluation: latency-, not throughput-bound; the LSTM step on its
// 15 910 coordinates runs in k_cwlstm_step).  Included by l2o_kernels.hip.
//   k_mlp_fwd : 4 samples per workgroup, 64 threads per sample over the 784 inputs (w1 rows
//               straight from L2 for the common hidden width 20)
//               -> H, dZ = (softmax - onehot)/batch, dH, per-sample loss (scratch in HBM)
//   k_mlp_bwd : 64 input rows of gw1 per workgroup (coalesced image rows), one extra
//               workgroup for gw2, gb2, gb1 and the fixed-order loss sum
#pragma once

constexpr int kMlpMaxH = 32;
constexpr int kMlpMaxO = 16;
constexpr int kMlpTPS = 64;       // forward: threads per sample (k = lane + 64 i)
constexpr int kMlpSPB = 256 / kMlpTPS;   // samples per forward workgroup
constexpr int kMlpKPB = 64;       // gw1 rows per backward workgroup
constexpr int kMlpKPT = 16;       // inputs per forward thread: n_in <= kMlpTPS * kMlpKPT
constexpr int kMlpNPT = 32;       // samples per backward thread and pass

// cooperative global -> LDS copy by 256 threads with 8 independent loads in flight per thread
// (a plain one-load-per-iteration loop pays one memory latency per iteration)
template <class T>
__device__ __forceinline__ void coop_copy256(T* __restrict__ dst, const T* __restrict__ src, int n, int tid) {
  for (int base = tid; base < n; base += 256 * 8) {
    T v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = base + 256 * j;
      v[j] = i < n ? src[i] : T(0);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = base + 256 * j;
      if (i < n) dst[i] = v[j];
    }
  }
}

#ifdef L2O_MLP_CLOCK
#define MLP_CK(i) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 0) ck[i] = __builtin_readcyclecounter(); } while (0)
#else
#define MLP_CK(i) ((void)0)
#endif

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

// HP = compile-time padded hidden width (H == HP, or H < HP with zero-padded LDS columns):
// the hot loops carry no run-time guard -- with `if (u < H)` inside the unrolled loop hipcc
// emits one branch + ds_read + wait + FMA per element (90 cycles each, measured).
template <int HP>
__global__ __launch_bounds__(256) void k_mlp_fwd(MlpParams p) {
  extern __shared__ float sm[];
  const int n_in = p.n_in, H = p.H, O = p.O, Bn = p.batch;
  constexpr int HS = HP | 1;                            // odd row stride: conflict-free w1 reads
  float* w1s = sm;                                      // [n_in][HS]   (generic width only)
  float* part = w1s + (HP == 20 ? 0 : n_in * HS);       // [SPB][TPS][HP]
  float* hs = part + kMlpSPB * kMlpTPS * HP;            // [SPB][H]
  float* zs = hs + kMlpSPB * H;                         // [SPB][O]
  float* sw2 = zs + kMlpSPB * O;                        // [H][O]  (staged: the tail below is latency bound)
  float* sb1 = sw2 + H * O;                             // [H]
  float* sb2 = sb1 + H;                                 // [O]
#ifdef L2O_MLP_CLOCK
  long long ck[10];
#endif
  const int tid = threadIdx.x;
  const int sl = tid / kMlpTPS, l32 = tid % kMlpTPS;
  MLP_CK(0);
  int lab = 0;                                          // label of sample blockIdx * SPB + tid (tid < SPB)
  if (tid < kMlpSPB && (int)blockIdx.x * kMlpSPB + tid < Bn) lab = p.labels[p.idx[blockIdx.x * kMlpSPB + tid]];
  coop_copy256(sw2, p.w2, H * O, tid);
  coop_copy256(sb1, p.b1, H, tid);
  coop_copy256(sb2, p.b2, O, tid);
  const int n = blockIdx.x * kMlpSPB + sl;
  const bool valid = n < Bn;
  const int row = valid ? p.idx[n] : 0;
  // the sample's inputs of this thread (k = l32 + 64 i) go out first, all at once: one memory
  // latency instead of one per loop iteration, and it overlaps the staging of w1
  const float* xrow = p.images + (size_t)row * n_in;
  float xr[kMlpKPT];
#pragma unroll
  for (int i = 0; i < kMlpKPT; ++i) {
    const int k = l32 + kMlpTPS * i;
    xr[i] = k < n_in ? xrow[k] : 0.0f;
  }
  if constexpr (HP == 20) {
    // the common width: w1 rows (20 floats = 5 dwordx4) straight from L2, four rows per thread in
    // flight; no 62 KB staging pass per workgroup, no barrier before the GEMV
    __syncthreads();                                      // (sw2, sb1, sb2 staged)
    MLP_CK(1);
    float acc[HP];
#pragma unroll
    for (int u = 0; u < HP; ++u) acc[u] = 0.0f;
    const float4* w4 = reinterpret_cast<const float4*>(p.w1);
#pragma unroll
    for (int i0 = 0; i0 < kMlpKPT; i0 += 4) {
      if (kMlpTPS * i0 < n_in) {                          // wave-uniform
        float4 wv[4][5];
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const int k = min(l32 + kMlpTPS * (i0 + ii), n_in - 1);   // out-of-range slots carry xr == 0
#pragma unroll
          for (int c5 = 0; c5 < 5; ++c5) wv[ii][c5] = w4[k * 5 + c5];
        }
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const float xv = xr[i0 + ii];
#pragma unroll
          for (int c5 = 0; c5 < 5; ++c5) {
            acc[4 * c5 + 0] = __builtin_fmaf(xv, wv[ii][c5].x, acc[4 * c5 + 0]);
            acc[4 * c5 + 1] = __builtin_fmaf(xv, wv[ii][c5].y, acc[4 * c5 + 1]);
            acc[4 * c5 + 2] = __builtin_fmaf(xv, wv[ii][c5].z, acc[4 * c5 + 2]);
            acc[4 * c5 + 3] = __builtin_fmaf(xv, wv[ii][c5].w, acc[4 * c5 + 3]);
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < HP; ++u) part[(sl * kMlpTPS + l32) * HP + u] = acc[u];
  } else {
    for (int i = tid; i < n_in * HS; i += 256) w1s[i] = 0.0f;   // generic width: zero the padding columns
    __syncthreads();
    for (int base = tid; base < n_in * H; base += 256 * 8) {
      float v[8];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int i = base + 256 * jj;
        v[jj] = i < n_in * H ? p.w1[i] : 0.0f;
      }
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int i = base + 256 * jj;
        if (i < n_in * H) w1s[(i / H) * HS + (i % H)] = v[jj];
      }
    }
    __syncthreads();
    MLP_CK(1);
    float acc[HP];
#pragma unroll
    for (int u = 0; u < HP; ++u) acc[u] = 0.0f;
#pragma unroll
    for (int i = 0; i < kMlpKPT; ++i) {
      const int k = min(l32 + kMlpTPS * i, n_in - 1);   // out-of-range slots carry xr == 0
      const float xv = xr[i];
      const float* wr = w1s + k * HS;
#pragma unroll
      for (int u = 0; u < HP; ++u) acc[u] = __builtin_fmaf(xv, wr[u], acc[u]);
    }
#pragma unroll
    for (int u = 0; u < HP; ++u) part[(sl * kMlpTPS + l32) * HP + u] = acc[u];
  }
  MLP_CK(2);
  __syncthreads();
  float* gH = p.scratch;
  float* gdZ = gH + Bn * H;
  float* gdH = gdZ + Bn * O;
  float* gloss = gdH + Bn * H;
  if (tid < kMlpSPB * H) {                              // hidden activation: thread = (sample, unit)
    const int s2 = tid / H, u = tid % H;
    float a = sb1[u];
#pragma unroll 8
    for (int l = 0; l < kMlpTPS; ++l) a += part[(s2 * kMlpTPS + l) * HP + u];
    hs[s2 * H + u] = p.act == 0 ? 1.0f / (1.0f + expf(-a)) : fmaxf(a, 0.0f);
  }
  __syncthreads();
  MLP_CK(3);
  if (tid < kMlpSPB * O) {                              // logits: thread = (sample, class)
    const int s2 = tid / O, o = tid % O;
    float a = sb2[o];
    for (int u = 0; u < H; ++u) a = __builtin_fmaf(hs[s2 * H + u], sw2[u * O + o], a);
    zs[s2 * O + o] = a;
  }
  __syncthreads();
  MLP_CK(4);
  if (tid < kMlpSPB) {                                  // softmax cross-entropy: thread = sample
    const int n2 = blockIdx.x * kMlpSPB + tid;
    if (n2 < Bn) {
      const float* z = zs + tid * O;
      float zmax = z[0];
      for (int o = 1; o < O; ++o) zmax = fmaxf(zmax, z[o]);
      float se = 0.0f;
      for (int o = 0; o < O; ++o) se += expf(z[o] - zmax);
      const float lse = zmax + logf(se);
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
  MLP_CK(5);
  if (tid < kMlpSPB * H) {                              // dH and H to HBM: thread = (sample, unit)
    const int s2 = tid / H, u = tid % H;
    const int n2 = blockIdx.x * kMlpSPB + s2;
    if (n2 < Bn) {
      float a = 0.0f;
      for (int o = 0; o < O; ++o) a = __builtin_fmaf(zs[s2 * O + o], sw2[u * O + o], a);
      const float h = hs[s2 * H + u];
      gH[n2 * H + u] = h;
      gdH[n2 * H + u] = p.act == 0 ? a * h * (1.0f - h) : (h > 0.0f ? a : 0.0f);
    }
  }
  MLP_CK(6);
#ifdef L2O_MLP_CLOCK
  if (threadIdx.x == 0 && blockIdx.x == 0)
    printf("k_mlp_fwd ticks: x+stage %lld  gemv %lld  hidden %lld  logits %lld  softmax %lld  dH %lld\n", ck[1] - ck[0],
           ck[2] - ck[1], ck[3] - ck[2], ck[4] - ck[3], ck[5] - ck[4], ck[6] - ck[5]);
#endif
}

template <int HP>
__global__ __launch_bounds__(256) void k_mlp_bwd(MlpParams p) {
  extern __shared__ float sm[];
  const int n_in = p.n_in, H = p.H, O = p.O, Bn = p.batch;
  const float* gH = p.scratch;
  const float* gdZ = gH + Bn * H;
  const float* gdH = gdZ + Bn * O;
  const int tid = threadIdx.x;
  const int nkb = (n_in + kMlpKPB - 1) / kMlpKPB;
  if ((int)blockIdx.x == nkb) {                         // the small tensors + the loss
    // scratch (H | dZ | dH | loss_n, contiguous) -> LDS first; everything below reads LDS
    const int tot = Bn * (2 * H + O + 1);
    coop_copy256(sm, p.scratch, tot, tid);
    __syncthreads();
    const float* sH = sm;
    const float* sdZ = sH + Bn * H;
    const float* sdH = sdZ + Bn * O;
    const float* sloss = sdH + Bn * H;
    if (tid < 64) {                                     // fixed-order loss sum: lane strides, then the wave tree
      float a = 0.0f;
      for (int n = tid; n < Bn; n += 64) a += sloss[n];
      a = l2o::wave_sum64(a);
      if (tid == 0) p.loss[0] = a / (float)Bn;
    }
    if (p.gw1 == nullptr) return;
    for (int e = tid; e < H * O; e += 256) {
      const int u = e / O, o = e % O;
      float a = 0.0f;
      for (int n = 0; n < Bn; ++n) a = __builtin_fmaf(sH[n * H + u], sdZ[n * O + o], a);
      p.gw2[e] = a;
    }
    for (int o = tid; o < O; o += 256) {
      float a = 0.0f;
      for (int n = 0; n < Bn; ++n) a += sdZ[n * O + o];
      p.gb2[o] = a;
    }
    for (int u = tid; u < H; u += 256) {
      float a = 0.0f;
      for (int n = 0; n < Bn; ++n) a += sdH[n * H + u];
      p.gb1[u] = a;
    }
    return;
  }
  if (p.gw1 == nullptr) return;
  float* dhs = sm;                                      // [Bn][HP]  (columns >= H zero)
  int* rows = reinterpret_cast<int*>(dhs + Bn * HP);    // [Bn]
  float* part = reinterpret_cast<float*>(rows + Bn);    // [4][KPB][HP]
  if (HP != 20) {
    for (int i = tid; i < Bn * HP; i += 256) dhs[i] = 0.0f;
    __syncthreads();
  }
  {
    const int Hc = HP == 20 ? 20 : H;
    for (int base = tid; base < Bn * Hc; base += 256 * 8) {
      float v[8];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int i = base + 256 * jj;
        v[jj] = i < Bn * Hc ? gdH[i] : 0.0f;
      }
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int i = base + 256 * jj;
        if (i < Bn * Hc) dhs[(i / Hc) * HP + (i % Hc)] = v[jj];
      }
    }
  }
  coop_copy256(rows, p.idx, Bn, tid);
  __syncthreads();
  const int kl = tid & (kMlpKPB - 1), np = tid >> 6;    // 64 rows x 4 sample-quarters
  const int k = blockIdx.x * kMlpKPB + kl;
  float acc[HP];
#pragma unroll
  for (int u = 0; u < HP; ++u) acc[u] = 0.0f;
  {
    const int kc = min(k, n_in - 1);
    // samples n = np + 4 i: the gathered image values of a pass are all requested before the
    // first FMA (one memory latency per pass of 32, not one per sample)
    for (int n0 = np; n0 < Bn; n0 += 4 * kMlpNPT) {
      float xr[kMlpNPT];
#pragma unroll
      for (int i = 0; i < kMlpNPT; ++i) {
        const int n = n0 + 4 * i;
        xr[i] = n < Bn ? p.images[(size_t)rows[n] * n_in + kc] : 0.0f;
      }
#pragma unroll
      for (int i = 0; i < kMlpNPT; ++i) {
        const int n = min(n0 + 4 * i, Bn - 1);            // out-of-range slots carry xr == 0
        const float xv = xr[i];
        const float* dh = dhs + n * HP;
#pragma unroll
        for (int u = 0; u < HP; ++u) acc[u] = __builtin_fmaf(xv, dh[u], acc[u]);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < HP; ++u) part[(np * kMlpKPB + kl) * HP + u] = acc[u];
  __syncthreads();
  for (int e = tid; e < kMlpKPB * H; e += 256) {
    const int kk = e / H, u = e % H;
    const int kg = blockIdx.x * kMlpKPB + kk;
    if (kg < n_in) {
      const float s = (part[(0 * kMlpKPB + kk) * HP + u] + part[(1 * kMlpKPB + kk) * HP + u]) +
                      (part[(2 * kMlpKPB + kk) * HP + u] + part[(3 * kMlpKPB + kk) * HP + u]);
      p.gw1[kg * H + u] = s;
    }
  }
}


// ---------------------------------------------------------------------------------------------
// The reference's own MLP (problems.mnist: hidden width 20, DM/problems.py:254-288) without LDS
// staging and without workgroup barriers -- the evaluation is 8 MFLOP, i.e. pure latency, and
// the 4-samples-per-workgroup form above spends 16 + 10 us on 32 + 14 workgroups (five barrier
// rounds of a few hundred cycles each after the GEMV):
//   k_mlp_fwd20 : ONE WAVE PER SAMPLE (batch workgroups of 64 threads).  Lane l owns the inputs
//                 k = l + 64 i; the 20 hidden pre-activations are 20 deterministic wave sums
//                 (DPP + permlane swaps); layer 2, the softmax and dH run on lanes 0..19 with
//                 v_readlane broadcasts.  No LDS at all.
//   k_mlp_bwd20 : 64 rows of gw1 per workgroup of 8 waves; wave w owns the samples n = w + 8 i,
//                 whose index and dH row are wave-uniform -> scalar loads, SGPR operands of the
//                 FMAs; the image values are coalesced 256-byte row pieces.  One LDS round to
//                 add the 8 partial sums in a fixed order.  The last workgroup computes gw2, gb2,
//                 gb1 and the fixed-order loss sum as above.
// ---------------------------------------------------------------------------------------------
constexpr int kMlp20 = 20;
constexpr int kMlpBwdWaves = 16;
__device__ __forceinline__ float mlp_rl(float v, int l) {         // value of lane l (wave-uniform l), in every lane
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

__global__ __launch_bounds__(64) void k_mlp_fwd20(MlpParams p) {
  constexpr int H = kMlp20;
  const int n_in = p.n_in, O = p.O, Bn = p.batch;
  const int lane = threadIdx.x;
  const int n = blockIdx.x;
  // Program order = issue order: everything that does NOT depend on the sample index goes out first
  // (the rows of w1 this lane needs, 13 x 80 bytes for 784 inputs = 260 registers of a 512-register
  // wave, and the small operands of the tail), then the index -> image row chain rides on top.
  const float4* w4 = reinterpret_cast<const float4*>(p.w1);
  constexpr int KI = 13;                                 // 64 * 13 = 832 >= 784 (the reference's input width)
  const bool all = n_in <= 64 * KI;                      // (else: batches of four rows below)
  float4 wv[KI][5];
  if (all) {
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      const int k = min(lane + 64 * i, n_in - 1);        // out-of-range slots carry xr == 0
#pragma unroll
      for (int c5 = 0; c5 < 5; ++c5) wv[i][c5] = w4[k * 5 + c5];
    }
  }
  const float b1v = lane < H ? p.b1[lane] : 0.0f;
  const float b2v = lane < O ? p.b2[lane] : 0.0f;
  float w2c[H];                                          // lane o < O: column o of w2
#pragma unroll
  for (int u = 0; u < H; ++u) w2c[u] = lane < O ? p.w2[u * O + lane] : 0.0f;
  float w2r[kMlpMaxO];                                   // lane u < H: row u of w2
#pragma unroll
  for (int o = 0; o < kMlpMaxO; ++o) w2r[o] = (lane < H && o < O) ? p.w2[lane * O + o] : 0.0f;
  const int row = p.idx[n];
  const int lab = p.labels[row];
  const float* xrow = p.images + (size_t)row * n_in;
  float xr[kMlpKPT];
#pragma unroll
  for (int i = 0; i < kMlpKPT; ++i) {
    const int k = lane + 64 * i;
    xr[i] = k < n_in ? xrow[k] : 0.0f;
  }
  float acc[H];
#pragma unroll
  for (int u = 0; u < H; ++u) acc[u] = 0.0f;
  if (all) {
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      const float xv = xr[i];
#pragma unroll
      for (int c5 = 0; c5 < 5; ++c5) {
        acc[4 * c5 + 0] = __builtin_fmaf(xv, wv[i][c5].x, acc[4 * c5 + 0]);
        acc[4 * c5 + 1] = __builtin_fmaf(xv, wv[i][c5].y, acc[4 * c5 + 1]);
        acc[4 * c5 + 2] = __builtin_fmaf(xv, wv[i][c5].z, acc[4 * c5 + 2]);
        acc[4 * c5 + 3] = __builtin_fmaf(xv, wv[i][c5].w, acc[4 * c5 + 3]);
      }
    }
  } else {
#pragma unroll
    for (int i0 = 0; i0 < kMlpKPT; i0 += 4) {
      if (64 * i0 < n_in) {                              // wave-uniform
        float4 wb[4][5];
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const int k = min(lane + 64 * (i0 + ii), n_in - 1);
#pragma unroll
          for (int c5 = 0; c5 < 5; ++c5) wb[ii][c5] = w4[k * 5 + c5];
        }
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const float xv = xr[i0 + ii];
#pragma unroll
          for (int c5 = 0; c5 < 5; ++c5) {
            acc[4 * c5 + 0] = __builtin_fmaf(xv, wb[ii][c5].x, acc[4 * c5 + 0]);
            acc[4 * c5 + 1] = __builtin_fmaf(xv, wb[ii][c5].y, acc[4 * c5 + 1]);
            acc[4 * c5 + 2] = __builtin_fmaf(xv, wb[ii][c5].z, acc[4 * c5 + 2]);
            acc[4 * c5 + 3] = __builtin_fmaf(xv, wb[ii][c5].w, acc[4 * c5 + 3]);
          }
        }
      }
    }
  }
  float a = 0.0f;                                        // lane u < 20 keeps pre-activation u
#pragma unroll
  for (int u = 0; u < H; ++u) {
    const float t = l2o::wave_sum64(acc[u]);
    a = lane == u ? t : a;
  }
  a += b1v;
  float h = p.act == 0 ? 1.0f / (1.0f + expf(-a)) : fmaxf(a, 0.0f);
  if (lane >= H) h = 0.0f;
  float z = b2v;                                         // lane o < O: logit o
#pragma unroll
  for (int u = 0; u < H; ++u) z = __builtin_fmaf(mlp_rl(h, u), w2c[u], z);
  float zmax = mlp_rl(z, 0);
  for (int o = 1; o < O; ++o) zmax = fmaxf(zmax, mlp_rl(z, o));
  const float e = lane < O ? expf(z - zmax) : 0.0f;
  float se = 0.0f;
  for (int o = 0; o < O; ++o) se += mlp_rl(e, o);
  const float lse = zmax + logf(se);
  const float zl = mlp_rl(z, lab);
  float* gH = p.scratch;
  float* gdZ = gH + Bn * H;
  float* gdH = gdZ + Bn * O;
  float* gloss = gdH + Bn * H;
  if (lane == 0) gloss[n] = lse - zl;
  const float d = lane < O ? (expf(z - lse) - (lane == lab ? 1.0f : 0.0f)) * (1.0f / (float)Bn) : 0.0f;
  if (lane < O) gdZ[n * O + lane] = d;
  float dh = 0.0f;                                       // lane u < 20: dL/dh_u
#pragma unroll
  for (int o = 0; o < kMlpMaxO; ++o)
    if (o < O) dh = __builtin_fmaf(mlp_rl(d, o), w2r[o], dh);
  if (lane < H) {
    gH[n * H + lane] = h;
    gdH[n * H + lane] = p.act == 0 ? dh * h * (1.0f - h) : (h > 0.0f ? dh : 0.0f);
  }
}

__global__ __launch_bounds__(64 * kMlpBwdWaves) void k_mlp_bwd20(MlpParams p) {
  constexpr int H = kMlp20, NW = kMlpBwdWaves, HS = H + 1;     // odd LDS row stride: conflict-free
  constexpr int NT = 64 * NW;
  extern __shared__ float sm[];
  const int n_in = p.n_in, O = p.O, Bn = p.batch;
  const float* gH = p.scratch;
  const float* gdZ = gH + Bn * H;
  const float* gdH = gdZ + Bn * O;
  const int tid = threadIdx.x;
  const int nkb = (n_in + 63) / 64;
  if ((int)blockIdx.x == nkb) {                         // the small tensors + the loss
    const int tot = Bn * (2 * H + O + 1);
    for (int i = tid; i < tot; i += NT) sm[i] = p.scratch[i];
    __syncthreads();
    const float* sH = sm;
    const float* sdZ = sH + Bn * H;
    const float* sdH = sdZ + Bn * O;
    const float* sloss = sdH + Bn * H;
    if (tid < 64) {                                     // fixed-order loss sum: lane strides, then the wave tree
      float a = 0.0f;
      for (int n = tid; n < Bn; n += 64) a += sloss[n];
      a = l2o::wave_sum64(a);
      if (tid == 0) p.loss[0] = a / (float)Bn;
    }
    if (p.gw1 == nullptr) return;
    // waves 1.. : gw2 (H*O), gb2 (O), gb1 (H): FOUR threads (a DPP quad) per output, thread part sums the
    // samples n = part + 4 i, then the quad adds its four partial sums in a fixed order
    const int t4 = tid - 64;
    if (t4 < 0) return;
    const int part = t4 & 3;
    for (int e = t4 >> 2; e < H * O + O + H; e += (NT - 64) / 4) {     // (quad-uniform trip count)
      float a = 0.0f;
      if (e < H * O) {
        const int u = e / O, o = e % O;
        for (int n = part; n < Bn; n += 4) a = __builtin_fmaf(sH[n * H + u], sdZ[n * O + o], a);
      } else if (e < H * O + O) {
        const int o = e - H * O;
        for (int n = part; n < Bn; n += 4) a += sdZ[n * O + o];
      } else {
        const int u = e - H * O - O;
        for (int n = part; n < Bn; n += 4) a += sdH[n * H + u];
      }
      a = l2o::quad_sum(a);
      if (part == 0) {
        if (e < H * O) p.gw2[e] = a;
        else if (e < H * O + O) p.gb2[e - H * O] = a;
        else p.gb1[e - H * O - O] = a;
      }
    }
    return;
  }
  if (p.gw1 == nullptr) return;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = blockIdx.x * 64 + lane;
  const int kc = min(k, n_in - 1);
  float acc[H];
#pragma unroll
  for (int u = 0; u < H; ++u) acc[u] = 0.0f;
  const l2o_cfp dH = (l2o_cfp)gdH;
  const __attribute__((address_space(4))) int* idx = (const __attribute__((address_space(4))) int*)p.idx;
  for (int n0 = wv; n0 < Bn; n0 += NW * 8) {            // 8 samples of this wave per pass, their loads in flight together
    float xr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int n = n0 + NW * i;
      xr[i] = n < Bn ? p.images[(size_t)idx[n < Bn ? n : 0] * n_in + kc] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int n = min(n0 + NW * i, Bn - 1);             // out-of-range slots carry xr == 0
      const l2o_cfp dh = dH + n * H;
#pragma unroll
      for (int u = 0; u < H; ++u) acc[u] = __builtin_fmaf(xr[i], dh[u], acc[u]);
    }
  }
  float* part = sm;                                     // [NW][64][HS]
#pragma unroll
  for (int u = 0; u < H; ++u) part[(wv * 64 + lane) * HS + u] = acc[u];
  __syncthreads();
  for (int e = tid; e < 64 * H; e += NT) {
    const int kk = e / H, u = e % H;
    const int kg = blockIdx.x * 64 + kk;
    if (kg < n_in) {
      float s = 0.0f;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += part[(w * 64 + kk) * HS + u];
      p.gw1[kg * H + u] = s;
    }
  }
}
