// l2o_vecops.h -- the small elementwise / reduction passes of the meta-gradient's less common branches (ABI v11).
// Until round 4 these ran as torch tensor arithmetic in open_l2o_amd/meta.py (.sum, torch.sqrt, torch.where, chains of
// adds): generic `layers` BPTT, the Linear-only net, second derivatives.  Not hot -- a T-step training unroll calls each
// of them once or T times on a few thousand to a few hundred thousand floats -- but the training path now runs on l2o_*
// entry points only.  Included by l2o_kernels.hip.
#pragma once

// out[t][i] = g_final[i] + sum_{tau > t} g[tau][i]  for t = T-1 .. 0: dL/d(delta_t) of loss = sum_t f(x_t)
// (DM/meta.py:372-376 with the optimizee gradients held constant, DM/meta.py:328-329).  g: a device table of T
// pointers (the recorded gradients live in per-step buffers).  One thread per coordinate, T dependent adds in the
// order of the former host loop (acc = g_final; for t descending: out[t] = acc; acc += g[t]) -- bit-identical to it.
__global__ __launch_bounds__(256) void k_suffix_sums(const float* const* __restrict__ g, const float* __restrict__ g_final,
                                                     float* __restrict__ out, long n, int T) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float acc = g_final[i];
  for (int t = T - 1; t >= 0; --t) {
    out[(size_t)t * n + i] = acc;
    acc += g[t][i];
  }
}

// Column sums of `batch` row-major [rows, cols] matrices: out[b][k] (+)= sum_r A[b][r][k]  (bias gradients = dz^T 1;
// the column sums of square_cos' wcos).  Two passes, both in a FIXED order (bit-reproducible): kColsumSplit row slabs per
// matrix, each summed top to bottom by one thread per column (consecutive threads = consecutive columns: coalesced), then
// the slabs added in ascending order.
constexpr int kColsumSplit = 64;
__global__ __launch_bounds__(256) void k_colsum_part(const float* __restrict__ A, long rows, int cols, float* __restrict__ part) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int slab = blockIdx.y;
  const long b = blockIdx.z;
  if (k >= cols) return;
  const long per = (rows + kColsumSplit - 1) / kColsumSplit;
  const long r0 = slab * per, r1 = r0 + per < rows ? r0 + per : rows;
  const float* p = A + ((size_t)b * rows) * cols + k;
  float acc = 0.0f;
  for (long r = r0; r < r1; ++r) acc += p[(size_t)r * cols];
  part[((size_t)b * kColsumSplit + slab) * cols + k] = acc;
}
__global__ __launch_bounds__(256) void k_colsum_final(const float* __restrict__ part, int cols, float* __restrict__ out,
                                                      int accumulate) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  const long b = blockIdx.y;
  if (k >= cols) return;
  float acc = 0.0f;
  for (int s = 0; s < kColsumSplit; ++s) acc += part[((size_t)b * kColsumSplit + s) * cols + k];
  float* o = out + (size_t)b * cols + k;
  *o = accumulate ? *o + acc : acc;
}

// out = ca a + cb b + cc c  (b, c may be NULL; out may alias any input): the running adjoints of the second-derivative
// path (lam <- g_t + lam + w H u) and the accumulation of per-step weight-gradient blocks.
__global__ __launch_bounds__(256) void k_lincomb(float* out, const float* a, float ca, const float* b, float cb,
                                                 const float* c, float cc, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = ca * a[i];
  if (b) v = __builtin_fmaf(cb, b[i], v);
  if (c) v = __builtin_fmaf(cc, c[i], v);
  out[i] = v;
}

// second_derivatives for RNNProp (DM/meta_rnnprop_train.py:380-388 WITHOUT the stop_gradient): the network inputs
// m~ = m^/(sqrt(v^) + 1e-8), g~ = g/(sqrt(v^) + 1e-8), m^ = m/(1 - b1^k), v^ = v/(1 - b2^k) depend on g_t directly and
// through the moment recurrences m_t = b1 m_{t-1} + (1 - b1) g_t, v_t = b2 v_{t-1} + (1 - b2) g_t^2 that later steps
// read.  From the step kernel's du (the adjoint of the input projection's pre-activations, H columns starting at
// column `du_col` of the Bm rows, leading dimension ldb) this forms u_t = dL/dg_t and the adjoints carried to step t - 1:
//   a0 = du . w_fc[0] (dL/dm~), a1 = du . w_fc[1] (dL/dg~);  den = sqrt(v^) + 1e-8;
//   d_den = -(a0 m^ + a1 g) / den^2;  d_v^ = d_den / (2 sqrt(v^)) (0 where v^ = 0)
//   dm = a0 / den / (1 - b1^k) + dm_in;  dv = d_v^ / (1 - b2^k) + dv_in
//   dg = a1 / den + (1 - b1) dm + 2 (1 - b2) g dv;   dm_out = b1 dm, dv_out = b2 dv
struct RnnpropAdjArgs {
  const float* Bm; long ldb; int du_col; int H;
  const float* w_fc;            // [2][H]
  const float *g, *m, *v;
  float om1, om2, omb1, omb2, b1, b2;
  float *dm, *dv, *dg;          // dm / dv in-out, dg out
  long n;
};
__global__ __launch_bounds__(256) void k_rnnprop_input_adjoint(RnnpropAdjArgs a) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const float* du = a.Bm + (size_t)i * a.ldb + a.du_col;
  float a0 = 0.0f, a1 = 0.0f;
  for (int h = 0; h < a.H; ++h) {
    a0 = __builtin_fmaf(du[h], a.w_fc[h], a0);
    a1 = __builtin_fmaf(du[h], a.w_fc[a.H + h], a1);
  }
  const float g = a.g[i], m_hat = a.m[i] / a.om1, sq = __builtin_sqrtf(a.v[i] / a.om2);
  const float den = sq + 1e-8f;
  const float d_den = -(a0 * m_hat + a1 * g) / (den * den);
  const float d_vhat = sq > 0.0f ? d_den * 0.5f / sq : 0.0f;
  const float dm = a0 / den / a.om1 + a.dm[i];
  const float dv = d_vhat / a.om2 + a.dv[i];
  a.dg[i] = a1 / den + dm * a.omb1 + dv * (2.0f * a.omb2) * g;
  a.dm[i] = dm * a.b1;
  a.dv[i] = dv * a.b2;
}
