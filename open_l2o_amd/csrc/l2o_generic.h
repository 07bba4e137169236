// l2o_generic.h -- the coordinate-wise LSTM optimizer step for ANY `layers` tuple (1..3 layers, hidden <= 64):
// StandardDeepLSTM._build (DM/networks.py:207-232) behind CoordinateWiseDeepLSTM / RNNprop._build
// (:254-271, :287-295) for the configurations the matrix-core kernels do not cover (they implement the
// harness' (20, 20)); e.g. the reference's own networks_test.py builds layers=(1,) and (1, 1).
// Plain fp32 VALU, one thread per coordinate, Sonnet-layout weights read through the scalar unit
// (wave-uniform addresses), the [input | h_prev] row of a coordinate staged in LDS.  Not a fast path:
// a correct device path for the plugin contract `net(inputs, prev_state) -> (delta, next_state)`.
// Also serves RNNprop.__call__(m, g, state): `direct_inputs` feeds (m~, g~) as given (no moment update).
// State layout (l2o_gen_state_floats): per layer l, hidden [N][H_l] then cell [N][H_l] -- the reference's
// ((hidden_1, cell_1), (hidden_2, cell_2), ...) tuples stored back to back.
// Included by l2o_kernels.hip.
#pragma once

constexpr int kGenMaxH = 64;
constexpr int kGenMaxL = 3;
constexpr int kGenThreads = 64;
constexpr int kGenRow = 2 * kGenMaxH + 1;    // [input (<= 64) | h_prev (<= 64)], odd stride: conflict-free column reads

struct GenParams {
  int n_layers, H[kGenMaxL], in_dim, pre, direct, tanh_output;
  float scale, k_inv, exp_k, beta1, beta2, omb1, omb2, om1, om2;
  const float* wg[kGenMaxL];
  const float* bg[kGenMaxL];
  const float *wl, *bl, *wfc, *bfc;
  const float* g;        // [N] gradient (direct: g~)
  const float* m_in;     // [N] direct: m~
  float* m;              // [N] RNNProp moments in-out (NULL when direct)
  float* v;
  float* state;
  float* x;              // [N] in-out
  long N;
};

__global__ __launch_bounds__(kGenThreads) void k_cwlstm_generic(GenParams p) {
  __shared__ float row[kGenThreads][kGenRow];
  __shared__ float nxt[kGenThreads][kGenMaxH + 1];
  const int t = threadIdx.x;
  const long i = (long)blockIdx.x * kGenThreads + t;
  const bool live = i < p.N;
  const long ii = live ? i : p.N - 1;                       // dead lanes shadow the last coordinate (no stores)
  // ---- preprocessing -> the layer-0 input features (DM/networks.py:218-221)
  float gv = p.g[ii];
  int width = 1;
  if (p.pre == L2O_PRE_FC_ELU) {
    float f0, f1;
    if (p.direct) {
      f0 = p.m_in[ii]; f1 = gv;
    } else {
      float mm = p.m[ii], vv = p.v[ii];
      mm = p.beta1 * mm + p.omb1 * gv;
      vv = p.beta2 * vv + p.omb2 * gv * gv;
      if (live) { p.m[i] = mm; p.v[i] = vv; }
      const float mh = mm / p.om1, vh = vv / p.om2;         // DM/meta_rnnprop_train.py:383-388
      const float den = sqrtf(vh) + 1e-8f;
      f0 = mh / den; f1 = gv / den;
    }
    width = p.in_dim;
    const l2o_cfp wfc = (l2o_cfp)p.wfc;
    const l2o_cfp bfc = (l2o_cfp)p.bfc;
    for (int u = 0; u < width; ++u) {
      const float a = __builtin_fmaf(f1, wfc[width + u], __builtin_fmaf(f0, wfc[u], bfc[u]));
      row[t][u] = a > 0.0f ? a : expm1f(a);                 // tf.nn.elu
    }
  } else if (p.pre == L2O_PRE_LOGSIGN) {                    // DM/preprocess.py:63-70
    row[t][0] = fmaxf(logf(fabsf(gv) + 1.1920928955078125e-07f) * p.k_inv, -1.0f);
    row[t][1] = fminf(fmaxf(gv * p.exp_k, -1.0f), 1.0f);
    width = 2;
  } else {
    row[t][0] = gv;
  }
  // ---- the LSTM stack (snt.DeepRNN of snt.LSTM: gates i, j, f, o; forget_bias 1; DM/networks.py:192-200, 225)
  float* st = p.state;
  for (int l = 0; l < p.n_layers; ++l) {
    const int H = p.H[l], K = width + H, G = 4 * H;
    float* hbuf = st;
    float* cbuf = st + p.N * (long)H;
    for (int u = 0; u < H; ++u) row[t][width + u] = hbuf[ii * H + u];
    const l2o_cfp wg = (l2o_cfp)p.wg[l];
    const l2o_cfp bg = (l2o_cfp)p.bg[l];
    for (int u = 0; u < H; ++u) {
      float zi = bg[u], zj = bg[H + u], zf = bg[2 * H + u], zo = bg[3 * H + u];
      for (int k = 0; k < K; ++k) {
        const float av = row[t][k];
        zi = __builtin_fmaf(av, wg[k * G + u], zi);
        zj = __builtin_fmaf(av, wg[k * G + H + u], zj);
        zf = __builtin_fmaf(av, wg[k * G + 2 * H + u], zf);
        zo = __builtin_fmaf(av, wg[k * G + 3 * H + u], zo);
      }
      const float cp = cbuf[ii * H + u];
      const float cn = cp / (1.0f + expf(-(zf + 1.0f))) + tanhf(zj) / (1.0f + expf(-zi));
      const float hn = tanhf(cn) / (1.0f + expf(-zo));
      nxt[t][u] = hn;
      if (live) { cbuf[i * H + u] = cn; hbuf[i * H + u] = hn; }
    }
    for (int u = 0; u < H; ++u) row[t][u] = nxt[t][u];      // next layer's input (same thread: no barrier)
    width = H;
    st += 2 * p.N * (long)H;
  }
  // ---- Linear(-> 1), (tanh) * scale, x += delta (DM/networks.py:227-232; DM/meta.py:353)
  const l2o_cfp wl = (l2o_cfp)p.wl;
  float d = ((l2o_cfp)p.bl)[0];
  for (int u = 0; u < width; ++u) d = __builtin_fmaf(row[t][u], wl[u], d);
  if (p.tanh_output) d = tanhf(d);
  if (live) p.x[i] += d * p.scale;
}
