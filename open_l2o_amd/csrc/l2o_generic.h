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

// ---------------------------------------------------------------------------
// One step of back-propagation through time for the same ANY-`layers` stack (round 3): what
// tf.train.AdamOptimizer(lr).minimize(loss) differentiates through `update` (DM/meta.py:319-336, 398-414; the
// optimizee gradient is a constant unless second_derivatives, :328-329) when networks.factory built a stack the
// matrix-core kernels do not cover (DM/networks.py:157 accepts any tuple; networks_test.py trains layers=(1,)).
// One thread per coordinate; the step is recomputed from the state saved before it.  The per-layer rows it emits
//     act_l [N][K_l] = [input_l | h_l(t-1)],   dz_l [N][4 H_l] = dL/d(gate pre-activations, order i, j, f, o)
// are what the host contracts into the weight gradients (act_l^T dz_l; column sums for the biases; h_last^T dd for
// the output Linear; feats^T du for RNNProp's input projection) -- and, until the backward sweep overwrites them, the
// kernel's own storage for the gate activations (dz_l) and tanh(c') (tc), so that no per-thread array depends on H.
// Carries: per layer dL/dh_l(t-1) [N][H_l] then dL/dc_l(t-1) [N][H_l] (the layout of the state).  Not a fast path.
struct GenBwdParams {
  int n_layers, H[kGenMaxL], in_dim, pre, tanh_output;
  float scale, k_inv, exp_k, om1, om2;
  const float* wg[kGenMaxL];
  const float* bg[kGenMaxL];
  const float *wl, *bl, *wfc, *bfc;
  const float* g;          // [N] the step's gradient
  const float* m;          // [N] RNNProp moments AFTER the step's update
  const float* v;
  const float* st_prev;    // state before the step (l2o_gen_state_floats layout)
  const float* dx_next;    // [N] dL/d(delta_t)
  const float* carry_in;   // state layout: dL/dh_l(t), dL/dc_l(t) from step t + 1 (zeros at the last step)
  float* carry_out;        // ... for step t - 1
  float* act[kGenMaxL];    // [N][K_l]
  float* dz[kGenMaxL];     // [N][4 H_l]
  float* tc;               // scratch [sum_l N H_l]: tanh(c_l(t))
  float* h_last;           // [N][H_last]
  float* dd;               // [N] dL/d(Linear output)
  float* feats;            // [N][2]  RNNProp (m~, g~)
  float* du;               // [N][in_dim] RNNProp: dL/d(input projection pre-activation)
  float* dg;               // optional [N]: dL/dg_t (identity / LogAndSign)
  long N;
};

__global__ __launch_bounds__(kGenThreads) void k_cwlstm_generic_bwd(GenBwdParams p) {
  const long n = (long)blockIdx.x * kGenThreads + threadIdx.x;
  if (n >= p.N) return;                                     // (no cross-thread traffic: dead threads just leave)
  const int L = p.n_layers;
  // ---- forward, layer by layer; inputs of layer l land in act_l[n][0 .. in_l)
  const float gv = p.g[n];
  int width = 1;
  float f0 = 0.0f, f1 = 0.0f;
  {
    float* a0 = p.act[0] + n * (long)((p.pre == L2O_PRE_FC_ELU ? p.in_dim : (p.pre == L2O_PRE_LOGSIGN ? 2 : 1)) + p.H[0]);
    if (p.pre == L2O_PRE_FC_ELU) {
      const float mh = p.m[n] / p.om1, vh = p.v[n] / p.om2;
      const float den = sqrtf(vh) + 1e-8f;
      f0 = mh / den; f1 = gv / den;
      width = p.in_dim;
      const l2o_cfp wfc = (l2o_cfp)p.wfc;
      const l2o_cfp bfc = (l2o_cfp)p.bfc;
      for (int u = 0; u < width; ++u) {
        const float a = __builtin_fmaf(f1, wfc[width + u], __builtin_fmaf(f0, wfc[u], bfc[u]));
        a0[u] = a > 0.0f ? a : expm1f(a);
      }
      p.feats[n * 2] = f0; p.feats[n * 2 + 1] = f1;
    } else if (p.pre == L2O_PRE_LOGSIGN) {
      a0[0] = fmaxf(logf(fabsf(gv) + 1.1920928955078125e-07f) * p.k_inv, -1.0f);
      a0[1] = fminf(fmaxf(gv * p.exp_k, -1.0f), 1.0f);
      width = 2;
    } else {
      a0[0] = gv;
    }
  }
  const int in0 = width;
  {
    const float* st = p.st_prev;
    float* tcl = p.tc;
    for (int l = 0; l < L; ++l) {
      const int H = p.H[l], K = width + H, G = 4 * H;
      float* row = p.act[l] + n * (long)K;
      const float* hprev = st + n * (long)H;
      const float* cprev = st + p.N * (long)H + n * (long)H;
      for (int u = 0; u < H; ++u) row[width + u] = hprev[u];
      const l2o_cfp wg = (l2o_cfp)p.wg[l];
      const l2o_cfp bg = (l2o_cfp)p.bg[l];
      float* gates = p.dz[l] + n * (long)G;                 // gate ACTIVATIONS until the backward sweep
      float* hout = l + 1 < L ? p.act[l + 1] + n * (long)(H + p.H[l + 1]) : p.h_last + n * (long)H;
      for (int u = 0; u < H; ++u) {
        float zi = bg[u], zj = bg[H + u], zf = bg[2 * H + u], zo = bg[3 * H + u];
        for (int k = 0; k < K; ++k) {
          const float av = row[k];
          zi = __builtin_fmaf(av, wg[k * G + u], zi);
          zj = __builtin_fmaf(av, wg[k * G + H + u], zj);
          zf = __builtin_fmaf(av, wg[k * G + 2 * H + u], zf);
          zo = __builtin_fmaf(av, wg[k * G + 3 * H + u], zo);
        }
        const float gi = 1.0f / (1.0f + expf(-zi)), gj = tanhf(zj), gf = 1.0f / (1.0f + expf(-(zf + 1.0f))),
                    go = 1.0f / (1.0f + expf(-zo));
        const float t = tanhf(gf * cprev[u] + gi * gj);
        gates[u] = gi; gates[H + u] = gj; gates[2 * H + u] = gf; gates[3 * H + u] = go;
        tcl[n * (long)H + u] = t;
        hout[u] = t * go;
      }
      width = H;
      st += 2 * p.N * (long)H;
      tcl += p.N * (long)H;
    }
  }
  // ---- output Linear and its adjoint
  const int HL = p.H[L - 1];
  const l2o_cfp wl = (l2o_cfp)p.wl;
  float dlin = ((l2o_cfp)p.bl)[0];
  for (int u = 0; u < HL; ++u) dlin = __builtin_fmaf(p.h_last[n * (long)HL + u], wl[u], dlin);
  float ddv = p.dx_next[n] * p.scale;
  if (p.tanh_output) { const float th = tanhf(dlin); ddv *= 1.0f - th * th; }
  p.dd[n] = ddv;
  // ---- backward, top layer first.  dh[] = dL/dh_l(t) arriving from above (the Linear, or layer l + 1's input adjoint)
  float dh[kGenMaxH];
  for (int u = 0; u < HL; ++u) dh[u] = ddv * wl[u];
  long st_off = 0, tc_off = 0;
  for (int l = 0; l < L; ++l) { st_off += 2 * p.N * (long)p.H[l]; tc_off += p.N * (long)p.H[l]; }
  for (int l = L - 1; l >= 0; --l) {
    const int H = p.H[l], G = 4 * H;
    const int inw = l == 0 ? in0 : p.H[l - 1];
    const int K = inw + H;
    st_off -= 2 * p.N * (long)H;
    tc_off -= p.N * (long)H;
    const float* cprev = p.st_prev + st_off + p.N * (long)H + n * (long)H;
    const float* cin_h = p.carry_in + st_off + n * (long)H;
    const float* cin_c = p.carry_in + st_off + p.N * (long)H + n * (long)H;
    float* cout_h = p.carry_out + st_off + n * (long)H;
    float* cout_c = p.carry_out + st_off + p.N * (long)H + n * (long)H;
    float* dzr = p.dz[l] + n * (long)G;
    const float* tcl = p.tc + tc_off + n * (long)H;
    for (int u = 0; u < H; ++u) {
      const float gi = dzr[u], gj = dzr[H + u], gf = dzr[2 * H + u], go = dzr[3 * H + u], t = tcl[u];
      const float dhu = dh[u] + cin_h[u];
      const float dc = cin_c[u] + dhu * go * (1.0f - t * t);
      cout_c[u] = dc * gf;
      dzr[u] = dc * gj * gi * (1.0f - gi);
      dzr[H + u] = dc * gi * (1.0f - gj * gj);
      dzr[2 * H + u] = dc * cprev[u] * gf * (1.0f - gf);
      dzr[3 * H + u] = dhu * t * go * (1.0f - go);
    }
    // d[input_l | h_l(t-1)] = dz_l . W_l^T  (row k of W_l is contiguous)
    const l2o_cfp wg = (l2o_cfp)p.wg[l];
    for (int k = 0; k < K; ++k) {
      float s = 0.0f;
      for (int q = 0; q < G; ++q) s = __builtin_fmaf(dzr[q], wg[k * G + q], s);
      if (k >= inw) cout_h[k - inw] = s;                   // dL/dh_l(t-1)
      else dh[k] = s;                                       // input adjoint: layer l - 1's dh, or the features' (l == 0)
    }
  }
  // ---- through the preprocessing (dh[0 .. in0) = dL/d features)
  if (p.pre == L2O_PRE_FC_ELU) {
    const l2o_cfp wfc = (l2o_cfp)p.wfc;
    const l2o_cfp bfc = (l2o_cfp)p.bfc;
    for (int u = 0; u < in0; ++u) {
      const float a = __builtin_fmaf(f1, wfc[in0 + u], __builtin_fmaf(f0, wfc[u], bfc[u]));
      p.du[n * (long)in0 + u] = dh[u] * (a > 0.0f ? 1.0f : expf(a));
    }
  } else if (p.dg) {
    float dgv = dh[0];
    if (p.pre == L2O_PRE_LOGSIGN) {                         // DM/preprocess.py:63-70, both clamps
      const float ag = fabsf(gv) + 1.1920928955078125e-07f;
      const float d0 = logf(ag) * p.k_inv > -1.0f ? copysignf(p.k_inv / ag, gv) : 0.0f;
      const float d1 = fabsf(gv * p.exp_k) < 1.0f ? p.exp_k : 0.0f;
      dgv = dh[0] * d0 + dh[1] * d1;
    }
    p.dg[n] = dgv;
  }
}
