/* l2o_oracle.c -- plain C (C99 + OpenMP) restatement of the Open-L2O inner unroll loop.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): a second, independent CPU
 * restatement used (a) to cross-check the NumPy oracle and (b) as the multi-threaded
 * "port" CPU baseline that bench.py times next to the GPU.  Never linked into, or
 * called from, the product package.
 *
 * Reference semantics, file:line under
 * /root/reference/Model_Free_L2O/"L2O-DM and L2O-RNNProp"/ (shorthand DM/):
 *   unroll           DM/meta.py:338-376 ; RNNProp inputs DM/meta_rnnprop_eval.py (update)
 *   network          DM/networks.py:207-232 (preprocess -> DeepRNN(LSTM,LSTM) -> Linear -> scale)
 *   LogAndSign       DM/preprocess.py:63-70
 *   snt.LSTM         dm-sonnet 1.11 (not in tree): gates i,j,f,o ; forget_bias 1
 *   optimizees       DM/problems.py:98-99 (quadratic), :128-131 (lasso), :206-211 (rastrigin)
 * All arithmetic in float (fp32), libm expf/tanhf/logf/cosf/sinf, no fast-math.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define H 20
#define G 80

typedef struct {
  int32_t rnnprop;      /* 0 CoordinateWiseDeepLSTM, 1 RNNprop */
  int32_t pre;          /* 0 identity, 1 LogAndSign, 2 fc+ELU */
  int32_t tanh_output;
  int32_t P;            /* LSTM-1 input width: 1, 2 or 20 */
  double scale, k, beta1, beta2;
  const float *wg1, *bg1, *wg2, *bg2, *wl, *bl, *wfc, *bfc;
} c_net;

typedef struct {
  int32_t kind;         /* 1 quadratic, 2 lasso, 3 rastrigin, 4 square_cos (C = column sums of wcos) */
  int32_t B, B_global, D, M;
  double l1, alpha;
  const float *W, *y, *C, *x_scale;
} c_prob;

static inline float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

static void lstm_cell(const float* in, int nin, float* h, float* c, const float* wg, const float* bg) {
  float z[G];
  memcpy(z, bg, sizeof(z));
  for (int k = 0; k < nin; ++k) {
    const float a = in[k];
    const float* w = wg + (size_t)k * G;
    for (int j = 0; j < G; ++j) z[j] += a * w[j];
  }
  for (int k = 0; k < H; ++k) {
    const float a = h[k];
    const float* w = wg + (size_t)(nin + k) * G;
    for (int j = 0; j < G; ++j) z[j] += a * w[j];
  }
  for (int u = 0; u < H; ++u) {
    const float cn = sigm(z[2 * H + u] + 1.0f) * c[u] + sigm(z[u]) * tanhf(z[H + u]);
    c[u] = cn;
    h[u] = tanhf(cn) * sigm(z[3 * H + u]);
  }
}

/* one optimizer step for one coordinate; returns delta */
static float net_coord(const c_net* n, float g, float mt, float* h1, float* c1, float* h2, float* c2) {
  float in[H];
  int nin;
  if (n->pre == 2) { /* ELU(Linear([m~, g~])) DM/networks.py:218-219 */
    nin = H;
    for (int u = 0; u < H; ++u) {
      const float v = mt * n->wfc[u] + g * n->wfc[H + u] + n->bfc[u];
      in[u] = v > 0.0f ? v : expm1f(v);
    }
  } else if (n->pre == 1) {
    nin = 2;
    const float eps = 1.1920928955078125e-07f;
    const float lg = logf(fabsf(g) + eps) / (float)n->k;
    in[0] = lg > -1.0f ? lg : -1.0f;
    float sg = g * (float)exp(n->k);
    in[1] = sg < -1.0f ? -1.0f : (sg > 1.0f ? 1.0f : sg);
  } else {
    nin = 1;
    in[0] = g;
  }
  lstm_cell(in, nin, h1, c1, n->wg1, n->bg1);
  lstm_cell(h1, H, h2, c2, n->wg2, n->bg2);
  float d = n->bl[0];
  for (int u = 0; u < H; ++u) d += h2[u] * n->wl[u];
  if (n->tanh_output) d = tanhf(d);
  return d * (float)n->scale;
}

/* f_b and (optionally) gradient of one problem at xs (already scaled) */
static float prob_fg(const c_prob* p, int b, const float* xs, float* r, float* g) {
  const int D = p->D, M = p->M;
  const float* W = p->W + (size_t)b * M * D;
  const float* y = p->y + (size_t)b * M;
  float f = 0.0f;
  for (int i = 0; i < M; ++i) {
    float acc = 0.0f;
    const float* row = W + (size_t)i * D;
    for (int j = 0; j < D; ++j) acc += row[j] * xs[j];
    r[i] = acc - y[i];
    f += r[i] * r[i];
  }
  if (p->kind != 1 && p->kind != 4) f *= 0.5f;
  const float twopi = p->kind == 4 ? (float)(2 * 3.1415926) : 6.2831853071795864769f;
  const float alpha = p->kind == 4 ? 10.0f : (float)p->alpha;
  if (p->kind == 2)
    for (int j = 0; j < D; ++j) f += (float)p->l1 * fabsf(xs[j]);
  if (p->kind == 3 || p->kind == 4) {
    const float* C = p->C + (size_t)b * D;
    float cq = 0.0f;
    for (int j = 0; j < D; ++j) cq += C[j] * cosf(twopi * xs[j]);
    f += -alpha * cq + alpha * (float)D;
  }
  if (g) {
    for (int j = 0; j < D; ++j) g[j] = 0.0f;
    for (int i = 0; i < M; ++i) {
      const float ri = r[i];
      const float* row = W + (size_t)i * D;
      for (int j = 0; j < D; ++j) g[j] += row[j] * ri;
    }
    const float inv = 1.0f / (float)p->B_global;
    for (int j = 0; j < D; ++j) {
      float gj = (p->kind == 1 || p->kind == 4) ? 2.0f * g[j] : g[j];
      if (p->kind == 2) gj += (float)p->l1 * (xs[j] > 0.0f ? 1.0f : (xs[j] < 0.0f ? -1.0f : 0.0f));
      if (p->kind == 3 || p->kind == 4) gj += twopi * alpha * p->C[(size_t)b * D + j] * sinf(twopi * xs[j]);
      g[j] = gj * inv;
    }
  }
  return f;
}

/* The whole unroll.  x [B,D], state h1,c1,h2,c2 [B*D,20], m,v [B,D] (RNNProp) are updated
 * in place; fx[0..T] receives sum_b f_b / B_global.  Returns the number of threads used. */
int l2o_c_unroll(const c_net* n, const c_prob* p, float* x, float* h1, float* c1, float* h2, float* c2, float* m,
                 float* v, int T, int step0, float* fx) {
  const int B = p->B, D = p->D, M = p->M;
  float* fpart = (float*)calloc((size_t)(T + 1) * B, sizeof(float));
  int nthreads = 1;
#pragma omp parallel
  {
#ifdef _OPENMP
#pragma omp single
    nthreads = omp_get_num_threads();
#endif
    float* xs = (float*)malloc(sizeof(float) * D);
    float* g = (float*)malloc(sizeof(float) * D);
    float* r = (float*)malloc(sizeof(float) * M);
#pragma omp for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
      float* xb = x + (size_t)b * D;
      const float* sb = p->x_scale ? p->x_scale + (size_t)b * D : NULL;
      for (int t = 0; t <= T; ++t) {
        for (int j = 0; j < D; ++j) xs[j] = sb ? xb[j] * sb[j] : xb[j];
        fpart[(size_t)t * B + b] = prob_fg(p, b, xs, r, t < T ? g : NULL);
        if (t == T) break;
        const float kf = (float)(step0 + t);
        for (int j = 0; j < D; ++j) {
          const size_t c = (size_t)b * D + j;
          float gj = sb ? g[j] * sb[j] : g[j];
          float mt = 0.0f;
          if (n->rnnprop) { /* DM/meta_rnnprop_train.py:383-388 */
            const float b1 = (float)n->beta1, b2 = (float)n->beta2;
            m[c] = b1 * m[c] + (float)(1.0 - n->beta1) * gj;
            v[c] = b2 * v[c] + (float)(1.0 - n->beta2) * gj * gj;
            const float mh = m[c] / (1.0f - powf(b1, kf));
            const float vh = v[c] / (1.0f - powf(b2, kf));
            const float den = sqrtf(vh) + 1e-8f;
            mt = mh / den;
            gj = gj / den;
          }
          xb[j] += net_coord(n, gj, mt, h1 + c * H, c1 + c * H, h2 + c * H, c2 + c * H);
        }
      }
    }
    free(xs); free(g); free(r);
  }
  for (int t = 0; t <= T; ++t) {
    float s = 0.0f;
    for (int b = 0; b < B; ++b) s += fpart[(size_t)t * B + b];
    fx[t] = s / (float)p->B_global;
  }
  free(fpart);
  return nthreads;
}
