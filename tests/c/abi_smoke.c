/* abi_smoke.c -- a plain-C client of include/l2o_abi.h (no Python, no torch): what a maintainer's
 * C / cgo / JNI binding would do.  Built and run by tests/test_c_abi.py.
 *
 *   abi_smoke --host   host-only entry points (version, options, weight packer, argument checks)
 *   abi_smoke          + on the GPU: one fused unroll (l2o_unroll, L2O-DM on Quadratic B=4, D=16, T=5)
 *                      checked against (a) f(x_0) computed here in C from the definition
 *                      (DM/problems.py:98-99) and (b) the step-granular entry points
 *                      (l2o_problem_fg + l2o_cwlstm_step per step) on the same inputs.
 * Exit code 0 = all checks passed. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "l2o_abi.h"

#define CHECK(cond, ...)                      \
  do {                                        \
    if (!(cond)) {                            \
      fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
      fprintf(stderr, __VA_ARGS__);           \
      fprintf(stderr, "\n");                  \
      return 1;                               \
    }                                         \
  } while (0)
#define HIP(expr) CHECK((expr) == hipSuccess, "%s", #expr)
#define L2O(expr) CHECK((expr) == L2O_OK, "%s -> %s", #expr, l2o_last_error())

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static float urand(void) { /* xorshift64*, U[0, 1) */
  rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
  return (float)((rng_state * 0x2545F4914F6CDD1Dull) >> 40) / 16777216.0f;
}
static float nrand(void) { /* roughly N(0, 1): sum of 12 uniforms */
  float s = 0.f;
  for (int i = 0; i < 12; ++i) s += urand();
  return s - 6.0f;
}

enum { B = 4, D = 16, T = 5, H = 20 };

int main(int argc, char** argv) {
  const int host_only = argc > 1 && strcmp(argv[1], "--host") == 0;
  CHECK(l2o_abi_version() == L2O_ABI_VERSION, "ABI version %d != header %d", l2o_abi_version(), L2O_ABI_VERSION);
  /* (ABI v9: no process-wide option state -- the switches of a call travel in cfg.options, see (a'') below) */

  /* L2O-DM: CoordinateWiseDeepLSTM, layers (20, 20), identity preprocess, scale 1 (DM/util.py:138-142) */
  l2o_net_cfg cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.kind = L2O_NET_CW; cfg.preprocess = L2O_PRE_IDENTITY; cfg.n_layers = 2; cfg.hidden = H;
  cfg.scale = 1.0; cfg.beta1 = cfg.beta2 = 0.95;
  /* Sonnet layouts: w_gates [in + H, 4H], b_gates [4H], Linear w [H, 1], b [1] */
  static float wg1[(1 + H) * 4 * H], bg1[4 * H], wg2[2 * H * 4 * H], bg2[4 * H], wl[H], bl[1];
  for (int i = 0; i < (1 + H) * 4 * H; ++i) wg1[i] = nrand() / sqrtf(1.f + H);
  for (int i = 0; i < 2 * H * 4 * H; ++i) wg2[i] = nrand() / sqrtf(2.f * H);
  for (int i = 0; i < 4 * H; ++i) { bg1[i] = 0.1f * nrand(); bg2[i] = 0.1f * nrand(); }
  for (int i = 0; i < H; ++i) wl[i] = 0.1f * nrand() / sqrtf((float)H);
  bl[0] = 0.01f;
  const size_t nw = l2o_wpack_floats(&cfg);
  CHECK(nw > 0, "l2o_wpack_floats");
  float* wpack = (float*)malloc(nw * sizeof(float));
  L2O(l2o_wpack_host(&cfg, wg1, bg1, wg2, bg2, wl, bl, NULL, NULL, wpack));
  CHECK(l2o_wpack_host(&cfg, NULL, bg1, wg2, bg2, wl, bl, NULL, NULL, wpack) == L2O_ERR_ARG, "NULL weights are rejected");
  l2o_net_cfg bad = cfg;
  bad.hidden = 7;
  CHECK(l2o_wpack_floats(&bad) == 0, "an unsupported net reports 0 floats");
  CHECK(l2o_state_floats(B, D) == (size_t)B * 1 * 4 * H * 16, "l2o_state_floats");
  CHECK(l2o_unroll(NULL, NULL, NULL, NULL, NULL, NULL, NULL, 1, 1, NULL, NULL, NULL) != L2O_OK && strlen(l2o_last_error()) > 0,
        "NULL arguments are rejected with a message");
  {
    /* ABI v13, host-side contracts of the multi-instance MLP unroll and the deep-MLP evaluation (no launches) */
    l2o_mlp mlp;
    memset(&mlp, 0, sizeof mlp);
    mlp.n_in = 784; mlp.n_hidden = 20; mlp.n_out = 10; mlp.batch = 64;
    CHECK(l2o_mlp_unroll_multi_workspace_bytes(&mlp, 8) > 8 * l2o_mlp_unroll_multi_workspace_bytes(&mlp, 1) / 2 &&
          l2o_mlp_unroll_multi_workspace_bytes(&mlp, 1) > 0, "l2o_mlp_unroll_multi_workspace_bytes grows with the instances");
    CHECK(l2o_mlp_unroll_multi_workspace_bytes(&mlp, 9) == 0, "more than eight instances per launch are refused");
    mlp.batch = 32;
    CHECK(l2o_mlp_unroll_multi_workspace_bytes(&mlp, 1) == 0, "other minibatch sizes have no one-XCD kernel");
    CHECK(l2o_mlp_unroll_multi(&cfg, wpack, &mlp, NULL, 1, 1, 1, NULL, NULL) != L2O_OK, "NULL instances are rejected");
    l2o_mlp_deep deep;
    memset(&deep, 0, sizeof deep);
    deep.n_in = 784; deep.n_out = 10; deep.batch = 128; deep.n_hidden_layers = 2; deep.hidden[0] = 20; deep.hidden[1] = 20;
    CHECK(l2o_mlp_deep_scratch_floats(&deep) == (size_t)128 * 32 * 5 + 128, "l2o_mlp_deep_scratch_floats");
    deep.n_hidden_layers = 4;
    CHECK(l2o_mlp_deep_scratch_floats(&deep) == 0, "four hidden layers are refused");
  }
  if (host_only) {
    printf("abi_smoke: host-only checks passed (ABI v%d, wpack %zu floats)\n", l2o_abi_version(), nw);
    return 0;
  }

  /* ---- device part ---- */
  static float W[B * D * D], y[B * D], x0[B * D];
  for (int i = 0; i < B * D * D; ++i) W[i] = urand();       /* DM/problems.py:84-96 */
  for (int i = 0; i < B * D; ++i) { y[i] = urand(); x0[i] = 0.01f * nrand(); }
  double f0 = 0.0;                                           /* f = mean_b sum_i (W x - y)_i^2, :98-99 */
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < D; ++i) {
      double r = -y[b * D + i];
      for (int j = 0; j < D; ++j) r += (double)W[(b * D + i) * D + j] * x0[b * D + j];
      f0 += r * r;
    }
  f0 /= B;

  float *dW, *dy, *dx, *dst, *dfxp, *dfx, *dwp, *dg, *dfp;
  void* dws = NULL;
  const size_t nst = l2o_state_floats(B, D);
  HIP(hipMalloc((void**)&dW, sizeof W)); HIP(hipMalloc((void**)&dy, sizeof y)); HIP(hipMalloc((void**)&dx, sizeof x0));
  HIP(hipMalloc((void**)&dst, nst * 4)); HIP(hipMalloc((void**)&dfxp, (T + 1) * B * 4)); HIP(hipMalloc((void**)&dfx, (T + 1) * 4));
  HIP(hipMalloc((void**)&dwp, nw * 4)); HIP(hipMalloc((void**)&dg, sizeof x0)); HIP(hipMalloc((void**)&dfp, B * 4));
  HIP(hipMemcpy(dW, W, sizeof W, hipMemcpyHostToDevice)); HIP(hipMemcpy(dy, y, sizeof y, hipMemcpyHostToDevice));
  HIP(hipMemcpy(dwp, wpack, nw * 4, hipMemcpyHostToDevice));

  l2o_problem prob;
  memset(&prob, 0, sizeof prob);
  prob.kind = L2O_PROB_QUADRATIC; prob.B_local = B; prob.B_global = B; prob.D = D; prob.M = D;
  prob.W = dW; prob.y = dy;
  CHECK(l2o_unroll_supported(&cfg, &prob) == 1, "l2o_unroll_supported");
  const size_t wsb = l2o_unroll_workspace_bytes(&cfg, &prob, T);
  if (wsb) { HIP(hipMalloc(&dws, wsb)); HIP(hipMemset(dws, 0, wsb)); }

  float fx_fused[T + 1], fx_steps[T + 1], x_fused[B * D], x_steps[B * D];
  hipStream_t s;
  HIP(hipStreamCreate(&s));
  /* (a) the fused unroll */
  HIP(hipMemcpy(dx, x0, sizeof x0, hipMemcpyHostToDevice)); HIP(hipMemset(dst, 0, nst * 4));
  CHECK(l2o_unroll_workspace_layout(&cfg, &prob) == (wsb ? (((long long)B << 8) | 2) : 0), "workspace layout id");
  if (dws) L2O(l2o_unroll_workspace_init(dws, wsb, s));      /* (already zero: the call a caller makes when the layout changes) */
  L2O(l2o_unroll_reduce(&cfg, dwp, &prob, NULL, dx, dst, NULL, NULL, T, 1, 0, dfxp, dfx, dws, NULL, s));
  HIP(hipStreamSynchronize(s));
  HIP(hipMemcpy(fx_fused, dfx, sizeof fx_fused, hipMemcpyDeviceToHost)); HIP(hipMemcpy(x_fused, dx, sizeof x_fused, hipMemcpyDeviceToHost));
  if (dws) {
    unsigned status = 0;
    HIP(hipMemcpy(&status, dws, 4, hipMemcpyDeviceToHost));
    L2O(l2o_unroll_status(&status));
  }
  /* (a') the fault-injection word (workspace bytes 8..11, ABI v12): the two-CU kernel raises the sticky status at once,
   * l2o_unroll_status decodes it as the RECOVERABLE L2O_ERR_TIMEOUT, and the same unroll on the exchange-free kernel
   * (L2O_OPT_PAIR = 0) gives the trajectory of (a) -- what the Python host does on a partner timeout */
  if (dws && (l2o_last_unroll_form() & 0xff) == L2O_FORM_UNROLL_PAIR) {
    float fx_rec[T + 1];
    unsigned one = 1, status = 0, zero = 0;
    HIP(hipMemcpy((char*)dws + 8, &one, 4, hipMemcpyHostToDevice));
    HIP(hipMemcpy(dx, x0, sizeof x0, hipMemcpyHostToDevice)); HIP(hipMemset(dst, 0, nst * 4));
    L2O(l2o_unroll_reduce(&cfg, dwp, &prob, NULL, dx, dst, NULL, NULL, T, 1, 0, dfxp, dfx, dws, NULL, s));
    HIP(hipStreamSynchronize(s));
    HIP(hipMemcpy(&status, dws, 4, hipMemcpyDeviceToHost));
    CHECK(status != 0 && l2o_unroll_status(&status) == L2O_ERR_TIMEOUT, "injected fault: status %u not reported as L2O_ERR_TIMEOUT", status);
    HIP(hipMemcpy(dws, &zero, 4, hipMemcpyHostToDevice)); HIP(hipMemcpy((char*)dws + 8, &zero, 4, hipMemcpyHostToDevice));
    l2o_net_cfg c1 = cfg;
    c1.options = L2O_OPTW(L2O_OPT_PAIR, 0);
    HIP(hipMemcpy(dx, x0, sizeof x0, hipMemcpyHostToDevice)); HIP(hipMemset(dst, 0, nst * 4));
    L2O(l2o_unroll_reduce(&c1, dwp, &prob, NULL, dx, dst, NULL, NULL, T, 1, 0, dfxp, dfx, dws, NULL, s));
    HIP(hipStreamSynchronize(s));
    CHECK((l2o_last_unroll_form() & 0xff) != L2O_FORM_UNROLL_PAIR, "L2O_OPT_PAIR = 0 still ran the two-CU kernel");
    HIP(hipMemcpy(fx_rec, dfx, sizeof fx_rec, hipMemcpyDeviceToHost));
    for (int t = 0; t <= T; ++t)
      CHECK(fabsf(fx_rec[t] - fx_fused[t]) <= 1e-5f * fabsf(fx_fused[t]), "recovered unroll: fx[%d] = %g vs %g", t, fx_rec[t], fx_fused[t]);
  }
  /* (a'') kernel switches are PER CALL and caller-owned (cfg.options): the same unroll on the one-CU kernel and
   * with the exact (fp32 MFMA) gate GEMM -- all within the parity tolerance of (a), and
   * a later call with options = 0 is the default kernel again (the library kept nothing) */
  {
    const uint64_t variants[2] = {L2O_OPTW(L2O_OPT_PAIR, 0), L2O_OPTW(L2O_OPT_EXACT_GATES, 1)};
    for (int k = 0; k < 2; ++k) {
      float fx_v[T + 1];
      l2o_net_cfg c2 = cfg;
      c2.options = variants[k];
      HIP(hipMemcpy(dx, x0, sizeof x0, hipMemcpyHostToDevice)); HIP(hipMemset(dst, 0, nst * 4));
      L2O(l2o_unroll_reduce(&c2, dwp, &prob, NULL, dx, dst, NULL, NULL, T, 1, 0, dfxp, dfx, k == 0 ? NULL : dws, NULL, s));
      HIP(hipStreamSynchronize(s));
      HIP(hipMemcpy(fx_v, dfx, sizeof fx_v, hipMemcpyDeviceToHost));
      for (int t = 0; t <= T; ++t)
        CHECK(fabsf(fx_v[t] - fx_fused[t]) <= 1e-5f * fabsf(fx_fused[t]), "options variant %d: fx[%d] = %g vs %g", k, t, fx_v[t], fx_fused[t]);
    }
    float fx_again[T + 1];
    HIP(hipMemcpy(dx, x0, sizeof x0, hipMemcpyHostToDevice)); HIP(hipMemset(dst, 0, nst * 4));
    L2O(l2o_unroll_reduce(&cfg, dwp, &prob, NULL, dx, dst, NULL, NULL, T, 1, 0, dfxp, dfx, dws, NULL, s));
    HIP(hipStreamSynchronize(s));
    HIP(hipMemcpy(fx_again, dfx, sizeof fx_again, hipMemcpyDeviceToHost));
    CHECK(memcmp(fx_again, fx_fused, sizeof fx_again) == 0, "options = 0 after other calls is not the default kernel again");
    int32_t cus = l2o_coresident_workgroups(dfxp, s);
    CHECK(cus > 0, "l2o_coresident_workgroups = %d", cus);
  }
  /* (b) the same unroll through the step-granular entry points */
  HIP(hipMemcpy(dx, x0, sizeof x0, hipMemcpyHostToDevice)); HIP(hipMemset(dst, 0, nst * 4));
  for (int t = 0; t <= T; ++t) {
    L2O(l2o_problem_fg(&prob, dx, dfp, t < T ? dg : NULL, s));
    L2O(l2o_reduce_fx(dfp, 1, B, B, dfx + t, s));
    if (t < T) L2O(l2o_cwlstm_step(&cfg, dwp, dg, NULL, NULL, 0.0, 0.0, dst, dx, B, D, s));
  }
  HIP(hipStreamSynchronize(s));
  HIP(hipMemcpy(fx_steps, dfx, sizeof fx_steps, hipMemcpyDeviceToHost)); HIP(hipMemcpy(x_steps, dx, sizeof x_steps, hipMemcpyDeviceToHost));

  CHECK(fabs(fx_fused[0] - f0) <= 1e-5 * fabs(f0), "f(x_0): fused %.8g vs C definition %.8g", fx_fused[0], f0);
  double worst = 0.0;
  for (int t = 0; t <= T; ++t) {
    CHECK(isfinite(fx_fused[t]), "fx[%d] is not finite", t);
    const double e = fabs((double)fx_fused[t] - fx_steps[t]) / fabs((double)fx_steps[t]);
    if (e > worst) worst = e;
  }
  CHECK(worst <= 1e-5, "fused vs step-granular loss trajectory: rel err %.3g", worst);
  for (int i = 0; i < B * D; ++i) CHECK(fabs(x_fused[i] - x_steps[i]) <= 1e-5, "x_T[%d]: %.8g vs %.8g", i, x_fused[i], x_steps[i]);
  CHECK(fx_fused[T] != fx_fused[0], "the optimizer moved the iterate");
  printf("abi_smoke: l2o_unroll (T=%d) fx = [%.6g .. %.6g], f(x_0) matches the definition, fused vs step path rel err %.3g\n",
         T, fx_fused[0], fx_fused[T], worst);
  return 0;
}
