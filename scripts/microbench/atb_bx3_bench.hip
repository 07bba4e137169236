// k_atb (fp32 matrix pipe) against k_atb_bx3 (bf16 x 3) of csrc/l2o_atb.h on the weight-gradient contraction of a
// config-2 training step (R = T * B * D rows, KA x KB = 82 x 161, the needed blocks only): time per launch, the
// achieved HBM rate (4 (KA + KB) bytes per row, both operands read once) and the error of both against a float64
// product (host, OpenMP).  Standalone so that the kernel can be iterated on without the 2-minute library build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize \
//         -fopenmp -I open_l2o_amd/csrc -I include scripts/microbench/atb_bx3_bench.hip -o scripts/microbench/atb_bx3_bench
//   ./atb_bx3_bench [rows = 1638400] [mask = 1] [workgroups per CU]      (-DL2O_ATB_ABLATE=1..4: see l2o_atb.h)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "l2o_common.h"
#include "l2o_lstm_bx3.h"
using l2o::f32x4;
#include "l2o_atb.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef void (*kern_t)(const float*, const float*, long, int, int, float*);

int main(int argc, char** argv) {
  const long R = argc > 1 ? atol(argv[1]) : 1638400;
  const int mask = argc > 2 ? atoi(argv[2]) : 1;
  const int wgs_override = argc > 3 ? atoi(argv[3]) : 0;      // workgroups per CU of k_atb_bx3 (default: what its LDS allows)
  const int KA = mask == 2 ? 103 : 82, KB = mask == 2 ? 181 : 161, NT = mask == 2 ? 12 : 11;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  std::vector<float> A((size_t)R * KA), B((size_t)R * KB);
  // activations in (-1, 1) with a common-sign bias column, gate gradients with a wide dynamic range
#pragma omp parallel for schedule(static)
  for (long r = 0; r < R; ++r) {
    unsigned long long st = 0x9E3779B97F4A7C15ull * (unsigned long long)(r + 1);
    auto uni = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((st >> 40) * (1.0 / 16777216.0)); };
    auto nrm = [&]() { return (uni() + uni() + uni() + uni() - 2.0f) * 1.7320508f; };
    for (int c = 0; c < KA; ++c) A[(size_t)r * KA + c] = c == KA - 1 ? 1.0f : std::tanh(nrm());
    const float s = std::exp(2.0f * nrm());
    for (int c = 0; c < KB; ++c) B[(size_t)r * KB + c] = 1e-3f * s * nrm();
  }
  float *dA, *dB, *part, *out;
  CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4));
  CK(hipMalloc(&part, sizeof(float) * (size_t)kAtbMaxGroups * KA * KB)); CK(hipMalloc(&out, sizeof(float) * KA * KB));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  // float64 reference on the needed tiles
  std::vector<double> ref((size_t)KA * KB, 0.0), mag((size_t)KA * KB, 0.0);
#pragma omp parallel for schedule(dynamic, 1)
  for (int i = 0; i < KA; ++i)
    for (long r = 0; r < R; ++r) {
      const double a = A[(size_t)r * KA + i];
      const float* b = &B[(size_t)r * KB];
      for (int j = 0; j < KB; ++j)
        if (atb_needed(mask, i >> 4, j >> 4)) {
          const double p = a * (double)b[j];
          ref[(size_t)i * KB + j] += p;
          mag[(size_t)i * KB + j] += std::fabs(p);
        }
    }
  struct V { const char* name; kern_t fn; int wgs; };
  V vs[2];
  if (mask == 2) {
    vs[0] = {"k_atb     <7,12,2> fp32 pipe", k_atb<7, 12, 2>, L2O_ATB_WGS_PER_CU};
    vs[1] = {"k_atb_bx3 <7,12,2> bf16 x 3 ", k_atb_bx3<7, 12, 2>, atb_bx3_wgs_per_cu<7, 12>()};
  } else {
    vs[0] = {"k_atb     <6,11,1> fp32 pipe", k_atb<6, 11, 1>, L2O_ATB_WGS_PER_CU};
    vs[1] = {"k_atb_bx3 <6,11,1> bf16 x 3 ", k_atb_bx3<6, 11, 1>, atb_bx3_wgs_per_cu<6, 11>()};
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int n = KA * KB;
  std::vector<float> got(n);
  for (auto& v : vs) {
    const long nblk = (R + kAtbRows - 1) / kAtbRows;
    int groups = (wgs_override && &v == &vs[1] ? wgs_override : v.wgs) * cus;
    if (groups > kAtbMaxGroups) groups = kAtbMaxGroups;
    if (nblk < groups) groups = (int)nblk;
    float best = 1e30f, best_red = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(v.fn, dim3(groups), dim3(256), 0, 0, dA, dB, R, KA, KB, part);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep && ms < best) best = ms;
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_atb_reduce, dim3((n + 31) / 32), dim3(256), 0, 0, part, groups, n, out, mask, NT, KB);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep && ms < best_red) best_red = ms;
    }
    CK(hipGetLastError());
#if L2O_ATB_ABLATE == 5
    if (&v == &vs[1]) {                                   // sums over all waves and the 6 launches above
      unsigned long long ph[8];
      CK(hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_atb_phase), sizeof(ph)));
      const double it = (double)ph[5];
      const char* nm[5] = {"fetch issue", "multiply", "barrier 1", "split + stage (incl. load wait)", "barrier 2"};
      double tot = 0;
      for (int k = 0; k < 5; ++k) tot += ph[k] / it;
      for (int k = 0; k < 5; ++k) printf("  phase %-32s %8.0f ticks per wave and block (%.0f %%)\n", nm[k], ph[k] / it, 100.0 * ph[k] / it / tot);
      printf("  one block of one wave: %.0f ticks\n", tot);
    }
#endif
    CK(hipMemcpy(got.data(), out, sizeof(float) * n, hipMemcpyDeviceToHost));
    // error in units of the entry's sum of |products| (what an fp32 dot product's error scales with)
    double worst = 0, worst_abs = 0, rms = 0;
    long cnt = 0;
    for (int i = 0; i < KA; ++i)
      for (int j = 0; j < KB; ++j)
        if (atb_needed(mask, i >> 4, j >> 4) && mag[(size_t)i * KB + j] > 0) {
          const double e = std::fabs((double)got[i * KB + j] - ref[(size_t)i * KB + j]);
          const double rel = e / mag[(size_t)i * KB + j];
          worst = std::fmax(worst, rel); worst_abs = std::fmax(worst_abs, e); rms += rel * rel; ++cnt;
        }
    const double bytes = 4.0 * (KA + KB) * (double)R;
    printf("%s  groups %4d  %8.1f us  %6.2f TB/s (%.3f of 8)   reduce %6.1f us   err / sum|products|: max %.3g rms %.3g (max abs %.3g)\n",
           v.name, groups, best * 1e3, bytes / (best * 1e-3) / 1e12, bytes / (best * 1e-3) / 8e12, best_red * 1e3, worst,
           std::sqrt(rms / cnt), worst_abs);
  }
  return 0;
}
