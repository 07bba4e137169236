// What does a process see of a CU mask?  Prints the device's CU count, the stream's CU mask (hipExtStreamGetCUMask) and the
// MEASURED number of co-resident one-per-CU workgroups (the probe kernel of l2o_kernels.hip: every workgroup bumps a
// counter and waits -- bounded -- until all have arrived).
//   hipcc --offload-arch=gfx950 -O2 cu_mask_probe.hip -o cu_mask_probe && ./cu_mask_probe
//   ROC_GLOBAL_CU_MASK=0xffffffffffffffffffffffffffffffff ./cu_mask_probe ;  HSA_CU_MASK=0:0-127 ./cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>

__global__ __launch_bounds__(256) void k_probe(unsigned* ctr, unsigned* seen, int n) {
  extern __shared__ float big[];           // 100 KB of LDS: one workgroup per CU
  if (threadIdx.x == 0) {
    big[0] = 1.0f;
    atomicAdd(ctr, 1u);
    unsigned v = 0;
    for (int spin = 0; spin < (1 << 14); ++spin) {
      v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v >= (unsigned)n) break;
      __builtin_amdgcn_s_sleep(8);
    }
    atomicMax(seen, v);
    if (blockIdx.x == 0) seen[1] = v;      // what the FIRST workgroup saw when it gave up / was satisfied
  }
}

int main() {
  int dev = 0, ncu = 0;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  hipStream_t s;
  hipStreamCreate(&s);
  std::vector<uint32_t> mask(16, 0);
  hipError_t e = hipExtStreamGetCUMask(s, (uint32_t)mask.size(), mask.data());
  int pop = 0;
  for (auto w : mask) pop += __builtin_popcount(w);
  printf("device CUs %d | hipExtStreamGetCUMask(stream): %s, popcount %d", ncu, hipGetErrorName(e), pop);
  e = hipExtStreamGetCUMask(nullptr, (uint32_t)mask.size(), mask.data());
  pop = 0;
  for (auto w : mask) pop += __builtin_popcount(w);
  printf(" | (null stream): %s, popcount %d\n", hipGetErrorName(e), pop);
  {
    hipEvent_t t0, t1; (void)t0; (void)t1;
    auto c0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 1000; ++i) hipExtStreamGetCUMask(s, (uint32_t)mask.size(), mask.data());
    auto c1 = std::chrono::steady_clock::now();
    printf("hipExtStreamGetCUMask: %.3f us per call\n", std::chrono::duration<double, std::micro>(c1 - c0).count() / 1000.0);
  }
  unsigned* d;
  hipMalloc(&d, 16);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  for (int rep = 0; rep < 2; ++rep) {
    hipMemsetAsync(d, 0, 16, s);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, s);
    hipLaunchKernelGGL(k_probe, dim3(ncu), dim3(256), 100 * 1024, s, d, d + 1, ncu);
    hipEventRecord(b, s);
    hipStreamSynchronize(s);
    unsigned h[4];
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("probe %d: arrived %u, max seen %u, first workgroup saw %u of %d  (%.3f ms)\n", rep, h[0], h[1], h[2], ncu, ms);
  }
  int occ = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_probe, 256, 100 * 1024);
  printf("hipOccupancyMaxActiveBlocksPerMultiprocessor(k_probe) = %d\n", occ);
  return 0;
}
