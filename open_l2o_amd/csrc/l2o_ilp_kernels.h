// l2o_ilp_kernels.h -- the kernels that are compiled in a SECOND translation unit (l2o_kernels_ilp.hip) under hipcc's
// max-ILP scheduling strategy (-mllvm -amdgpu-sched-strategy=max-ilp, a per-TU switch).  Included by l2o_kernels.hip right
// behind the headers that define them: in the main TU as `extern template` declarations (no code there; the launchers bind
// to the host stubs of the other object), in the ILP TU as explicit instantiations.
//
// Why (round 6, profiles/r06_frag_pipeline_ab.txt): where ONE wave per SIMD walks a dependent chain -- the plain two-CU
// unroll, the optimizer phase of the streaming unroll -- the strategy is worth 1.2-2.1 % (config 2 9.10 -> 9.21 G, config 3
// 6.16 -> 6.23 G, config 4's shard of 8 6.82 -> 6.96 G; three alternating runs each, results bit-identical); on
// k_unroll_lds it costs 9.5 %, the RECORDING two-CU unroll gets 2 % slower, k_mlp_xcd does not move.  So only the plain
// (HIST = false, EXACT = false) two-CU unrolls and the plain eight-wave streaming unrolls live here.
#pragma once

#ifdef L2O_TU_ILP
#define L2O_ILP_INST template
#else
#define L2O_ILP_INST extern template
#endif

#define L2O_ILP_PAIR(PRE, KIND)                                                            \
  L2O_ILP_INST __global__ void k_unroll_pair<PRE, KIND, 2, false, false>(UnrollPairArgs);  \
  L2O_ILP_INST __global__ void k_unroll_pair<PRE, KIND, 4, false, false>(UnrollPairArgs);  \
  L2O_ILP_INST __global__ void k_unroll_pair<PRE, KIND, 8, false, false>(UnrollPairArgs);
#define L2O_ILP_PAIR_NET(PRE) \
  L2O_ILP_PAIR(PRE, L2O_PROB_QUADRATIC) L2O_ILP_PAIR(PRE, L2O_PROB_LASSO) L2O_ILP_PAIR(PRE, L2O_PROB_RASTRIGIN) \
  L2O_ILP_PAIR(PRE, L2O_PROB_SQUARE_COS)
L2O_ILP_PAIR_NET(L2O_PRE_IDENTITY)
L2O_ILP_PAIR_NET(L2O_PRE_LOGSIGN)
L2O_ILP_PAIR_NET(L2O_PRE_FC_ELU)

namespace l2o {
#define L2O_ILP_CU8(PRE, KR)                                                         \
  L2O_ILP_INST __global__ void k_unroll_cu8<PRE, 1, KR, false>(UnrollArgs);          \
  L2O_ILP_INST __global__ void k_unroll_cu8<PRE, 2, KR, false>(UnrollArgs);
#define L2O_ILP_CU8_NET(PRE) L2O_ILP_CU8(PRE, 2) L2O_ILP_CU8(PRE, 3) L2O_ILP_CU8(PRE, 4)
L2O_ILP_CU8_NET(L2O_PRE_IDENTITY)
L2O_ILP_CU8_NET(L2O_PRE_LOGSIGN)
L2O_ILP_CU8_NET(L2O_PRE_FC_ELU)
}  // namespace l2o
