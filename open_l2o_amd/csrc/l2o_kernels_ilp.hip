// l2o_kernels_ilp.hip -- the second translation unit of libl2o_hip.so: the kernels listed in l2o_ilp_kernels.h, compiled with
// -mllvm -amdgpu-sched-strategy=max-ilp (csrc/Makefile).  It re-reads l2o_kernels.hip up to that list (argument structs,
// device helpers, the kernel templates) and nothing behind it: no C-ABI entry point, no other kernel is emitted here.
#define L2O_TU_ILP 1
#include "l2o_kernels.hip"
