// Include FIRST in a translation unit whose device code must not contain packed fp32 instructions (common.h explains why):
// every function parsed after this point -- the unit's kernels AND the inline device functions of the HIP headers they call --
// gets target("no-packed-fp32-ops").  The attribute has to cover the header functions too: LLVM refuses to inline a callee
// whose target features are not a subset of the caller's, so with the attribute on the kernels alone (round 3) __shfl_xor,
// __uint_as_float, __syncthreads ... stayed out-of-line CALLS inside the hot loops (1852 s_swappc_b64 in uplift.o).
// Close with no_packed_fp32_end.h at the end of the unit.  (No include guard: a bracket, not a declaration.)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang attribute push(__attribute__((target("no-packed-fp32-ops"))), apply_to = function)
#endif
