// Splitting fp32 values in two fp16 halves for the split-operand matrix products (device code only).
#pragma once
#include <hip/hip_runtime.h>

namespace danra {

// (x0, x1) -> hi = the pair rounded towards zero to fp16 (one packed conversion), lo = fp16(x - hi), rounded to nearest:
// x = hi + lo to 22 significant bits while |x| is in [2^-3, 65504]; for a smaller |x| lo is an fp16 subnormal and x is
// kept to an absolute 2^-25.  Four instructions per pair: the mixed-precision FMA reads the fp16 half of `hi` as an fp32
// operand and writes its result straight to a half of `lo`.  (Operands here are VALU results; an operand that is a
// matrix-instruction result would need its wait states in front of the asm statements.)
__device__ __forceinline__ void split_pair_f16(float x0, float x1, unsigned& hi, unsigned& lo) {
    typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));
    const fp16x2_t h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
    hi = __builtin_bit_cast(unsigned, h);
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(x1));
}

// (x0, x1) -> h = the pair rounded towards zero to fp16, l = fp16(2^11 (x - h)): x = h + 2^-11 l, the split of the
// attention core's K and V (the scaled remainder stays a normal fp16 number for |x| down to 2^-14).  Five instructions
// per pair.
__device__ __forceinline__ void split_scaled_pair_f16(float x0, float x1, unsigned& h, unsigned& l) {
    typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));
    h = __builtin_bit_cast(unsigned, (fp16x2_t)__builtin_amdgcn_cvt_pkrtz(x0, x1));
    const float m2048 = -2048.f, s0 = x0 * 2048.f, s1 = x1 * 2048.f;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(m2048), "v"(s0));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(m2048), "v"(s1));
}

}  // namespace danra
