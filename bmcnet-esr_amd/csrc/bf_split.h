// fp32 -> bf16 plane splitting shared by the bf16-plane kernels (conv_bf.hip, pgemm_bf.hip).
#pragma once
#include "bmc_common.h"

namespace {

typedef unsigned int u32;
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u32 pack_bf16(float a, float b) {   // v_cvt_pk_bf16_f32: RNE, low half = a
    return __builtin_bit_cast(u32, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ float lo_f(u32 p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_f(u32 p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// four fp32 values -> NP planes of four bf16 (two dwords per plane)
template <int NP>
__device__ __forceinline__ void split4(const f32x4 v, u32x2 (&pl)[NP]) {
    f32x4 r = v;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const u32 a = pack_bf16(r[0], r[1]), b = pack_bf16(r[2], r[3]);
        pl[p] = u32x2{a, b};
        if (p + 1 < NP) {   // exact residual (Sterbenz: the bf16 value shares the leading bits of r)
            r[0] -= lo_f(a); r[1] -= hi_f(a); r[2] -= lo_f(b); r[3] -= hi_f(b);
        }
    }
}

}  // namespace
