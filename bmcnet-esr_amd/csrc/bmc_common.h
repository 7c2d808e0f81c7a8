// Shared helpers for the gfx950 kernels of libbmc_hip.so (internal; the public ABI is include/bmc_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "bmc_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// 16 bytes from GLOBAL memory.  A pointer that reached the kernel through LDS or a select (descriptor tables, "this pixel
// or the zero constant") is a generic pointer to the compiler, and a load through it is a flat_load: that counts on vmcnt
// AND lgkmcnt and may return out of order, so every wait for one is vmcnt(0) + lgkmcnt(0) -- a prefetch ring of depth one.
// The address-space cast makes it a global_load, whose waits count.
__device__ __forceinline__ f32x4 ldg16(const void* p) {
    return *(const __attribute__((address_space(1))) f32x4*)(unsigned long long)p;
}
__device__ __forceinline__ float ldg4(const void* p) { return *(const __attribute__((address_space(1))) float*)(unsigned long long)p; }
__device__ __forceinline__ void stg4(void* p, float v) { *(__attribute__((address_space(1))) float*)(unsigned long long)p = v; }
__device__ __forceinline__ void stg16(void* p, f32x4 v) {
    *(__attribute__((address_space(1))) f32x4*)(unsigned long long)p = v;
}

void bmc_set_error(const char* fmt, ...);

#define BMC_CHECK_ARG(cond, ...)          \
    do {                                  \
        if (!(cond)) {                    \
            bmc_set_error(__VA_ARGS__);   \
            return -1;                    \
        }                                 \
    } while (0)

#define BMC_CHECK_LAUNCH(name)                                                  \
    do {                                                                        \
        hipError_t e__ = hipGetLastError();                                     \
        if (e__ != hipSuccess) {                                                \
            bmc_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return -2;                                                          \
        }                                                                       \
    } while (0)

static inline int bmc_round_up(int v, int m) { return (v + m - 1) / m * m; }
int bmc_num_cus(void);  // multiprocessor count of the current device (cached)

// Device view of bmc_src_t (same fields; kept POD so it can sit in kernel args).
struct SrcDev {
    const float* ptr;
    long long batch_stride;
    int pix_stride;
    int nch;
    int batch_shift;
    int batch_mod;
};
// the kernels keep tables of these in LDS, sized as 8 floats per entry (conv.hip, conv1.hip, chain.hip)
static_assert(sizeof(SrcDev) == 32 && alignof(SrcDev) == 8, "SrcDev must stay 32 bytes: LDS source tables are sized by it");
static inline SrcDev to_dev(const bmc_src_t& s) {
    SrcDev d;
    d.ptr = s.ptr; d.batch_stride = s.batch_stride; d.pix_stride = s.pix_stride; d.nch = s.nch;
    d.batch_shift = s.batch_shift; d.batch_mod = s.batch_mod == BMC_SRC_TABLE ? BMC_SRC_TABLE : (s.batch_mod < 1 ? 1 : s.batch_mod);
    return d;
}
__device__ __forceinline__ const float* src_batch_ptr(const SrcDev& s, int b) {
    int bs = b + s.batch_shift;
    if (s.batch_mod > 0) bs %= s.batch_mod;
    return s.ptr + (long long)bs * s.batch_stride;
}
// The same for the kernels that accept BMC_SRC_TABLE operands (the pixel-reduction GEMMs: pgemm.hip, pgemm_bf.hip).
__device__ __forceinline__ const float* src_batch_ptr_tab(const SrcDev& s, int b) {
    if (s.batch_mod == BMC_SRC_TABLE) {    // ptr = device table of per-image base pointers (bmc_ptr_table): operands gathered from several tensors
        // (the descriptor may differ from lane to lane -- lanes of one wave read columns of different sources: no readfirstlane here)
        const unsigned long long v = *(const __attribute__((address_space(1))) unsigned long long*)(
            reinterpret_cast<unsigned long long>(s.ptr) + 8ull * (unsigned)b);
        return reinterpret_cast<const float*>(v);
    }
    return src_batch_ptr(s, b);
}
// Compile-time selection: the table lookup is a compiler-visible vector-memory load, and one such load anywhere on a path into a
// loop of hand-counted LDS-DMA makes the compiler wait for ALL outstanding vector-memory operations there (pgemm<1>: 245 -> 264 us
// with the table path merely compiled in) -- so kernels exist once without it (TAB = false: every launch of a large frame) and once
// with it (the merged small-frame launches).
template <bool TAB>
__device__ __forceinline__ const float* src_bp(const SrcDev& s, int b) {
    if constexpr (TAB) return src_batch_ptr_tab(s, b);
    else return src_batch_ptr(s, b);
}
static inline bool src_is_table(const SrcDev& s) { return s.batch_mod == BMC_SRC_TABLE; }
// ... where descriptor and image are wave-uniform and the result feeds a scalar-base memory instruction
template <bool TAB>
__device__ __forceinline__ const float* src_bp_uni(const SrcDev& s, int b) {
    if constexpr (!TAB) return src_batch_ptr(s, b);
    const unsigned long long v = reinterpret_cast<unsigned long long>(src_batch_ptr_tab(s, b));
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
}
