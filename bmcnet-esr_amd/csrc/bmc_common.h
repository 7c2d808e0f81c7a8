// Shared helpers for the gfx950 kernels of libbmc_hip.so (internal; the public ABI is include/bmc_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "bmc_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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
static inline SrcDev to_dev(const bmc_src_t& s) {
    SrcDev d;
    d.ptr = s.ptr; d.batch_stride = s.batch_stride; d.pix_stride = s.pix_stride; d.nch = s.nch;
    d.batch_shift = s.batch_shift; d.batch_mod = s.batch_mod < 1 ? 1 : s.batch_mod;
    return d;
}
__device__ __forceinline__ const float* src_batch_ptr(const SrcDev& s, int b) {
    int bs = b + s.batch_shift;
    if (s.batch_mod > 0) bs %= s.batch_mod;
    return s.ptr + (long long)bs * s.batch_stride;
}
