// Read-only streaming bandwidth of one MI355X, three ways (diagnostic for the pixel-reduction GEMM / 1x1 kernels, whose read
// streams sit near 3 TB/s):  (a) plain global_load_dwordx4 into registers, (b) global_load_lds_dwordx4 (LDS-DMA) with one tile in
// flight per workgroup, as pgemm.hip issues it, (c) LDS-DMA with several tiles in flight.   hipcc --offload-arch=gfx950 -O3
//   usage: hbm_read_bw [MB per buffer = 1024] [buffers = 2]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void plain_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b, long long n4, float* out) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const long long stride = (long long)gridDim.x * 512 * 4;
    for (long long i = (long long)blockIdx.x * 512 * 4 + threadIdx.x; i < n4; i += stride) {
        f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long j = i + (long long)k * 512;
            v[2 * k] = j < n4 ? a[j] : s;
            v[2 * k + 1] = j < n4 ? b[j] : s;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[k];
    }
    if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[0] = s[0];
}

// LDS-DMA: workgroup = 512 threads; one "tile" = 32 KB of a + 32 KB of b (4 + 4 wave-instructions of 1 KB per wave), DEPTH tiles in
// flight, tiles dealt round-robin to the workgroups (as pgemm.hip).  Nothing reads the LDS.
template <int DEPTH>
__global__ __launch_bounds__(512) void dma_kernel(const float* __restrict__ a, const float* __restrict__ b, long long ntiles, float* out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = tid >> 6;
    auto issue = [&](long long tile, int buf) {
        const float* pa = a + tile * 8192;
        const float* pb = b + tile * 8192;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(lds + (2 * buf) * 8192 + (i * 512 + wave * 64) * 4);
            const unsigned lb = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(lds + (2 * buf + 1) * 8192 + (i * 512 + wave * 64) * 4);
            const unsigned off = (unsigned)((i * 512 + tid) * 16);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(pa), "s"(__builtin_amdgcn_readfirstlane(la)) : "memory");
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(pb), "s"(__builtin_amdgcn_readfirstlane(lb)) : "memory");
        }
    };
    long long t = blockIdx.x;
    int issued = 0;
    for (int k = 0; k < DEPTH; ++k)
        if (t + (long long)k * gridDim.x < ntiles) { issue(t + (long long)k * gridDim.x, k); ++issued; }
    int it = 0;
    for (; t < ntiles; t += gridDim.x, ++it) {
        // oldest tile landed (8 instructions per tile and wave)
        const int younger = issued - (it + 1);
        if (younger >= 3) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if (younger == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const long long nx = t + (long long)DEPTH * gridDim.x;
        if (nx < ntiles) { issue(nx, it % DEPTH); ++issued; }
    }
    if (lds[tid] == 12345.678f) out[0] = lds[tid];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const long long mb = argc > 1 ? atoll(argv[1]) : 1024;
    const long long bytes = mb << 20, n4 = bytes / 16, ntiles = bytes / 32768;
    float *a, *b, *out;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto&& launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        const int iters = 20;
        for (int i = 0; i < iters; ++i) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-58s %8.1f us   %6.2f TB/s\n", name, ms / iters * 1e3, 2.0 * bytes / (ms / iters * 1e-3) / 1e12);
    };
    for (int wgs : {256, 512, 1024, 2048})
        timeit(("plain loads, " + std::to_string(wgs) + " workgroups x 512").c_str(), [&] { hipLaunchKernelGGL(plain_kernel, dim3(wgs), dim3(512), 0, 0, (const f32x4*)a, (const f32x4*)b, n4, out); });
    timeit("LDS-DMA, 256 workgroups, 1 tile (64 KB) in flight", [&] { hipLaunchKernelGGL(dma_kernel<1>, dim3(256), dim3(512), 65536, 0, a, b, ntiles, out); });
    timeit("LDS-DMA, 256 workgroups, 2 tiles (128 KB) in flight", [&] { hipLaunchKernelGGL(dma_kernel<2>, dim3(256), dim3(512), 131072, 0, a, b, ntiles, out); });
    timeit("LDS-DMA, 512 workgroups (2 per CU), 1 tile each", [&] { hipLaunchKernelGGL(dma_kernel<1>, dim3(512), dim3(512), 65536, 0, a, b, ntiles, out); });
    CK(hipDeviceSynchronize());
    return 0;
}
