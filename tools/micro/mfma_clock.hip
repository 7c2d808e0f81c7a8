// Sustained fp32 MFMA rate and clock of one MI355X under pure matrix load (no memory traffic): 256 CUs x 8 waves, every wave a
// chain of independent v_mfma_f32_32x32x2_f32 (4 accumulators) and, second kernel, v_mfma_f32_16x16x4_f32.  Reports TFLOP/s and the
// clock that rate implies (256 FLOP / clk / CU), next to the shader clock counted by s_memtime over the same interval
// (s_memtime ticks at a constant 100 MHz: cycles = ticks x f / 100 MHz cannot be had from it; the rate is the measurement).
//   hipcc --offload-arch=gfx950 -O3 ; usage: mfma_clock [iterations per wave = 20000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool DATA>
__global__ __launch_bounds__(512) void mfma32(int iters, float* out) {
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    // operands with busy mantissas, different in every lane, sign-alternating between the accumulators (the sums stay bounded):
    // matrix-pipe power depends on the data
    const float x = __int_as_float(0x3f800000u | ((threadIdx.x * 2654435761u) & 0x7fffffu));
    const float y = __int_as_float(0x3f000000u | ((threadIdx.x * 40503u + blockIdx.x * 977u) & 0x7fffffu)) * (DATA ? 1.f : 0.f) + (DATA ? 0.f : 1e-9f);
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(-x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, -x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a3, 0, 0, 0);
    }
    const f32x16 s = a0 + a1 + a2 + a3;
    if (s[0] == 12345.678f) out[0] = s[1];
}

__global__ __launch_bounds__(512) void mfma16(int iters, float* out) {
    f32x4 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    const float x = (float)threadIdx.x * 1e-9f, y = 1e-9f;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
    }
    const f32x4 s = a0 + a1 + a2 + a3;
    if (s[0] == 12345.678f) out[0] = s[1];
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float* out;
    if (hipMalloc(&out, 64) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which) {
        const double flop_per_mfma = which == 0 ? 2.0 * 32 * 32 * 2 : 2.0 * 16 * 16 * 4;
        for (int reps : {1, 20, 200}) {        // one launch, then back-to-back launches: the clock under sustained load
            auto launch = [&] {
                if (which == 0) hipLaunchKernelGGL(mfma32<true>, dim3(256), dim3(512), 0, 0, iters, out);
                else hipLaunchKernelGGL(mfma16, dim3(256), dim3(512), 0, 0, iters * 4, out);
            };
            launch();
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            for (int r = 0; r < reps; ++r) launch();
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double n_mfma = 256.0 * 8 * 4 * (which == 0 ? iters : iters * 4) * reps;
            const double tflops = n_mfma * flop_per_mfma / (ms * 1e-3) / 1e12;
            printf("%s, %3d launches of %.0f us: %6.1f TFLOP/s  -> %.2f GHz at 256 FLOP/clk/CU\n", which == 0 ? "v_mfma_f32_32x32x2_f32 (busy operands)" : "v_mfma_f32_16x16x4_f32 (near-zero operands)",
                   reps, ms / reps * 1e3, tflops, tflops * 1e12 / (256.0 * 256.0) / 1e9);
        }
    }
    return 0;
}
