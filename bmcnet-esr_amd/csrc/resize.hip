// Bicubic resize of the HR prediction to the ground-truth size, forward and backward (NCHW planes, fp32):
// F.interpolate(pred, size=gt.size()[-2:], mode='bicubic', align_corners=False) at train.py:227-231 and
// infer_BMCNet.py:77-78 -- taken whenever scale * round(sensor / scale) != sensor (EventZoom: 124x224 -> 124x222).
//
// ATen semantics, per axis: scale = in / out (float), src = fma(scale, dst + 0.5, -0.5) (not clamped),
// i0 = floor(src), t = src - i0, taps i0-1 .. i0+2 with the cubic-convolution weights for A = -0.75, tap indices
// clamped to [0, in-1].  HBM-bound and tiny (2 planes of 124x222 per sample): one thread per element.
// The backward is the transposed operator written as a GATHER (each input element sums the output elements whose
// taps touch it, clamped taps included), so it is deterministic -- no float atomics.
#include "bmc_common.h"

namespace {

constexpr float CUBIC_A = -0.75f;

__device__ __forceinline__ void cubic_taps(int dst, float scale, int& i0, float (&w)[4]) {
    const float src = fmaf(scale, (float)dst + 0.5f, -0.5f);
    const float fl = floorf(src);
    const float t = src - fl;
    i0 = (int)fl;
    auto c1 = [](float v) { return ((CUBIC_A + 2.f) * v - (CUBIC_A + 3.f)) * v * v + 1.f; };                     // |v| <= 1
    auto c2 = [](float v) { return ((CUBIC_A * v - 5.f * CUBIC_A) * v + 8.f * CUBIC_A) * v - 4.f * CUBIC_A; };   // 1 < |v| < 2
    w[0] = c2(t + 1.f); w[1] = c1(t); w[2] = c1(1.f - t); w[3] = c2(2.f - t);
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ void bicubic_fwd_kernel(const float* __restrict__ x, long long planes, int H, int W, int Ho, int Wo,
                                   float sy, float sx, float* __restrict__ y) {
    const long long total = planes * Ho * Wo;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int X = (int)(idx % Wo);
        const long long r = idx / Wo;
        const int Y = (int)(r % Ho);
        const float* const xp = x + (r / Ho) * (long long)H * W;
        int iy0, ix0;
        float wy[4], wx[4];
        cubic_taps(Y, sy, iy0, wy);
        cubic_taps(X, sx, ix0, wx);
        int cx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) cx[k] = clampi(ix0 - 1 + k, 0, W - 1);
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* const row = xp + (long long)clampi(iy0 - 1 + j, 0, H - 1) * W;
            // ATen's order: the four taps of a row first, then the rows
            const float rv = row[cx[0]] * wx[0] + row[cx[1]] * wx[1] + row[cx[2]] * wx[2] + row[cx[3]] * wx[3];
            acc += rv * wy[j];
        }
        y[idx] = acc;
    }
}

// weight with which output index `dst` reads input index `src_i` along one axis (sum over its clamped taps)
__device__ __forceinline__ float axis_weight(int dst, float scale, int src_i, int n_in) {
    int i0;
    float w[4];
    cubic_taps(dst, scale, i0, w);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (clampi(i0 - 1 + k, 0, n_in - 1) == src_i) s += w[k];
    return s;
}
// conservative range of output indices whose taps can touch input index i (the exact test is axis_weight != 0)
__device__ __forceinline__ void out_range(int i, float scale, int n_out, int& lo, int& hi) {
    const float inv = 1.f / scale;
    lo = (int)floorf(((float)i - 2.f + 0.5f) * inv - 0.5f) - 1;
    hi = (int)ceilf(((float)i + 2.f + 0.5f) * inv - 0.5f) + 1;
    lo = lo < 0 ? 0 : lo;
    hi = hi > n_out - 1 ? n_out - 1 : hi;
}

__global__ void bicubic_bwd_kernel(const float* __restrict__ gy, long long planes, int H, int W, int Ho, int Wo,
                                   float sy, float sx, float* __restrict__ gx) {
    const long long total = planes * H * W;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int xq = (int)(idx % W);
        const long long r = idx / W;
        const int yq = (int)(r % H);
        const float* const gp = gy + (r / H) * (long long)Ho * Wo;
        int ylo, yhi, xlo, xhi;
        out_range(yq, sy, Ho, ylo, yhi);
        out_range(xq, sx, Wo, xlo, xhi);
        float acc = 0.f;
        for (int Y = ylo; Y <= yhi; ++Y) {
            const float wy = axis_weight(Y, sy, yq, H);
            if (wy == 0.f) continue;
            float rs = 0.f;
            for (int X = xlo; X <= xhi; ++X) {
                const float wx = axis_weight(X, sx, xq, W);
                if (wx != 0.f) rs += wx * gp[(long long)Y * Wo + X];
            }
            acc += wy * rs;
        }
        gx[idx] = acc;
    }
}

inline int nblk(long long n) {
    long long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" int bmc_bicubic_resize_fwd(const float* x, long long planes, int H, int W, int Ho, int Wo, float* y,
                                      bmc_stream_t s) {
    BMC_CHECK_ARG(x && y && planes > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "bmc_bicubic_resize_fwd: bad arguments");
    hipLaunchKernelGGL(bicubic_fwd_kernel, dim3(nblk(planes * Ho * Wo)), dim3(256), 0, (hipStream_t)s, x, planes, H, W, Ho,
                       Wo, (float)H / (float)Ho, (float)W / (float)Wo, y);
    BMC_CHECK_LAUNCH("bmc_bicubic_resize_fwd");
    return 0;
}

extern "C" int bmc_bicubic_resize_bwd(const float* gy, long long planes, int H, int W, int Ho, int Wo, float* gx,
                                      bmc_stream_t s) {
    BMC_CHECK_ARG(gy && gx && planes > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "bmc_bicubic_resize_bwd: bad arguments");
    hipLaunchKernelGGL(bicubic_bwd_kernel, dim3(nblk(planes * H * W)), dim3(256), 0, (hipStream_t)s, gy, planes, H, W, Ho, Wo,
                       (float)H / (float)Ho, (float)W / (float)Wo, gx);
    BMC_CHECK_LAUNCH("bmc_bicubic_resize_bwd");
    return 0;
}
