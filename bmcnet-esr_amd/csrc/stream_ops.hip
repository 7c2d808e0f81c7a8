// HBM-bound streaming kernels of the hot path: ReLU backward, bias-gradient
// column sums, LayerNorm2d fwd/bwd, 128x128 row softmax fwd/bwd, input packing,
// pixel (un)shuffle + bilinear head.  All NHWC fp32, 16-byte accesses, grid-stride.
#include "bmc_common.h"

namespace {

constexpr int MAXBLK = 2048;

inline int nblocks(long long work_items, int per_block) {
    long long b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    return (int)(b > MAXBLK ? MAXBLK : b);
}

// ------------------------------------------------------------------ relu bwd
__global__ void relu_bwd_kernel(const f32x4* __restrict__ dy, const f32x4* __restrict__ y, f32x4* __restrict__ g,
                                long long n4) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const f32x4 d = dy[i], v = y[i];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = v[k] > 0.f ? d[k] : 0.f;
        g[i] = o;
    }
}

// ------------------------------------------------------------------ sum over batch groups
// out[i] = sum_k in[k*n + i], fixed order: the gradient of an operand that several batch groups of a launch read
// (batch_mod < B), e.g. the shared part of conv_fs (models/BMCNet.py:70-73).
__global__ void group_sum_kernel(const f32x4* __restrict__ in, int groups, long long n4, f32x4* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        f32x4 s = in[i];
        for (int k = 1; k < groups; ++k) s += in[(long long)k * n4 + i];
        out[i] = s;
    }
}

// ------------------------------------------------------------------ column sums
// stage 1: each block sums a contiguous pixel range per channel into ws[block][C];
// stage 2: one block sums the partials (fixed order -> deterministic).
__global__ void colsum_stage1(const float* __restrict__ x, long long npix, int pix_stride, int C, float* __restrict__ ws) {
    // thread t handles channel t % C (C <= blockDim), pixel lane t / C
    const int c = threadIdx.x % C, pl = threadIdx.x / C, npl = blockDim.x / C;
    float s = 0.f;
    if (pl < npl)
        for (long long p = (long long)blockIdx.x * npl + pl; p < npix; p += (long long)gridDim.x * npl)
            s += x[p * pix_stride + c];
    __shared__ float red[1024];
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < C) {
        float t = 0.f;
        for (int k = 0; k < npl; ++k) t += red[k * C + threadIdx.x];
        ws[(long long)blockIdx.x * C + threadIdx.x] = t;
    }
}
// one block per channel: 256 threads stride over the per-block partials, fixed-order tree in LDS
__global__ void colsum_stage2(const float* __restrict__ ws, int nblk, int C, float* __restrict__ out, int accumulate) {
    const int c = blockIdx.x;
    float s = 0.f;
    for (int b = threadIdx.x; b < nblk; b += blockDim.x) s += ws[(long long)b * C + c];
    __shared__ float red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[c] = accumulate ? out[c] + red[0] : red[0];
}

// ------------------------------------------------------------------ LayerNorm2d
// LPP lanes per pixel (power of two <= 64, LPP * 4 >= C), each lane owns one float4 of the channel row; C is any multiple
// of 4 (lanes past the row are idle: C = 48 runs with LPP = 16, 12 of them active).
template <int LPP>
__global__ void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                              long long npix, int C, float eps, float* __restrict__ y, float* __restrict__ stats) {
    constexpr int PPW = 64 / LPP;
    const int lane = threadIdx.x & 63, sub = lane % LPP, pw = lane / LPP;
    const bool act = sub * 4 < C;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 gm = act ? *reinterpret_cast<const f32x4*>(gamma + sub * 4) : zero4;
    const f32x4 bt = act ? *reinterpret_cast<const f32x4*>(beta + sub * 4) : zero4;
    for (long long p0 = wave * PPW; p0 < npix; p0 += nwaves * PPW) {
        const long long p = p0 + pw;
        const bool ok = p < npix && act;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) v = *reinterpret_cast<const f32x4*>(x + p * C + sub * 4);
        float s = v[0] + v[1] + v[2] + v[3];
#pragma unroll
        for (int m = 1; m < LPP; m <<= 1) s += __shfl_xor(s, m);
        const float mu = s * (1.f / C);
        f32x4 d;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { d[k] = act ? v[k] - mu : 0.f; q += d[k] * d[k]; }
#pragma unroll
        for (int m = 1; m < LPP; m <<= 1) q += __shfl_xor(q, m);
        const float rstd = 1.f / sqrtf(q * (1.f / C) + eps);
        if (ok) {
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = d[k] * rstd * gm[k] + bt[k];
            *reinterpret_cast<f32x4*>(y + p * C + sub * 4) = o;
            if (sub == 0) { stats[2 * p] = mu; stats[2 * p + 1] = rstd; }
        }
    }
}

// backward: gx = rstd * (g - yhat*mean(g*yhat) - mean(g)), g = dy*gamma; per-block partial
// dgamma = sum dy*yhat, dbeta = sum dy go to ws[block][2][C].
template <int LPP>
__global__ void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ stats,
                              const float* __restrict__ gamma, long long npix, int C, float* __restrict__ dx,
                              float* __restrict__ ws) {
    constexpr int PPW = 64 / LPP;
    const int lane = threadIdx.x & 63, sub = lane % LPP, pw = lane / LPP;
    const bool act = sub * 4 < C;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 gm = act ? *reinterpret_cast<const f32x4*>(gamma + sub * 4) : zero4;
    f32x4 dg = {0.f, 0.f, 0.f, 0.f}, db = {0.f, 0.f, 0.f, 0.f};
    for (long long p0 = wave * PPW; p0 < npix; p0 += nwaves * PPW) {
        const long long p = p0 + pw;
        const bool ok = p < npix && act;
        f32x4 v = {0.f, 0.f, 0.f, 0.f}, d = {0.f, 0.f, 0.f, 0.f};
        float mu = 0.f, rstd = 0.f;
        if (ok) {
            v = *reinterpret_cast<const f32x4*>(x + p * C + sub * 4);
            d = *reinterpret_cast<const f32x4*>(dy + p * C + sub * 4);
            mu = stats[2 * p]; rstd = stats[2 * p + 1];
        }
        f32x4 yh, g;
        float sg = 0.f, sgy = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            yh[k] = ok ? (v[k] - mu) * rstd : 0.f;
            g[k] = d[k] * gm[k];
            sg += g[k];
            sgy += g[k] * yh[k];
            dg[k] += d[k] * yh[k];
            db[k] += d[k];
        }
#pragma unroll
        for (int m = 1; m < LPP; m <<= 1) { sg += __shfl_xor(sg, m); sgy += __shfl_xor(sgy, m); }
        const float mg = sg * (1.f / C), mgy = sgy * (1.f / C);
        if (ok) {
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = rstd * (g[k] - yh[k] * mgy - mg);
            *reinterpret_cast<f32x4*>(dx + p * C + sub * 4) = o;
        }
    }
    // block reduce dgamma/dbeta: lanes with equal `sub` across the block
    __shared__ float red[2][256 * 4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { red[0][threadIdx.x * 4 + k] = dg[k]; red[1][threadIdx.x * 4 + k] = db[k]; }
    __syncthreads();
    if (threadIdx.x < C) {
        const int c = threadIdx.x, s = c >> 2, k = c & 3;
        float a = 0.f, b = 0.f;
        for (int t = s; t < (int)blockDim.x; t += LPP) { a += red[0][t * 4 + k]; b += red[1][t * 4 + k]; }
        ws[((long long)blockIdx.x * 2 + 0) * C + c] = a;
        ws[((long long)blockIdx.x * 2 + 1) * C + c] = b;
    }
}
// grid = 2*C blocks (dgamma channels then dbeta channels), same fixed-order reduction as colsum_stage2
__global__ void ln_bwd_finish(const float* __restrict__ ws, int nblk, int C, float* dgamma, float* dbeta, int accumulate) {
    const int which = blockIdx.x / C, c = blockIdx.x % C;
    float s = 0.f;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) s += ws[((long long)i * 2 + which) * C + c];
    __shared__ float red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float* o = which ? dbeta : dgamma;
        o[c] = accumulate ? o[c] + red[0] : red[0];
    }
}

// ------------------------------------------------------------------ row softmax (one wave per row)
__global__ void softmax_fwd_kernel(const float* __restrict__ a, long long rows, int C, float* __restrict__ p) {
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long r = wave; r < rows; r += nwaves) {
        const float* ar = a + r * C;
        float mx = -INFINITY;
        for (int c = lane; c < C; c += 64) mx = fmaxf(mx, ar[c]);
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += expf(ar[c] - mx);
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m);
        const float inv = 1.f / s;
        for (int c = lane; c < C; c += 64) p[r * C + c] = expf(ar[c] - mx) * inv;
    }
}
__global__ void softmax_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp, long long rows, int C,
                                   float scale, float* __restrict__ da) {
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long r = wave; r < rows; r += nwaves) {
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += p[r * C + c] * dp[r * C + c];
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m);
        for (int c = lane; c < C; c += 64) da[r * C + c] = p[r * C + c] * (dp[r * C + c] - s) * scale;
    }
}

// ------------------------------------------------------------------ input packing
__global__ void pack_inputs_kernel(const float* __restrict__ x, long long sb, long long sc, long long st, long long sy,
                                   long long sx, int B, int H, int W, int repeat, float* __restrict__ xp,
                                   float* __restrict__ xn) {
    const long long npix = (long long)B * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix * 4; i += (long long)gridDim.x * blockDim.x) {
        const long long pidx = i >> 2;
        const int q = (int)(i & 3);
        const int xw = pidx % W;
        const long long r = pidx / W;
        const int yh = r % H, b = (int)(r / H);
        const float* base = x + b * sb + yh * sy + xw * sx;
        const float f1p = base[0], f2p = base[st], f1n = base[sc], f2n = base[sc + st];
        f32x4 vp, vn;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ch = q * 4 + k;
            vp[k] = ch < repeat ? f1p : (ch < 2 * repeat ? f2p : 0.f);
            vn[k] = ch < repeat ? f1n : (ch < 2 * repeat ? f2n : 0.f);
        }
        *reinterpret_cast<f32x4*>(xp + pidx * 16 + q * 4) = vp;
        *reinterpret_cast<f32x4*>(xn + pidx * 16 + q * 4) = vn;
    }
}

// ------------------------------------------------------------------ pixel (un)shuffle, NCHW HR <-> NHWC LR
// LR channel index = c*r*r + i*r + j  <->  HR (c, y*r+i, x*r+j)
// split = S > 1: the LR tensor is stored as S batch-stacked channel groups, [S*B][H][W][C*r*r/S] (group s of sample b at
// batch s*B + b) -- the layout the multi-source convolutions read o[:, :s^2] / o[:, s^2:] in (models/BMCNet.py:63).
__device__ __forceinline__ long long lr_index(int b, int y, int x, int ch, int B, int H, int W, int CC, int split) {
    const int cg = CC / split, s = ch / cg, cw = ch - s * cg;
    return ((((long long)s * B + b) * H + y) * W + x) * cg + cw;
}
__global__ void unshuffle_kernel(const float* __restrict__ hr, int B, int C, int H, int W, int r, float* __restrict__ lr,
                                 int split) {
    const int CC = C * r * r;
    const long long total = (long long)B * H * W * CC;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        // iterate in HR-contiguous order for coalesced reads: idx -> (b, c, Y, X)
        const int X = idx % (W * r);
        long long t = idx / (W * r);
        const int Y = t % (H * r); t /= (H * r);
        const int c = t % C;
        const int b = (int)(t / C);
        const int y = Y / r, i = Y - y * r, x = X / r, j = X - x * r;
        lr[lr_index(b, y, x, (c * r + i) * r + j, B, H, W, CC, split)] = hr[idx];
    }
}
__global__ void shuffle_kernel(const float* __restrict__ lr, int B, int C, int H, int W, int r, const float* __restrict__ base,
                               long long sb, long long sc, long long sy, long long sx, float* __restrict__ hr, int split) {
    const int CC = C * r * r;
    const long long total = (long long)B * H * W * CC;
    const float inv = 1.f / r;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int X = idx % (W * r);
        long long t = idx / (W * r);
        const int Y = t % (H * r); t /= (H * r);
        const int c = t % C;
        const int b = (int)(t / C);
        const int y = Y / r, i = Y - y * r, x = X / r, j = X - x * r;
        float v = lr[lr_index(b, y, x, (c * r + i) * r + j, B, H, W, CC, split)];
        if (base) {
            // F.interpolate(bilinear, align_corners=False): src = (dst+0.5)/r - 0.5 clamped at 0
            float fy = fmaxf((Y + 0.5f) * inv - 0.5f, 0.f), fx = fmaxf((X + 0.5f) * inv - 0.5f, 0.f);
            const int yy0 = (int)fy, xx0 = (int)fx;
            const int yy1 = yy0 + 1 < H ? yy0 + 1 : H - 1, xx1 = xx0 + 1 < W ? xx0 + 1 : W - 1;
            const float ly = fy - yy0, lx = fx - xx0;
            const float* bp = base + b * sb + c * sc;
            const float v00 = bp[yy0 * sy + xx0 * sx], v01 = bp[yy0 * sy + xx1 * sx];
            const float v10 = bp[yy1 * sy + xx0 * sx], v11 = bp[yy1 * sy + xx1 * sx];
            const float top = v00 * (1.f - lx) + v01 * lx, bot = v10 * (1.f - lx) + v11 * lx;
            v += top * (1.f - ly) + bot * ly;
        }
        hr[idx] = v;
    }
}

// ------------------------------------------------------------------ head + MSE (models/BMCNet.py:119 + train.py:233)
// forward: pred = pixel_shuffle(x_o) + bilinear(base) as shuffle_kernel, plus per-block partial sums of (pred - gt)^2
// (fixed-order tree in LDS; mse_finish sums the partials in index order -> deterministic loss);
// backward: d x_o = pixel_unshuffle(dpred + (2 dloss / numel) (pred - gt)) in one pass: the MSE gradient is never
// materialised and never added by a separate kernel.
__global__ void head_mse_fwd_kernel(const float* __restrict__ lr, int B, int C, int H, int W, int r, const float* __restrict__ base,
                                    long long sb, long long sc, long long sy, long long sx, const float* __restrict__ gt,
                                    long long gsb, float* __restrict__ hr, float* __restrict__ partials) {
    const int CC = C * r * r;
    const long long per_b = (long long)C * H * W * r * r, total = per_b * B;
    const float inv = 1.f / r;
    float acc = 0.f;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int X = idx % (W * r);
        long long t = idx / (W * r);
        const int Y = t % (H * r); t /= (H * r);
        const int c = t % C;
        const int b = (int)(t / C);
        const int y = Y / r, i = Y - y * r, x = X / r, j = X - x * r;
        float v = lr[(((long long)b * H + y) * W + x) * CC + (c * r + i) * r + j];
        float fy = fmaxf((Y + 0.5f) * inv - 0.5f, 0.f), fx = fmaxf((X + 0.5f) * inv - 0.5f, 0.f);
        const int yy0 = (int)fy, xx0 = (int)fx;
        const int yy1 = yy0 + 1 < H ? yy0 + 1 : H - 1, xx1 = xx0 + 1 < W ? xx0 + 1 : W - 1;
        const float ly = fy - yy0, lx = fx - xx0;
        const float* bp = base + b * sb + c * sc;
        const float v00 = bp[yy0 * sy + xx0 * sx], v01 = bp[yy0 * sy + xx1 * sx];
        const float v10 = bp[yy1 * sy + xx0 * sx], v11 = bp[yy1 * sy + xx1 * sx];
        const float top = v00 * (1.f - lx) + v01 * lx, bot = v10 * (1.f - lx) + v11 * lx;
        v += top * (1.f - ly) + bot * ly;
        hr[idx] = v;
        const float d = v - gt[(long long)b * gsb + (idx - (long long)b * per_b)];
        acc += d * d;
    }
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}
__global__ void mse_finish_kernel(const float* __restrict__ partials, int n, float scale, float* __restrict__ loss) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partials[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = red[0] * scale;
}
__global__ void head_mse_bwd_kernel(const float* __restrict__ dpred, const float* __restrict__ pred, const float* __restrict__ gt,
                                    long long gsb, const float* __restrict__ gloss, int B, int C, int H, int W, int r,
                                    float* __restrict__ dlr) {
    const int CC = C * r * r;
    const long long per_b = (long long)C * H * W * r * r, total = per_b * B;
    const float coef = gloss ? 2.f * gloss[0] / (float)total : 0.f;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int X = idx % (W * r);
        long long t = idx / (W * r);
        const int Y = t % (H * r); t /= (H * r);
        const int c = t % C;
        const int b = (int)(t / C);
        const int y = Y / r, i = Y - y * r, x = X / r, j = X - x * r;
        float g = dpred ? dpred[idx] : 0.f;
        if (gloss) g += coef * (pred[idx] - gt[(long long)b * gsb + (idx - (long long)b * per_b)]);
        dlr[(((long long)b * H + y) * W + x) * CC + (c * r + i) * r + j] = g;
    }
}

// ------------------------------------------------------------------ weight packing
__global__ void pack_weight_kernel(const float* __restrict__ w, const int* __restrict__ kmap, int G, int Cout, int Cin,
                                   int taps, int Kpad, int Coutpad, float* __restrict__ out) {
    // out[g][chunk][tap][co][16]
    const long long per_g = (long long)Kpad * taps * Coutpad;
    const long long total = per_g * G;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int kk = idx & 15;
        long long t = idx >> 4;
        const int co = t % Coutpad; t /= Coutpad;
        const int tap = t % taps; t /= taps;
        const int chunk = t % (Kpad / 16);
        const int g = (int)(t / (Kpad / 16));
        const int k = chunk * 16 + kk;
        const int ci = kmap ? kmap[k] : (k < Cin ? k : -1);
        float v = 0.f;
        if (ci >= 0 && co < Cout) v = w[(((long long)g * Cout + co) * Cin + ci) * taps + tap];
        out[idx] = v;
    }
}
__global__ void pack_weight_t_kernel(const float* __restrict__ w, const int* __restrict__ kmap, int G, int Cout, int Cin,
                                     int taps, int k0, int nk, int nkpad, int Coutpad16, float* __restrict__ out) {
    // data-gradient operator for packed input channels [k0, k0+nk):
    // out[g][chunk over co][tap'][n = k - k0 (padded to nkpad)][16 co] = w[g][co][kmap[k]][taps-1-tap']
    const long long per_g = (long long)Coutpad16 * taps * nkpad;
    const long long total = per_g * G;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int cc = idx & 15;
        long long t = idx >> 4;
        const int n = t % nkpad; t /= nkpad;
        const int tap = t % taps; t /= taps;
        const int chunk = t % (Coutpad16 / 16);
        const int g = (int)(t / (Coutpad16 / 16));
        const int co = chunk * 16 + cc;
        float v = 0.f;
        if (n < nk && co < Cout) {
            const int ci = kmap ? kmap[k0 + n] : k0 + n;
            if (ci >= 0) v = w[(((long long)g * Cout + co) * Cin + ci) * taps + (taps - 1 - tap)];
        }
        out[idx] = v;
    }
}

}  // namespace

extern "C" int bmc_relu_bwd(const float* dy, const float* y, float* g, long long n, bmc_stream_t s) {
    BMC_CHECK_ARG(n % 4 == 0, "bmc_relu_bwd: n must be a multiple of 4");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(nblocks(n / 4, 256 * 4)), dim3(256), 0, (hipStream_t)s,
                       (const f32x4*)dy, (const f32x4*)y, (f32x4*)g, n / 4);
    BMC_CHECK_LAUNCH("bmc_relu_bwd");
    return 0;
}

extern "C" int bmc_group_sum(const float* in, int groups, long long n, float* out, bmc_stream_t s) {
    BMC_CHECK_ARG(in && out && groups >= 1 && n % 4 == 0, "bmc_group_sum: n must be a multiple of 4");
    hipLaunchKernelGGL(group_sum_kernel, dim3(nblocks(n / 4, 256 * 4)), dim3(256), 0, (hipStream_t)s, (const f32x4*)in, groups,
                       n / 4, (f32x4*)out);
    BMC_CHECK_LAUNCH("bmc_group_sum");
    return 0;
}

extern "C" int bmc_colsum(const float* x, long long npix, int pix_stride, int C, float* ws, float* out, int accumulate,
                          bmc_stream_t s) {
    BMC_CHECK_ARG(C >= 1 && C <= 1024, "bmc_colsum: C=%d out of range", C);
    const int threads = C <= 256 ? 256 : 1024;
    const int npl = threads / C;
    const int nblk = nblocks(npix, npl * 16);
    hipLaunchKernelGGL(colsum_stage1, dim3(nblk), dim3(threads), 0, (hipStream_t)s, x, npix, pix_stride, C, ws);
    hipLaunchKernelGGL(colsum_stage2, dim3(C), dim3(256), 0, (hipStream_t)s, ws, nblk, C, out, accumulate);
    BMC_CHECK_LAUNCH("bmc_colsum");
    return 0;
}

// lanes per pixel: the power of two >= C/4 (C = 16 -> 4, 48 -> 16, 128 -> 32, ...)
static int ln_lpp(int C) {
    int l = 1;
    while (l * 4 < C) l <<= 1;
    return l;
}
#define LN_DISPATCH(KERNEL, ...)                                                              \
    if (C < 4 || C > 256 || C % 4) { bmc_set_error("layernorm: C=%d unsupported (multiples of 4 up to 256)", C); return -1; } \
    switch (ln_lpp(C)) {                                                                      \
        case 1: hipLaunchKernelGGL((KERNEL<1>), grid, dim3(256), 0, st, __VA_ARGS__); break;   \
        case 2: hipLaunchKernelGGL((KERNEL<2>), grid, dim3(256), 0, st, __VA_ARGS__); break;   \
        case 4: hipLaunchKernelGGL((KERNEL<4>), grid, dim3(256), 0, st, __VA_ARGS__); break;   \
        case 8: hipLaunchKernelGGL((KERNEL<8>), grid, dim3(256), 0, st, __VA_ARGS__); break;   \
        case 16: hipLaunchKernelGGL((KERNEL<16>), grid, dim3(256), 0, st, __VA_ARGS__); break; \
        case 32: hipLaunchKernelGGL((KERNEL<32>), grid, dim3(256), 0, st, __VA_ARGS__); break; \
        default: hipLaunchKernelGGL((KERNEL<64>), grid, dim3(256), 0, st, __VA_ARGS__); break; \
    }

extern "C" int bmc_layernorm_fwd(const float* x, const float* gamma, const float* beta, long long npix, int C, float eps,
                                 float* y, float* stats, bmc_stream_t s) {
    hipStream_t st = (hipStream_t)s;
    dim3 grid(nblocks(npix * ln_lpp(C), 256 * 4));
    LN_DISPATCH(ln_fwd_kernel, x, gamma, beta, npix, C, eps, y, stats);
    BMC_CHECK_LAUNCH("bmc_layernorm_fwd");
    return 0;
}

extern "C" int bmc_layernorm_bwd(const float* dy, const float* x, const float* stats, const float* gamma, long long npix,
                                 int C, float* dx, float* ws, float* dgamma, float* dbeta, int accumulate, bmc_stream_t s) {
    hipStream_t st = (hipStream_t)s;
    int nb = nblocks(npix * ln_lpp(C), 256 * 4);
    if (nb > 1024) nb = 1024;
    dim3 grid(nb);
    LN_DISPATCH(ln_bwd_kernel, dy, x, stats, gamma, npix, C, dx, ws);
    hipLaunchKernelGGL(ln_bwd_finish, dim3(2 * C), dim3(256), 0, st, ws, nb, C, dgamma, dbeta, accumulate);
    BMC_CHECK_LAUNCH("bmc_layernorm_bwd");
    return 0;
}

extern "C" int bmc_softmax_fwd(const float* a, long long rows, int C, float* p, bmc_stream_t s) {
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3(nblocks(rows, 4)), dim3(256), 0, (hipStream_t)s, a, rows, C, p);
    BMC_CHECK_LAUNCH("bmc_softmax_fwd");
    return 0;
}
extern "C" int bmc_softmax_bwd(const float* p, const float* dp, long long rows, int C, float scale_out, float* da,
                               bmc_stream_t s) {
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3(nblocks(rows, 4)), dim3(256), 0, (hipStream_t)s, p, dp, rows, C, scale_out, da);
    BMC_CHECK_LAUNCH("bmc_softmax_bwd");
    return 0;
}

extern "C" int bmc_pack_inputs(const float* x, long long sb, long long sc, long long st_, long long sy, long long sx, int B,
                               int H, int W, int repeat, float* xin_p, float* xin_n, bmc_stream_t s) {
    BMC_CHECK_ARG(repeat >= 1 && 2 * repeat <= 16, "bmc_pack_inputs: repeat=%d unsupported", repeat);
    const long long n = (long long)B * H * W * 4;
    hipLaunchKernelGGL(pack_inputs_kernel, dim3(nblocks(n, 256)), dim3(256), 0, (hipStream_t)s, x, sb, sc, st_, sy, sx, B, H,
                       W, repeat, xin_p, xin_n);
    BMC_CHECK_LAUNCH("bmc_pack_inputs");
    return 0;
}

extern "C" int bmc_unshuffle_to_nhwc(const float* hr, int B, int C, int H, int W, int r, float* lr, int split, bmc_stream_t s) {
    BMC_CHECK_ARG(split >= 1 && (C * r * r) % split == 0, "bmc_unshuffle_to_nhwc: split=%d must divide C*r*r", split);
    const long long n = (long long)B * C * H * W * r * r;
    hipLaunchKernelGGL(unshuffle_kernel, dim3(nblocks(n, 256 * 4)), dim3(256), 0, (hipStream_t)s, hr, B, C, H, W, r, lr, split);
    BMC_CHECK_LAUNCH("bmc_unshuffle_to_nhwc");
    return 0;
}
extern "C" int bmc_shuffle_to_hr(const float* lr, int B, int C, int H, int W, int r, const float* base, long long sb,
                                 long long sc, long long sy, long long sx, float* hr, int split, bmc_stream_t s) {
    BMC_CHECK_ARG(split >= 1 && (C * r * r) % split == 0, "bmc_shuffle_to_hr: split=%d must divide C*r*r", split);
    const long long n = (long long)B * C * H * W * r * r;
    hipLaunchKernelGGL(shuffle_kernel, dim3(nblocks(n, 256 * 4)), dim3(256), 0, (hipStream_t)s, lr, B, C, H, W, r, base, sb,
                       sc, sy, sx, hr, split);
    BMC_CHECK_LAUNCH("bmc_shuffle_to_hr");
    return 0;
}

extern "C" int bmc_head_mse_fwd(const float* lr, int B, int C, int H, int W, int r, const float* base, long long sb,
                                long long sc, long long sy, long long sx, const float* gt, long long gt_batch_stride, float* hr,
                                float* partials, float* loss, bmc_stream_t s) {
    BMC_CHECK_ARG(lr && base && gt && hr && partials && loss, "bmc_head_mse_fwd: null pointer");
    const long long n = (long long)B * C * H * W * r * r;
    const int nb = nblocks(n, 256 * 4);      // <= 2048 partials
    hipLaunchKernelGGL(head_mse_fwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)s, lr, B, C, H, W, r, base, sb, sc, sy, sx, gt,
                       gt_batch_stride, hr, partials);
    hipLaunchKernelGGL(mse_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, partials, nb, 1.f / (float)n, loss);
    BMC_CHECK_LAUNCH("bmc_head_mse_fwd");
    return 0;
}
extern "C" int bmc_head_mse_bwd(const float* dpred, const float* pred, const float* gt, long long gt_batch_stride,
                                const float* gloss, int B, int C, int H, int W, int r, float* dlr, bmc_stream_t s) {
    BMC_CHECK_ARG(dlr && (dpred || gloss) && (!gloss || (pred && gt)), "bmc_head_mse_bwd: bad arguments");
    const long long n = (long long)B * C * H * W * r * r;
    hipLaunchKernelGGL(head_mse_bwd_kernel, dim3(nblocks(n, 256 * 4)), dim3(256), 0, (hipStream_t)s, dpred, pred, gt,
                       gt_batch_stride, gloss, B, C, H, W, r, dlr);
    BMC_CHECK_LAUNCH("bmc_head_mse_bwd");
    return 0;
}

extern "C" int bmc_pack_weight(const float* w, const int* kmap, int G, int Cout, int Cin, int taps, int Kpad, int Coutpad,
                               float* out, bmc_stream_t s) {
    BMC_CHECK_ARG(Kpad % 16 == 0 && Coutpad % 32 == 0, "bmc_pack_weight: Kpad %% 16 / Coutpad %% 32");
    const long long n = (long long)G * Kpad * taps * Coutpad;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(nblocks(n, 256)), dim3(256), 0, (hipStream_t)s, w, kmap, G, Cout, Cin, taps,
                       Kpad, Coutpad, out);
    BMC_CHECK_LAUNCH("bmc_pack_weight");
    return 0;
}
extern "C" int bmc_pack_weight_t(const float* w, const int* kmap, int G, int Cout, int Cin, int taps, int k0, int nk,
                                 int nkpad, int Coutpad16, float* out, bmc_stream_t s) {
    BMC_CHECK_ARG(Coutpad16 % 16 == 0 && nkpad % 32 == 0, "bmc_pack_weight_t: Coutpad16 %% 16 / nkpad %% 32");
    const long long n = (long long)G * Coutpad16 * taps * nkpad;
    hipLaunchKernelGGL(pack_weight_t_kernel, dim3(nblocks(n, 256)), dim3(256), 0, (hipStream_t)s, w, kmap, G, Cout, Cin,
                       taps, k0, nk, nkpad, Coutpad16, out);
    BMC_CHECK_LAUNCH("bmc_pack_weight_t");
    return 0;
}

// ---- per-image pointer tables (bmc_src_t, BMC_SRC_TABLE)
namespace {
struct PtrTab { unsigned long long p[256]; };
__global__ void ptr_table_kernel(const PtrTab t, int n, unsigned long long* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = t.p[i];
}
}  // namespace

extern "C" int bmc_ptr_table(const unsigned long long* ptrs, int n, unsigned long long* table, bmc_stream_t s) {
    BMC_CHECK_ARG(ptrs && table && n >= 1 && n <= 256, "bmc_ptr_table: 1 .. 256 pointers");
    PtrTab t;
    for (int i = 0; i < n; ++i) t.p[i] = ptrs[i];
    for (int i = n; i < 256; ++i) t.p[i] = 0;
    hipLaunchKernelGGL(ptr_table_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, t, n, table);
    BMC_CHECK_LAUNCH("bmc_ptr_table");
    return 0;
}
