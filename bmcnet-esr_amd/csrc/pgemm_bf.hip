// Pixel-reduction GEMM (weight gradients, channel Gram matrices) on the bf16 matrix cores of gfx950:
// the operator of pgemm.hip with NP bf16 planes per fp32 operand (bf_split.h; NP = 1: bf16 operands,
// NP = 3: exact 3-way split, six plane products = fp32-equivalent).
//
// The reduction axis is the PIXEL axis while a v_mfma_f32_32x32x16_bf16 operand wants 8 consecutive k per lane, so
// the LDS images stay pixel-major ([px][channels] bf16, as the tiles arrive from NHWC memory) and the fragments are
// fetched with the transposing LDS read ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group, delivered
// channel-per-lane).  The 9 taps of a 3x3 weight gradient are 9 row offsets into the same halo image.
// Rows are padded by 64 B (A: 256 -> 320 B, X: 128 -> 192 B) so that the 4 pixel rows of a transposed read fall on 4
// different 64-byte bank slots.
//
// One 8-wave workgroup per CU; a pixel tile (4x16, or 64 flat pixels) goes global -> registers (in flight under the
// previous tile's MFMAs) -> split into planes -> LDS; two barriers per tile.
// Workgroup shapes (rows x columns x taps; wave = 32 x 32 x its taps):
//   3x3, 3 planes : 128 x 32 x 9, waves 4 (rows) x 2 (tap groups 0-4 / 5-8): 80 accumulator registers per wave -- with
//                   all 9 taps (144) the staging registers spill, and a spill reload is a vector-memory operation
//                   whose wait (vmcnt is in order) serialises the tile prefetch;
//   3x3, 1 plane  : 128 x 64 x 9, waves 4 x 2 (columns);
//   1x1           : 128 x 128,    waves 4 x 2 (columns), 2 column tiles per wave.
#include "pgemm_k.h"
#include "bf_split.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
struct Frag8 { s16x4 lo, hi; };

__device__ __forceinline__ s16x4 tr_read(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}

template <int TAPS, int NP>
__global__ __launch_bounds__(512, 1) void pgemm_bf_kernel(const PgemmK a) {
    constexpr int HWD = PT_W + 2, HHT = PT_H + 2;
    constexpr int NHALO = TAPS == 9 ? HWD * HHT : PT;                   // 108 halo pixels or 64 pixels
    constexpr bool TSPLIT = TAPS == 9 && NP == 3;                       // tap groups instead of column groups
    constexpr int XCH = TAPS == 9 ? (TSPLIT ? 32 : 64) : 128;           // columns per workgroup
    constexpr int NT = TAPS == 9 ? 1 : 2;                               // 32-column tiles per wave
    constexpr int NTAP = TSPLIT ? 5 : TAPS;                             // taps per wave (the second group has 4)
    constexpr int XQ = XCH / 4;                                         // float4 per X row
    constexpr int AST = 256 + 64, XST = XCH == 32 ? 64 : XCH * 2 + 64;  // LDS row strides in bytes (mod 256 = 64 or 192)
    constexpr int APL = PT * AST, XPL = NHALO * XST;                    // bytes per plane
    constexpr int NAL = PT * 32 / 512, NXL = (NHALO * XQ + 511) / 512;  // float4 per thread and tile
    __shared__ __attribute__((aligned(16))) unsigned char lds[NP * (APL + XPL)];
    unsigned char* const Al = lds;
    unsigned char* const Xl = lds + NP * APL;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31;
    const int mw = wave & 3, nw = TSPLIT ? 0 : wave >> 2, tw = TSPLIT ? wave >> 2 : 0;

    int bid = blockIdx.x;
    const int split = bid % a.nsplit; bid /= a.nsplit;
    const int nb = bid % a.n_nblk; bid /= a.n_nblk;
    const int mb = bid % a.n_mblk;
    const int g = bid / a.n_mblk;
    const int m0 = mb * 128, n0 = nb * XCH;
    const bool wave_active = m0 + 32 * mw < a.Mpad;
    const int HWp = a.H * a.W;

    f32x16 acc[NTAP * NT];
#pragma unroll
    for (int t = 0; t < NTAP * NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // ---- tile loads: global -> registers (zero buffer for pixels outside the image / channels beyond M, N)
    const int ntiles = a.batch_per_group * a.tiles_per_img;
    // 1x1: a 64-pixel tile's MFMAs are shorter than the HBM latency -> two tiles of loads in flight (register ring with
    // static slots: the tile loop is unrolled by PD)
    constexpr int PD = TAPS == 1 ? 2 : 1;
    f32x4 ar[PD][NAL], xr[PD][NXL];
    // `vt` = tid behind an opaque barrier: keeps the compiler from hoisting the per-thread index arithmetic of the 8
    // loads / stores out of the tile loop into dozens of long-lived registers (the accumulators need them)
    constexpr bool NOHOIST = TAPS == 9 && !TSPLIT;     // 144 accumulator registers: no room for hoisted index math
    auto load_tile = [&](int tile, int slot) {
        int vt = tid;
        if (NOHOIST) asm volatile("" : "+v"(vt));
        const int bb = tile / a.tiles_per_img, tin = tile - bb * a.tiles_per_img;
        const int b = g * a.batch_per_group + bb;
        int y0 = 0, x0 = 0, p0 = 0;
        if (TAPS == 9) { y0 = (tin / a.tiles_x) * PT_H; x0 = (tin % a.tiles_x) * PT_W; }
        else p0 = tin * PT;
        const float* ab = src_batch_ptr(a.a, b);
#pragma unroll
        for (int i = 0; i < NAL; ++i) {
            const int e = i * 512 + vt, p = e >> 5, c4 = (e & 31) * 4;
            long long pix;
            bool ok;
            if (TAPS == 9) {
                const int y = y0 + (p >> 4), x = x0 + (p & 15);
                ok = y < a.H && x < a.W;
                pix = (long long)y * a.W + x;
            } else {
                pix = p0 + p;
                ok = pix < HWp;
            }
            ok = ok && m0 + c4 < a.M;
            const float* src = ok ? ab + pix * a.a.pix_stride + m0 + c4 : a.zeros;
            ar[slot][i] = *reinterpret_cast<const f32x4*>(src);
        }
#pragma unroll
        for (int i = 0; i < NXL; ++i) {
            const int e = i * 512 + vt, hp = e / XQ, c4 = (e % XQ) * 4;
            long long pix;
            bool ok;
            if (TAPS == 9) {
                const int hy = hp / HWD, hx = hp - hy * HWD;
                const int y = y0 - 1 + hy, x = x0 - 1 + hx;
                ok = hp < NHALO && y >= 0 && y < a.H && x >= 0 && x < a.W;
                pix = (long long)y * a.W + x;
            } else {
                pix = p0 + hp;
                ok = pix < HWp;
            }
            int ch = n0 + c4;
            ok = ok && ch < a.N;
            SrcDev S = a.src[0];
#pragma unroll
            for (int si = 1; si < BMC_MAX_SRC; ++si)
                if (ch >= S.nch && si < a.nsrc) { ch -= S.nch; S = a.src[si]; }
            const float* src = ok ? src_batch_ptr(S, b) + pix * S.pix_stride + ch : a.zeros;
            xr[slot][i] = *reinterpret_cast<const f32x4*>(src);
        }
    };
    // bias gradient = column sums of A: a thread always holds the same 4 channels ((tid & 31) * 4), so it adds up its
    // own fp32 registers; 16 threads per channel quad are folded to 4 partial rows per workgroup at the end
    const bool do_bias = a.bias_slabs != nullptr && nb == 0;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    auto store_tile = [&](int slot) {
        int vt = tid;
        if (NOHOIST) asm volatile("" : "+v"(vt));
#pragma unroll
        for (int i = 0; i < NAL; ++i) {
            const int e = i * 512 + vt, p = e >> 5, c4 = (e & 31) * 4;
            if (do_bias) bsum += ar[slot][i];
            u32x2 pl[NP];
            split4<NP>(ar[slot][i], pl);
#pragma unroll
            for (int q = 0; q < NP; ++q) *reinterpret_cast<u32x2*>(Al + q * APL + p * AST + c4 * 2) = pl[q];
        }
#pragma unroll
        for (int i = 0; i < NXL; ++i) {
            const int e = i * 512 + vt, hp = e / XQ, c4 = (e % XQ) * 4;
            if ((i + 1) * 512 <= NHALO * XQ || hp < NHALO) {
                u32x2 pl[NP];
                split4<NP>(xr[slot][i], pl);
#pragma unroll
                for (int q = 0; q < NP; ++q) *reinterpret_cast<u32x2*>(Xl + q * XPL + hp * XST + c4 * 2) = pl[q];
            }
        }
    };

    // ---- transposed fragment reads: lane l of a 16-lane group supplies the address of pixel row (l & 15) >> 2,
    // channels 4 (l & 3) .. + 3 of the group's 16 channels and receives channel (l & 15), 4 pixels
    const int th = lane >> 5, tg = (lane >> 4) & 1, tq = (lane & 15) >> 2, tp = lane & 3;
    const unsigned char* const a_lane = Al + (8 * th + tq) * AST + (32 * mw + 16 * tg + 4 * tp) * 2;
    const unsigned char* const x_lane = Xl + (8 * th + tq) * XST + (32 * NT * nw + 16 * tg + 4 * tp) * 2;
    auto frag = [&](const unsigned char* p, int stride) {   // 8 consecutive pixels (k = 8 h .. 8 h + 7) of this lane's channel
        Frag8 f;
        f.lo = tr_read(p);
        f.hi = tr_read(p + 4 * stride);
        return __builtin_bit_cast(bf16x8, f);
    };

    // taps of this wave: tap_lo .. tap_lo + ntap - 1 (wave-uniform; one code path and ONE accumulator set for both tap
    // groups -- separate instantiations per group make the register allocator keep two accumulator sets)
    const int tap_lo = TSPLIT ? 5 * tw : 0, ntap = TSPLIT ? (tw ? 4 : 5) : TAPS;
    auto compute = [&]() {
        int toff[NTAP];   // byte offset of the tap's first halo row
#pragma unroll
        for (int ti = 0; ti < NTAP; ++ti) {
            const int tap = tap_lo + ti;
            toff[ti] = TAPS == 9 ? ((tap / 3) * HWD + tap % 3) * XST : 0;
        }
#pragma unroll 1
        for (int kg = 0; kg < PT / 16; ++kg) {   // 16 pixels (one tile row) per MFMA k-step; not unrolled (registers)
            bf16x8 af[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) af[p] = frag(a_lane + p * APL + kg * 16 * AST, AST);
#pragma unroll
            for (int ti = 0; ti < NTAP; ++ti) {
                if (TSPLIT && ti >= ntap) continue;
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    const unsigned char* const xrow = x_lane + (TAPS == 9 ? kg * HWD * XST : kg * 16 * XST) + toff[ti] + 64 * u;
                    bf16x8 xf[NP];
#pragma unroll
                    for (int p = 0; p < NP; ++p) xf[p] = frag(xrow + p * XPL, XST);
                    f32x16& c = acc[ti * NT + u];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], xf[0], c, 0, 0, 0);
                    if constexpr (NP == 3) {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], xf[1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], xf[0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], xf[1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], xf[2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], xf[0], c, 0, 0, 0);
                    }
                }
            }
        }
    };

#pragma unroll
    for (int j = 0; j < PD; ++j)
        if (split + j * a.nsplit < ntiles) load_tile(split + j * a.nsplit, j);
    for (int base = split; base < ntiles; base += PD * a.nsplit) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
            const int tile = base + d * a.nsplit;
            if (tile < ntiles) {
                __syncthreads();            // the previous tile's fragment reads are done
                store_tile(d);
                __syncthreads();
                if (tile + PD * a.nsplit < ntiles) load_tile(tile + PD * a.nsplit, d);
                if (wave_active) compute();
            }
        }
    }

    if (do_bias) {   // 16 per-thread partials per channel quad -> 4 partial rows (layout of pgemm.hip)
        __syncthreads();
        float* const red = reinterpret_cast<float*>(lds);
        *reinterpret_cast<f32x4*>(red + (tid >> 5) * 128 + (tid & 31) * 4) = bsum;
        __syncthreads();
        const int part = tid >> 7, ch = tid & 127;
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) s += red[(part * 4 + r) * 128 + ch];
        if (m0 + ch < a.Mpad) a.bias_slabs[(((long long)split * a.G + g) * 4 + part) * a.Mpad + m0 + ch] = s;
    }
    if (wave_active) {
        float* const sl = a.slabs + (((long long)split * a.G + g) * TAPS) * a.Mpad * a.Npad;
        const int lh = lane >> 5;
#pragma unroll
        for (int ti = 0; ti < NTAP; ++ti)
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const int n = n0 + 32 * NT * nw + 32 * u + li;
                if (n < a.Npad && ti < ntap) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + 32 * mw + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        sl[((long long)(tap_lo + ti) * a.Mpad + m) * a.Npad + n] = acc[ti * NT + u][r];
                    }
                }
            }
    }
}

}  // namespace

int bmc_pgemm_cols(int taps, int math) {   // columns of C per workgroup (the host sizes n_nblk / nsplit with it)
    if (taps != 9) return 128;
    if (math == BMC_MATH_BF16X6) return 32;
    return 64;
}

int bmc_pgemm_bf_launch(const PgemmK& k, int taps, int planes, hipStream_t st) {
    dim3 grid((unsigned)((long long)k.G * k.n_mblk * k.n_nblk * k.nsplit)), block(512);
    if (taps == 9) {
        if (planes == 3) hipLaunchKernelGGL((pgemm_bf_kernel<9, 3>), grid, block, 0, st, k);
        else hipLaunchKernelGGL((pgemm_bf_kernel<9, 1>), grid, block, 0, st, k);
    } else {
        if (planes == 3) hipLaunchKernelGGL((pgemm_bf_kernel<1, 3>), grid, block, 0, st, k);
        else hipLaunchKernelGGL((pgemm_bf_kernel<1, 1>), grid, block, 0, st, k);
    }
    BMC_CHECK_LAUNCH("bmc_pgemm (bf16 planes)");
    return 0;
}
