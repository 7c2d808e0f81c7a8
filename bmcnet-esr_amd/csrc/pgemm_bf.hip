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
#include <type_traits>

#ifdef BMC_BF_STAMP
// experiment builds only (tools/): per-workgroup cycle totals of the producer / consumer phases
__device__ unsigned long long g_stamp[8 * 1024];
extern "C" int bmc_stamp_read(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
struct Frag8 { s16x4 lo, hi; };

__device__ __forceinline__ s16x4 tr_read(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}

template <int TAPS, int NP, bool TAB = false>
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
    // one tap ROW per workgroup (a.tap_groups = 3; 3x3, one plane): small images, a third of the pixel splits (pgemm.hip)
    const bool wg_taps = TAPS == 9 && !TSPLIT && a.tap_groups == 3;
    const int tgw = wg_taps ? bid % 3 : 0;
    if (wg_taps) bid /= 3;
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
    // Fast path (as in pgemm.hip): interior tiles whose columns lie in ONE source load with a uniform base + a per-item
    // 32-bit byte offset computed once per kernel -- instructions issued beside the MFMAs cost matrix-pipe time ~1:1.
    SrcDev xs = a.src[0];
    int xch0 = n0;
#pragma unroll
    for (int si = 1; si < BMC_MAX_SRC; ++si)
        if (xch0 >= xs.nch && si < a.nsrc) { xch0 -= xs.nch; xs = a.src[si]; }
    const bool all_ch = m0 + 128 <= a.M && n0 + XCH <= a.N && xch0 + XCH <= xs.nch &&
                        (long long)a.H * a.W * (a.a.pix_stride > xs.pix_stride ? a.a.pix_stride : xs.pix_stride) < (1ll << 28);
    unsigned fa_off[NAL], fx_off[NXL];
#pragma unroll
    for (int i = 0; i < NAL; ++i) {
        const int e = i * 512 + tid, p = e >> 5, c4 = (e & 31) * 4;
        fa_off[i] = (unsigned)(((TAPS == 9 ? (p >> 4) * a.W + (p & 15) : p) * a.a.pix_stride + m0 + c4) * 4);
    }
#pragma unroll
    for (int i = 0; i < NXL; ++i) {
        const int e = i * 512 + tid, hp = e / XQ, c4 = (e % XQ) * 4;
        const int hy = TAPS == 9 ? hp / HWD : 0, hx = TAPS == 9 ? hp - hy * HWD : hp;
        fx_off[i] = (unsigned)(((TAPS == 9 ? (hp < NHALO ? hy * a.W + hx : 0) : hp) * xs.pix_stride + xch0 + c4) * 4);
    }
    auto uniform_ptr = [](const float* p) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
    };
    auto load_tile = [&](int tile, int slot) {
        int vt = tid;
        if (NOHOIST) asm volatile("" : "+v"(vt));
        const int bb = tile / a.tiles_per_img, tin = tile - bb * a.tiles_per_img;
        const int b = g * a.batch_per_group + bb;
        int y0 = 0, x0 = 0, p0 = 0;
        if (TAPS == 9) { y0 = (tin / a.tiles_x) * PT_H; x0 = (tin % a.tiles_x) * PT_W; }
        else p0 = tin * PT;
        const float* ab = src_bp<TAB>(a.a, b);
        const bool interior = TAPS == 9 ? (y0 >= 1 && x0 >= 1 && y0 + PT_H + 1 <= a.H && x0 + PT_W + 1 <= a.W) : (p0 + PT <= HWp);
        if (interior && all_ch) {
            const char* const abt = uniform_ptr(ab + (TAPS == 9 ? (long long)y0 * a.W + x0 : (long long)p0) * a.a.pix_stride);
            const char* const xbt = uniform_ptr(src_bp<TAB>(xs, b) +
                                                (TAPS == 9 ? (long long)(y0 - 1) * a.W + (x0 - 1) : (long long)p0) * xs.pix_stride);
#pragma unroll
            for (int i = 0; i < NAL; ++i) ar[slot][i] = ldg16(abt + fa_off[i]);
#pragma unroll
            for (int i = 0; i < NXL; ++i) xr[slot][i] = ldg16(xbt + fx_off[i]);
            return;
        }
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
            ar[slot][i] = ldg16(src);
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
            const float* src = ok ? src_bp<TAB>(S, b) + pix * S.pix_stride + ch : a.zeros;
            xr[slot][i] = ldg16(src);
        }
    };
    // bias gradient = column sums of A: a thread always holds the same 4 channels ((tid & 31) * 4), so it adds up its
    // own fp32 registers; 16 threads per channel quad are folded to 4 partial rows per workgroup at the end
    const bool do_bias = a.bias_slabs != nullptr && nb == 0 && tgw == 0;
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
    const int tap_lo = TSPLIT ? 5 * tw : 3 * tgw, ntap = TSPLIT ? (tw ? 4 : 5) : (wg_taps ? 3 : TAPS);
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
                if ((TSPLIT || TAPS == 9) && ti >= ntap) continue;
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


// 3x3 weight gradient with 3 planes (the hot case), producer / consumer waves.  In pgemm_bf_kernel the split + store of
// a tile is a VALU-only phase that all 8 waves run at the same time, between two barriers, with the matrix pipes idle
// (29 % of the kernel: 0.58 ms with it, 0.41 ms with the MFMA loop alone).  Here the workgroup has 12 waves:
//   waves 0-7 (consumers): nothing but fragment reads + MFMAs on LDS image i & 1 (128 rows x 32 columns x 9 taps,
//                          4 row groups x 2 tap groups as above);
//   waves 8-11 (producers): tile i+1 from their registers -> split -> LDS image (i+1) & 1, then the global loads of
//                          tile i+2 into the same registers; the bias column sums come out of their fp32 registers.
// One barrier per tile; the producers' VALU / LDS-write / VMEM work runs beside the consumers' MFMAs.
// LDS: A image [64 px][128 ch] bf16 in 256-byte rows, 64-byte chunks XOR-swizzled by (row & 3) (no padding, so that two
// images fit); X image [108 halo px][32 ch] in 64-byte rows.
template <bool TAB>
__global__ __launch_bounds__(768, 1) void pgemm_bf9x3_kernel(const PgemmK a) {
    constexpr int NP = 3, HWD = PT_W + 2, HHT = PT_H + 2, NHALO = HWD * HHT, XCH = 32, XQ = XCH / 4;
    constexpr int AST = 256, XST = 64, APL = PT * AST, XPL = NHALO * XST, STAGE = NP * (APL + XPL);
    constexpr int NPT = 256;                                     // producer threads (4 waves)
    constexpr int NAL = PT * 32 / NPT, NXL = (NHALO * XQ + NPT - 1) / NPT;   // 8 + 4 float4 per producer thread and tile
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 8;

    int bid = blockIdx.x;
    const int split = bid % a.nsplit; bid /= a.nsplit;
    const int nb = bid % a.n_nblk; bid /= a.n_nblk;
    const int mb = bid % a.n_mblk;
    const int g = bid / a.n_mblk;
    const int m0 = mb * 128, n0 = nb * XCH;
    const int ntiles = a.batch_per_group * a.tiles_per_img;
    const int nmine = split < ntiles ? (ntiles - split + a.nsplit - 1) / a.nsplit : 0;

    if (producer) {
        const int pt = tid - 512;          // 0 .. 255
        f32x4 ar[1][NAL], xr[1][NXL];
        const bool do_bias = a.bias_slabs != nullptr && nb == 0;
        f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
        // VALU beside a saturated matrix pipe is scarce (~6 issue slots per 32-cycle MFMA on the SIMD the producer shares
        // with two consumers), so everything that does not depend on the tile is computed ONCE per item here: element
        // offset from the tile's first (halo) pixel, LDS byte offset, pixel coordinates inside the tile, source index.
        // Per tile and item that leaves: a bounds test only on border tiles, one pointer select, the load.
        int a_off[NAL], a_lds[NAL], a_yx[NAL];
        bool a_cok[NAL];
#pragma unroll
        for (int i2 = 0; i2 < NAL; ++i2) {
            const int e = i2 * NPT + pt, p = e >> 5, c4 = (e & 31) * 4;
            a_yx[i2] = ((p >> 4) << 16) | (p & 15);
            a_cok[i2] = m0 + c4 < a.M;
            a_off[i2] = (((p >> 4) * a.W + (p & 15)) * a.a.pix_stride + m0 + c4) * 4;     // bytes
            a_lds[i2] = p * AST + (((c4 >> 5) ^ (p & 3)) << 6) + (c4 & 31) * 2;
        }
        int x_off[NXL], x_lds[NXL], x_yx[NXL], x_si[NXL];
        bool x_cok[NXL];
#pragma unroll
        for (int i2 = 0; i2 < NXL; ++i2) {
            const int e = i2 * NPT + pt, hp = e / XQ, c4 = (e % XQ) * 4;
            const int hy = hp / HWD, hx = hp - hy * HWD;
            int ch = n0 + c4, si_sel = 0;
            x_cok[i2] = hp < NHALO && ch < a.N;
            SrcDev S = a.src[0];
#pragma unroll
            for (int si = 1; si < BMC_MAX_SRC; ++si)
                if (ch >= S.nch && si < a.nsrc) { ch -= S.nch; S = a.src[si]; si_sel = si; }
            x_si[i2] = si_sel;
            x_yx[i2] = (hy << 16) | hx;
            // bytes from the halo's first pixel (y0-1, x0-1); lanes of the last, partial item beyond the halo re-read that
            // pixel (their data is never stored): one halo row further down can be past the end of the tensor
            x_off[i2] = ((hp < NHALO ? hy * a.W + hx : 0) * S.pix_stride + ch) * 4;
            x_lds[i2] = NP * APL + hp * XST + c4 * 2;
        }
        // every channel of this workgroup's blocks exists and its 32 columns lie in the first source: no per-lane select
        const bool all_ch = m0 + 128 <= a.M && n0 + XCH <= a.N && n0 + XCH <= a.src[0].nch &&
                            (long long)a.H * a.W * (a.a.pix_stride > a.src[0].pix_stride ? a.a.pix_stride : a.src[0].pix_stride) < (1ll << 28);   // 32-bit byte offsets
        constexpr int NIT = NAL + NXL;
        struct TileP { const float* ab; const float* xb[BMC_MAX_SRC]; int y0, x0; bool live, fast; };
        // Tiles are visited in order split, split + nsplit, ...: the (image, tile row, tile column) decode is advanced
        // incrementally -- three integer divisions per tile are ~100 VALU instructions the producer cannot afford.
        const int step_img = a.nsplit / a.tiles_per_img, step_rem = a.nsplit - step_img * a.tiles_per_img;
        const int step_ty = step_rem / a.tiles_x, step_tx = step_rem - step_ty * a.tiles_x;
        int nx_bb = split / a.tiles_per_img, nx_ty, nx_tx;
        {
            const int tin = split - nx_bb * a.tiles_per_img;
            nx_ty = tin / a.tiles_x; nx_tx = tin - nx_ty * a.tiles_x;
        }
        auto tile_setup = [&](int i) {        // uniform per-tile values; must be called with i = 0, 1, 2, ... in order
            TileP t;                          // i >= nmine: a dead tile (every lane reads zeros)
            t.live = i < nmine;
            const int b = g * a.batch_per_group + (t.live ? nx_bb : 0);
            t.y0 = (t.live ? nx_ty : 0) * PT_H; t.x0 = (t.live ? nx_tx : 0) * PT_W;
            // advance to tile i + 1
            nx_bb += step_img; nx_ty += step_ty; nx_tx += step_tx;
            if (nx_tx >= a.tiles_x) { nx_tx -= a.tiles_x; ++nx_ty; }
            if (nx_ty >= a.tiles_y) { nx_ty -= a.tiles_y; ++nx_bb; }
            const bool interior = t.y0 >= 1 && t.x0 >= 1 && t.y0 + PT_H + 1 <= a.H && t.x0 + PT_W + 1 <= a.W;   // halo inside the image
            t.fast = t.live && interior && all_ch;
            t.ab = src_bp_uni<TAB>(a.a, b) + ((long long)t.y0 * a.W + t.x0) * a.a.pix_stride;
#pragma unroll
            for (int si = 0; si < BMC_MAX_SRC; ++si)
                t.xb[si] = src_bp_uni<TAB>(a.src[si], b) + ((long long)(t.y0 - 1) * a.W + (t.x0 - 1)) * a.src[si].pix_stride;
            return t;
        };
        // Loads are inline asm, waited for by hand (the compiler cannot count vmcnt across this loop).  Hazard of the
        // idiom: the compiler believes the register is written AT the asm; were it ever to copy the value elsewhere before
        // the wait (it does not: the `+v` pin after the wait keeps def and use in one register), the copy would be stale --
        // the parity tests would show garbage, not a small error.  The common case
        // costs no VALU: uniform base (SGPR pair) + the item's 32-bit byte offset.  (s_nop 4: the scalar base may have been produced
        // by a VALU readfirstlane; 5 wait states are required before VMEM uses it and inline asm is opaque to the hazard pass.)
        auto load_item = [&](const TileP& t, int it) {
            if (it < NAL) {
                const int i2 = it;
                if (t.fast) {
                    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(ar[0][i2]) : "v"(a_off[i2]), "s"(t.ab) : "memory");
                } else {
                    const bool ok = t.live && a_cok[i2] && t.y0 + (a_yx[i2] >> 16) < a.H && t.x0 + (a_yx[i2] & 0xffff) < a.W;
                    const float* src = ok ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(t.ab) + a_off[i2]) : a.zeros;
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ar[0][i2]) : "v"(src) : "memory");
                }
            } else {
                const int i2 = it - NAL;
                if (t.fast) {
                    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(xr[0][i2]) : "v"(x_off[i2]), "s"(t.xb[0]) : "memory");
                } else {
                    const int y = t.y0 - 1 + (x_yx[i2] >> 16), x = t.x0 - 1 + (x_yx[i2] & 0xffff);
                    const bool ok = t.live && x_cok[i2] && y >= 0 && y < a.H && x >= 0 && x < a.W;
                    const float* bp = t.xb[0];
                    if (a.nsrc > 1) {
#pragma unroll
                        for (int si = 1; si < BMC_MAX_SRC; ++si)
                            if (x_si[i2] == si) bp = t.xb[si];
                    }
                    const float* src = ok ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(bp) + x_off[i2]) : a.zeros;
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xr[0][i2]) : "v"(src) : "memory");
                }
            }
        };
        auto store_item = [&](int it, int buf) {
            unsigned char* const img = lds + buf * STAGE;
            if (it < NAL) {
                const int i2 = it;
                asm volatile("" : "+v"(ar[0][i2]));          // nothing that reads the item may move above its wait
                if (do_bias) bsum += ar[0][i2];
                u32x2 pl[NP];
                split4<NP>(ar[0][i2], pl);
#pragma unroll
                for (int q = 0; q < NP; ++q) *reinterpret_cast<u32x2*>(img + a_lds[i2] + q * APL) = pl[q];
            } else {
                const int i2 = it - NAL;
                asm volatile("" : "+v"(xr[0][i2]));
                if ((i2 + 1) * NPT <= NHALO * XQ || i2 * NPT + pt < NHALO * XQ) {
                    u32x2 pl[NP];
                    split4<NP>(xr[0][i2], pl);
#pragma unroll
                    for (int q = 0; q < NP; ++q) *reinterpret_cast<u32x2*>(img + x_lds[i2] + q * XPL) = pl[q];
                }
            }
        };
        auto wait_n = [&](auto n) {   // all but the newest n vector-memory operations of this wave are done
            constexpr int N = decltype(n)::value;
            __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
        };
        // Rotating pipeline, one item at a time: wait for the OLDEST outstanding load (item `it` of tile i+1, issued one
        // whole iteration ago), split + store it into image (i+1) & 1, and reissue the same registers for tile i+2: every
        // load has a full iteration to land, with one tile's worth of staging registers.
        {
            const TileP t0 = tile_setup(0);
#pragma unroll
            for (int it = 0; it < NIT; ++it) load_item(t0, it);
            wait_n(std::integral_constant<int, 0>{});
#pragma unroll
            for (int it = 0; it < NIT; ++it) store_item(it, 0);
            const TileP t1 = tile_setup(1);
#pragma unroll
            for (int it = 0; it < NIT; ++it) load_item(t1, it);
        }
        __syncthreads();
#ifdef BMC_BF_STAMP
        unsigned long long p_work = 0, p_wait = 0, p_store = 0, p_load = 0, p_t0 = __builtin_amdgcn_s_memtime();
#endif
        for (int i = 0; i < nmine; ++i) {
#ifdef BMC_BF_STAMP
            const unsigned long long w0 = __builtin_amdgcn_s_memtime();
#endif
            if (i + 1 < nmine) {
                const TileP t2 = tile_setup(i + 2);
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    wait_n(std::integral_constant<int, NIT - 1>{});
                    store_item(it, (i + 1) & 1);             // image (i+1)&1 was last read for tile i-1: barrier passed
                    load_item(t2, it);
                }
            }
#ifdef BMC_BF_STAMP
            p_work += __builtin_amdgcn_s_memtime() - w0;
#endif
            __syncthreads();
        }
#ifdef BMC_BF_STAMP
        if (pt == 0 && blockIdx.x < 1024) { g_stamp[blockIdx.x * 8 + 0] = p_work; g_stamp[blockIdx.x * 8 + 1] = __builtin_amdgcn_s_memtime() - p_t0; g_stamp[blockIdx.x * 8 + 2] = nmine; g_stamp[blockIdx.x * 8 + 3] = p_wait; g_stamp[1024 * 8 - 2048 + blockIdx.x * 2] = p_store; g_stamp[1024 * 8 - 2048 + blockIdx.x * 2 + 1] = p_load; }
#endif
        // bias: a producer thread always holds channel quad pt & 31; 8 threads per quad are folded to the 4 partial rows
        float* const red = reinterpret_cast<float*>(lds);
        if (do_bias) *reinterpret_cast<f32x4*>(red + (pt >> 5) * 128 + (pt & 31) * 4) = bsum;
        __syncthreads();            // (the consumers take part in this barrier too)
        if (do_bias) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int o = pt + 256 * j, part = o >> 7, ch = o & 127;
                if (m0 + ch < a.Mpad)
                    a.bias_slabs[(((long long)split * a.G + g) * 4 + part) * a.Mpad + m0 + ch] = red[part * 128 + ch] + red[(part + 4) * 128 + ch];
            }
        }
        return;
    }

    // ---- consumers
    const int li = lane & 31;
    const int mw = wave & 3, tw = wave >> 2;
    const bool wave_active = m0 + 32 * mw < a.Mpad;
    constexpr int NTAP = 5;
    f32x16 acc[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int th = lane >> 5, tg = (lane >> 4) & 1, tq = (lane & 15) >> 2, tp = lane & 3;
    const int a_off = (8 * th + tq) * AST + ((mw ^ tq) << 6) + (16 * tg + 4 * tp) * 2;
    const int x_off = NP * APL + (8 * th + tq) * XST + (16 * tg + 4 * tp) * 2;
    const int tap_lo = 5 * tw, ntap = tw ? 4 : 5;
    int toff[NTAP];
#pragma unroll
    for (int ti = 0; ti < NTAP; ++ti) {
        const int tap = tap_lo + ti;
        toff[ti] = ((tap / 3) * HWD + tap % 3) * XST;
    }
    auto frag = [&](const unsigned char* p, int stride) {
        Frag8 f;
        f.lo = tr_read(p);
        f.hi = tr_read(p + 4 * stride);
        return __builtin_bit_cast(bf16x8, f);
    };
    __syncthreads();
#ifdef BMC_BF_STAMP
    unsigned long long c_work = 0, c_t0 = __builtin_amdgcn_s_memtime();
#endif
    for (int i = 0; i < nmine; ++i) {
#ifdef BMC_BF_STAMP
        const unsigned long long w0 = __builtin_amdgcn_s_memtime();
#endif
        if (wave_active) {
            const unsigned char* const img = lds + (i & 1) * STAGE;
#pragma unroll 1
            for (int kg = 0; kg < PT / 16; ++kg) {
                bf16x8 af[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) af[p] = frag(img + a_off + p * APL + kg * 16 * AST, AST);
#pragma unroll
                for (int ti = 0; ti < NTAP; ++ti) {
                    if (ti >= ntap) continue;
                    const unsigned char* const xrow = img + x_off + kg * HWD * XST + toff[ti];
                    bf16x8 xf[NP];
#pragma unroll
                    for (int p = 0; p < NP; ++p) xf[p] = frag(xrow + p * XPL, XST);
                    f32x16& c = acc[ti];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], xf[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], xf[1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], xf[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], xf[1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], xf[2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], xf[0], c, 0, 0, 0);
                }
            }
        }
#ifdef BMC_BF_STAMP
        c_work += __builtin_amdgcn_s_memtime() - w0;
#endif
        __syncthreads();
    }
#ifdef BMC_BF_STAMP
    if ((tid == 0 || tid == 256) && blockIdx.x < 1024) { g_stamp[blockIdx.x * 8 + 4 + (tid >> 8) * 2] = c_work; g_stamp[blockIdx.x * 8 + 5 + (tid >> 8) * 2] = __builtin_amdgcn_s_memtime() - c_t0; }
#endif
    __syncthreads();                // the producers' bias fold
    if (wave_active) {
        float* const sl = a.slabs + (((long long)split * a.G + g) * 9) * a.Mpad * a.Npad;
        const int n = n0 + li;
#pragma unroll
        for (int ti = 0; ti < NTAP; ++ti)
            if (n < a.Npad && ti < ntap) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + 32 * mw + (r & 3) + 8 * (r >> 2) + 4 * th;
                    sl[((long long)(tap_lo + ti) * a.Mpad + m) * a.Npad + n] = acc[ti][r];
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
    const int tgs = (taps == 9 && planes == 1 && k.tap_groups == 3) ? 3 : 1;
    dim3 grid((unsigned)((long long)k.G * k.n_mblk * k.n_nblk * tgs * k.nsplit)), block(512);
    const bool tab = pgemm_uses_tables(k);
    if (taps == 9) {
        if (planes == 3) { if (tab) hipLaunchKernelGGL(pgemm_bf9x3_kernel<true>, grid, dim3(768), 0, st, k); else hipLaunchKernelGGL(pgemm_bf9x3_kernel<false>, grid, dim3(768), 0, st, k); }
        else if (tab) hipLaunchKernelGGL((pgemm_bf_kernel<9, 1, true>), grid, block, 0, st, k);
        else hipLaunchKernelGGL((pgemm_bf_kernel<9, 1>), grid, block, 0, st, k);
    } else {
        if (planes == 3) { if (tab) hipLaunchKernelGGL((pgemm_bf_kernel<1, 3, true>), grid, block, 0, st, k); else hipLaunchKernelGGL((pgemm_bf_kernel<1, 3>), grid, block, 0, st, k); }
        else if (tab) hipLaunchKernelGGL((pgemm_bf_kernel<1, 1, true>), grid, block, 0, st, k);
        else hipLaunchKernelGGL((pgemm_bf_kernel<1, 1>), grid, block, 0, st, k);
    }
    BMC_CHECK_LAUNCH("bmc_pgemm (bf16 planes)");
    return 0;
}
